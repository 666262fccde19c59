#!/bin/bash
# bench line + per-kernel rocprofv3 summary (kernel trace only) of the SAME command -> gpurun_out/<tag>_kernel_stats.txt
#   tools/step_profile.sh [tag] [bench flags]      (default tag r02)
tag=${1:-r02}; shift
R=$PWD; export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/${tag}_kernel_stats.txt
cmd="python $R/bench.py --steps 100 --warmup 60 --ramp-ms 0 --no-cpu-baseline --no-epoch-loop --no-clock-probe $*"
echo "# $cmd   (first 60 calls of every kernel = the untimed warm-up, left out of the table)" > $out
$cmd 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l[:1] == chr(123)][-1])
print('# bench line: ms/step', round(d['ms_per_step'],4), 'genotypes/s %.4g' % d['value'], 'kernel_ms (HIP events)', {k: (round(v,4) if not isinstance(v, list) else [round(x,4) for x in v]) for k,v in d['roofline']['kernel_ms'].items()})" >> $out
rm -rf /tmp/prof_$tag
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o run -- $cmd > /dev/null 2>&1)
python $R/tools/prof_summary.py $(find /tmp/prof_$tag -name "*.db" | head -1) 60 | grep -v synth_kernel | head -14 | cut -c1-60,70-130 >> $out
cat $out
