#!/bin/bash
# r03 run 32: where does the FP4 x FP6 pass 3 spend its time?  rocprofv3 kernel durations of timing-only ablations
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
out=gpurun_out/r03_33_p3_abl.txt; : > $out
for v in p3_NOADAM p3_NOZ p3_NOIDX p3_noz_noidx; do
  lib=""; [ "$v" != "DEFAULT" ] && lib=$PWD/tools/abl/$v.so
  rm -rf /tmp/prof_x
  (cd /tmp && NADM_LIB=$lib rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o run -- python $R/bench.py --steps 40 --warmup 20 --ramp-ms 0 --no-cpu-baseline > /dev/null 2>&1)
  echo "== $v" >> $out
  python tools/prof_summary.py $(find /tmp/prof_x -name "*.db" | head -1) 20 | grep -i "encode_bwd" | cut -c1-60,70-130 >> $out
done
cat $out
