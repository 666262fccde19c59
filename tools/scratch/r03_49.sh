#!/bin/bash
# r03 run 49: batch copy tiled by pass 3's chunks: tests + rocprofv3 kernel durations + step
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r03_49_tests.txt
out=gpurun_out/r03_49_p3.txt; : > $out
rm -rf /tmp/prof_x
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o run -- python $R/bench.py --steps 40 --warmup 20 --ramp-ms 0 --no-cpu-baseline > /dev/null 2>&1)
python tools/prof_summary.py $(find /tmp/prof_x -name "*.db" | head -1) 20 | grep -i "encode_bwd\|decode_bce\|encode_fwd\|mlp_" | cut -c1-60,70-130 >> $out
bash tools/abl_run.sh >> $out 2>&1
cat gpurun_out/r03_49_tests.txt $out
python bench.py --steps 50 --warmup 10 --k 7 --rows 2504 --snps 600000 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2', round(d['ms_per_step'],4), {n: round(v*1e3,1) for n,v in d['roofline']['kernel_ms'].items()})"
