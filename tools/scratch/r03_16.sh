#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_gpu.txt 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r03_pytest_gpu.txt | tail -5
timeout 600 bash tools/step_profile.sh r03_e > /dev/null 2>&1; head -9 gpurun_out/r03_e_kernel_stats.txt | cut -c1-140
run() { python bench.py --steps 100 --warmup 20 --ramp-ms 300 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('BENCH', ' '.join(sys.argv[1:]), round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['roofline']['kernel_ms'].items()})" "$@"; }
run; run --no-loss; run --force-ddp
