#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_gpu.txt 2>&1
tail -15 gpurun_out/r03_pytest_gpu.txt
for fl in "" "--no-loss" "--k 16 --rows 20000 --snps 1000000" "--k 7 --rows 2504 --snps 600000" "--min-k 2 --max-k 10 --rows 2504 --snps 600000 --steps 30"; do python bench.py --steps 60 --warmup 10 --ramp-ms 300 --no-cpu-baseline $fl 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('BENCH', '$fl', round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['roofline']['kernel_ms'].items()}, d['loss_last_step'])"; done 2>&1 | tee gpurun_out/r03_third_bench.txt
