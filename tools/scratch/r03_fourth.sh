#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_gpu.txt 2>&1
tail -5 gpurun_out/r03_pytest_gpu.txt
run() { python bench.py --steps 60 --warmup 10 --ramp-ms 300 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('BENCH', '${NADM_LIB##*/}', ' '.join(sys.argv[1:]), round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['roofline']['kernel_ms'].items()}, d['loss_last_step'])" "$@"; }
(run; run --time-kernels all; run --force-ddp; run --force-ddp --time-kernels all; run --k 16 --rows 20000 --snps 1000000; NADM_LIB=$PWD/tools/abl/fastloss8.so run --k 16 --rows 20000 --snps 1000000; run --k 12 --rows 20000 --snps 1000000; NADM_LIB=$PWD/tools/abl/fastloss8.so run --k 12 --rows 20000 --snps 1000000) 2>&1 | tee gpurun_out/r03_fourth_bench.txt
