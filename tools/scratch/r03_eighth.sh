#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_gpu.txt 2>&1; tail -4 gpurun_out/r03_pytest_gpu.txt
timeout 600 python tools/pmc_profile.py calib > gpurun_out/r03_pmc_calib.log 2>&1; tail -3 gpurun_out/r03_pmc_calib.log
run() { python bench.py --steps 100 --warmup 20 --ramp-ms 300 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('BENCH', ' '.join(sys.argv[1:]), round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['roofline']['kernel_ms'].items()})" "$@"; }
run; run --force-ddp; run --force-ddp --batch 100; run --batch 100; run --parallelism snp --batch 6400 --snps 62500; run --batch 6400 --snps 62500
