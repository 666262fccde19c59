#!/bin/bash
mkdir -p gpurun_out
run() { python bench.py --steps 60 --warmup 10 --ramp-ms 300 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('BENCH', '${NADM_LIB##*/}', ' '.join(sys.argv[1:]), round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['roofline']['kernel_ms'].items()})" "$@"; }
(run --k 16 --rows 20000 --snps 1000000; NADM_LIB=$PWD/tools/abl/w2exact.so run --k 16 --rows 20000 --snps 1000000; NADM_LIB=$PWD/tools/abl/w2fast.so run --k 16 --rows 20000 --snps 1000000; NADM_LIB=$PWD/tools/abl/w2fast.so run --k 12 --rows 20000 --snps 1000000; run --k 12 --rows 20000 --snps 1000000) 2>&1 | tee gpurun_out/r03_k16_occ.txt
