#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/scratch/ddp_host_cost.py > gpurun_out/r03_ddp_host_cost.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "pack2bit or rccl or small_parameter or production" 2>&1 | tail -4
run() { python bench.py --steps 60 --warmup 10 --ramp-ms 300 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('BENCH', ' '.join(sys.argv[1:]), round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['roofline']['kernel_ms'].items()})" "$@"; }
run; run --force-ddp
cat gpurun_out/r03_ddp_host_cost.txt | head -60
