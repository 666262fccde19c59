#!/bin/bash
mkdir -p gpurun_out
run() { python bench.py --steps 100 --warmup 20 --ramp-ms 300 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('BENCH', '$TAG', ' '.join(sys.argv[1:]), round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['roofline']['kernel_ms'].items()})" "$@"; }
run; TAG=cut run --force-ddp; TAG=uncut NADM_DDP_CUT=0 run --force-ddp; TAG=cut run --force-ddp --rows 8000; TAG=uncut NADM_DDP_CUT=0 run --force-ddp --rows 8000; run --rows 8000
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
