#!/bin/bash
# the driver's command twice more (another box): gpurun_out/r03_bench_default_again.txt
mkdir -p gpurun_out
for i in 1 2; do python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1; done > gpurun_out/r03_bench_default_again.txt
python3 -c "
import json
for l in open('gpurun_out/r03_bench_default_again.txt'):
    d=json.loads(l); r=d['roofline']; print(round(d['ms_per_step'],4), '%.4g' % d['value'], {k: round(v*1e3,1) for k,v in r['kernel_ms'].items()}, r['traffic'] is not None)"
