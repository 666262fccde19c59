#!/bin/bash
# r03 run 56: pass 1's batch-split rule: GPU tests, configs[1] and the default bench
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | head -4 > gpurun_out/r03_56.txt
for f in "--k 7 --rows 2504 --snps 600000" "" "--parallelism snp --batch 6400 --snps 62500"; do
  echo "bench $f: $(python bench.py --steps 50 --warmup 10 --no-cpu-baseline $f 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {n: round(v*1e3,1) for n,v in d['roofline']['kernel_ms'].items()})")" >> gpurun_out/r03_56.txt
done
cat gpurun_out/r03_56.txt
