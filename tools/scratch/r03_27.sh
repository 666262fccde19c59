#!/bin/bash
# r03 run 27: clamp-multiply gradient + polynomial pair loss: parity tests, then A/B against the previous build on the same box
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r03_27_tests.txt
for i in 1 2; do bash tools/abl_run.sh; done > gpurun_out/r03_27_ab.txt 2>&1
for k in 16 12; do
  for so in "" tools/abl/base.so; do
    lib=""; [ -n "$so" ] && lib=$PWD/$so
    echo "BENCH $so --k $k $(NADM_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --k $k --rows 20000 --snps 1000000 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {n: round(v*1e3,1) for n,v in d['roofline']['kernel_ms'].items()})")"
  done
done >> gpurun_out/r03_27_ab.txt 2>&1
cat gpurun_out/r03_27_tests.txt gpurun_out/r03_27_ab.txt
