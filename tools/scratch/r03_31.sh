#!/bin/bash
# r03 run 31: pass 3 on the FP4 x FP6 matrix instruction: GPU tests, then A/B against the bf16 kernel (same library source, -DNADM_P3_BF16)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r03_31_tests.txt
for i in 1 2; do bash tools/abl_run.sh; done > gpurun_out/r03_31_ab.txt 2>&1
cat gpurun_out/r03_31_tests.txt gpurun_out/r03_31_ab.txt
