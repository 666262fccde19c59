#!/bin/bash
# r03 run 40: step with pass 3 on the FP4 x FP6 instruction (image from the MLP backward) against the bf16 kernel (-DNADM_P3_BF16; it still pays the image)
mkdir -p gpurun_out
for i in 1 2; do bash tools/abl_run.sh; done > gpurun_out/r03_40_ab.txt 2>&1
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03_40_bench.json
cat gpurun_out/r03_40_ab.txt; cut -c1-400 gpurun_out/r03_40_bench.json
