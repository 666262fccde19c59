#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/scratch/tail_experiment.py > gpurun_out/r03_pass2_tail.txt 2>&1; cat gpurun_out/r03_pass2_tail.txt | grep -v "^\[W\|amdgpu.ids"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_gpu.txt 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r03_pytest_gpu.txt | tail -5
