#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_gpu.txt 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r03_pytest_gpu.txt | tail -5
timeout 600 bash tools/step_profile.sh r03_b > /dev/null 2>&1; head -9 gpurun_out/r03_b_kernel_stats.txt | cut -c1-140
