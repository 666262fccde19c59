#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest_gpu.txt 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r03_pytest_gpu.txt | tail -5
bash tools/collect_profiles.sh > gpurun_out/r03_collect.txt 2>&1; tail -26 gpurun_out/r03_collect.txt
