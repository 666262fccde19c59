#!/bin/bash
# r03 run 45: full GPU test suite, then the complete profile collection of the round
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r03_45_tests.txt
bash tools/collect_profiles.sh > gpurun_out/r03_45_collect.txt 2>&1
cat gpurun_out/r03_45_tests.txt; tail -30 gpurun_out/r03_45_collect.txt
