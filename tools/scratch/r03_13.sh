#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/scratch/mlp_phase_cost.py 2>&1 | grep -v "^\[W\|amdgpu.ids" | tee gpurun_out/r03_mlp_phase_cost.txt
