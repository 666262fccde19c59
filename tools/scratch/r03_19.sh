#!/bin/bash
mkdir -p gpurun_out
NADM_LIB=$PWD/tools/abl/mlpprobe.so timeout 600 python tools/scratch/mlp_probe.py 2>&1 | grep -v "^\[W\|amdgpu.ids" | tee gpurun_out/r03_mlp_probe.txt
