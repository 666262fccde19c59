#!/bin/bash
mkdir -p gpurun_out
timeout 300 tools/abl/ubc | tee gpurun_out/r03_ubench_cndmask.txt
bash tools/scratch/r03_16.sh
