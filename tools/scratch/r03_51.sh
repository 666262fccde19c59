#!/bin/bash
# r03 run 51: counter calibration with the tiled batch copy's write pattern, the HBM / SQ passes and the bench line that reports them
mkdir -p gpurun_out profiles
timeout 900 python tools/pmc_profile.py calib hbm sq > gpurun_out/r03_pmc.log 2>&1
cp gpurun_out/r03_pmc_hbm.json gpurun_out/r03_pmc_sq.json gpurun_out/r03_pmc_calib.json profiles/
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
tail -3 gpurun_out/r03_pmc.log; cat gpurun_out/r03_pmc_calib.json | cut -c1-600; tail -1 gpurun_out/r03_bench_default.json | cut -c1-1500
