#!/bin/bash
mkdir -p gpurun_out
timeout 300 tools/abl/ubm > gpurun_out/r03_ubench_mfma_spacing.txt 2>&1
ABL_ONLY="med3 lossprod" timeout 900 bash tools/abl_run.sh > gpurun_out/r03_abl_bce.txt 2>&1
timeout 300 bash tools/abl_run.sh --nothing 2>/dev/null | head -0
(for fl in "" "--no-loss"; do python bench.py --steps 60 --warmup 10 --ramp-ms 300 --no-cpu-baseline $fl 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('DEFAULT', '$fl', round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['roofline']['kernel_ms'].items()}, d['loss_last_step'])"; done) >> gpurun_out/r03_abl_bce.txt 2>&1
cat gpurun_out/r03_ubench_mfma_spacing.txt gpurun_out/r03_abl_bce.txt
