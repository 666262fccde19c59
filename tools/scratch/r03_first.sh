#!/bin/bash
# r03 first GPU call: occupancy micro-benchmark, today's baseline table, prologue ablations
mkdir -p gpurun_out
timeout 300 tools/abl/ubo > gpurun_out/r03_ubench_occupancy.txt 2>&1
timeout 600 bash tools/step_profile.sh r03_base > /dev/null 2>&1
ABL_ONLY="p2nosplit p1nosplit p1noload" timeout 900 bash tools/abl_run.sh > gpurun_out/r03_abl_prologue.txt 2>&1
(for fl in "" "--no-loss"; do python bench.py --steps 60 --warmup 10 --ramp-ms 300 --no-cpu-baseline --time-kernels all $fl 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('DEFAULT all-timed', '$fl', round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['roofline']['kernel_ms'].items()})"; done) >> gpurun_out/r03_abl_prologue.txt 2>&1
cat gpurun_out/r03_ubench_occupancy.txt gpurun_out/r03_abl_prologue.txt; head -12 gpurun_out/r03_base_kernel_stats.txt | cut -c1-150
