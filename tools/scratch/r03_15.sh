#!/bin/bash
mkdir -p gpurun_out
ABL_ONLY="stag1 stag3 stag8" timeout 900 bash tools/abl_run.sh > gpurun_out/r03_abl_stagger.txt 2>&1
(for fl in "" "--no-loss"; do python bench.py --steps 60 --warmup 10 --ramp-ms 300 --no-cpu-baseline $fl 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('DEFAULT', '$fl', round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['roofline']['kernel_ms'].items()})"; done) >> gpurun_out/r03_abl_stagger.txt 2>&1
cat gpurun_out/r03_abl_stagger.txt
