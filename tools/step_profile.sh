#!/bin/bash
# bench + per-kernel rocprofv3 summary of the same command (kernel trace only)
python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l[:1] == chr(123)][-1]); print('ms/step', d['ms_per_step'], 'genotypes/s', d['value'])"
R=$PWD; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats -d /tmp/prof -o run -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > /dev/null 2>&1
python $R/tools/prof_summary.py $(find /tmp/prof -name "*.db" | head -1) | grep -v synth_kernel | head -12 | cut -c1-60,70-130
