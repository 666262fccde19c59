#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 bash tools/step_profile.sh r03_a > /dev/null 2>&1
R=$PWD
rm -rf /tmp/prof_ddp; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ddp -o run -- python $R/bench.py --steps 40 --warmup 20 --ramp-ms 0 --no-cpu-baseline --force-ddp > /dev/null 2>&1)
DB=$(find /tmp/prof_ddp -name "*.db" | head -1)
python tools/trace_step.py $DB > gpurun_out/r03_ddp_timeline.txt 2>&1
python tools/prof_summary.py $DB 20 | head -16 | cut -c1-60,70-130 > gpurun_out/r03_ddp1_kernel_stats_a.txt
rm -rf /tmp/prof_pl; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_pl -o run -- python $R/bench.py --steps 40 --warmup 20 --ramp-ms 0 --no-cpu-baseline > /dev/null 2>&1)
python tools/trace_step.py $(find /tmp/prof_pl -name "*.db" | head -1) > gpurun_out/r03_plain_timeline.txt 2>&1
cat gpurun_out/r03_a_kernel_stats.txt | head -10 | cut -c1-140; cat gpurun_out/r03_ddp_timeline.txt gpurun_out/r03_ddp1_kernel_stats_a.txt gpurun_out/r03_plain_timeline.txt
