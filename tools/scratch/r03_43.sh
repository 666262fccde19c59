#!/bin/bash
# r03 run 43: FP4 x FP6 pass 3 with 16-byte loads: tests + rocprofv3 kernel durations (DEFAULT, without the Adam epilogue)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r03_43_tests.txt
out=gpurun_out/r03_43_p3.txt; : > $out
for v in DEFAULT; do
  lib=""; [ "$v" != "DEFAULT" ] && lib=$PWD/tools/abl/$v.so
  rm -rf /tmp/prof_x
  (cd /tmp && NADM_LIB=$lib rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o run -- python $R/bench.py --steps 40 --warmup 20 --ramp-ms 0 --no-cpu-baseline > /dev/null 2>&1)
  echo "== $v" >> $out
  python tools/prof_summary.py $(find /tmp/prof_x -name "*.db" | head -1) 20 | grep -i "encode_bwd\|dz_image\|decode_bce\|mlp_bwd" | cut -c1-60,70-130 >> $out
done
cat gpurun_out/r03_43_tests.txt $out
rm -f tools/abl/*.so
bash tools/abl_run.sh > gpurun_out/r03_43_ab.txt 2>&1; cat gpurun_out/r03_43_ab.txt
