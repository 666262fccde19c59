for v in "$@"; do
  echo "== $v"
  R=$PWD; cd /tmp; rm -rf /tmp/pp; NADM_LIB=$R/tools/abl/$v.so rocprofv3 --kernel-trace --stats -d /tmp/pp -o run -- python $R/bench.py --steps 60 --warmup 40 --ramp-ms 0 --no-cpu-baseline > /tmp/pp.log 2>&1; cd $R
  python tools/prof_summary.py $(find /tmp/pp -name "*.db" | head -1) 40 | grep "mlp_\|dq_pre" | cut -c1-50,70-130
  grep -o '"ms_per_step": [0-9.]*' /tmp/pp.log
done
