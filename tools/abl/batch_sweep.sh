for bb in 100 200 400 800; do for m in "" "--force-ddp"; do
python bench.py --batch $bb $m --steps 60 --warmup 10 --no-cpu-baseline --time-kernels all 2>/dev/null | BB=$bb MM="$m" python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch', os.environ['BB'], os.environ['MM'], round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['roofline']['kernel_ms'].items()})"
done; done
