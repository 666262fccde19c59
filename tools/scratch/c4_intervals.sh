#!/bin/bash
# per-5-epoch durations of a c4 full run from the trainer's own log timestamps
tag=$1; shift
env "$@" NADM_LOG=1 python tools/full_run.py c4 2> gpurun_out/c4_$tag.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', 'total', round(d['total_s'],2), 'train', round(d['train_call_s'],2), 'gmm', round(d['train_phases']['gmm_init_s'],2))"
grep "Loss in epoch" gpurun_out/c4_$tag.err | python -c "
import sys
ts=[]
for l in sys.stdin:
    h,m,s=l.split()[0].split(':'); ts.append(int(h)*3600+int(m)*60+float(s))
d=[round((b-a)*1000/5,1) for a,b in zip(ts,ts[1:])]
print('   ms/epoch per 5-epoch interval: min', min(d), 'median', sorted(d)[len(d)//2], 'max', max(d)); print('  ', d)"
