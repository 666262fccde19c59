#!/bin/bash
# Run bench.py against variant builds of libnadm.so (tools/abl/*.so); prints ms/step and decode kernel time.
for so in "" $(ls tools/abl/*.so 2>/dev/null); do
  for fl in "" "--no-loss"; do
    out=$(NADM_LIB=${so:+$PWD/$so} python bench.py --steps 60 --warmup 30 --no-cpu-baseline $fl 2>/dev/null | tail -1)
    echo "${so:-default} ${fl:-loss} $(python3 -c "
import json,sys
d=json.loads(sys.argv[1]); print('ms/step', round(d['ms_per_step'],4), 'decode_us', round(d['roofline']['kernel_ms']['decode_bce']*1e3,1))" "$out")"
  done
done
