#!/bin/bash
# Run bench.py against variant builds of libnadm.so (tools/abl/*.so, see tools/build_variant.sh); prints ms/step and the
# pass-2 kernel time with and without the loss value.  ABL_ONLY="a b" restricts the run to tools/abl/{a,b}.so.
# Extra arguments go to bench.py.
list=""
if [ -n "$ABL_ONLY" ]; then for n in $ABL_ONLY; do list="$list tools/abl/$n.so"; done; else list="DEFAULT $(ls tools/abl/*.so 2>/dev/null)"; fi
for so in $list; do
  for fl in "" "--no-loss"; do
    lib=""; [ "$so" != "DEFAULT" ] && lib=$PWD/$so
    out=$(NADM_LIB=$lib python bench.py --steps 60 --warmup 10 --ramp-ms 300 --no-cpu-baseline $fl "$@" 2>/dev/null | tail -1)
    echo "$(basename $so .so) ${fl:-loss} $(python3 -c "
import json,sys
d=json.loads(sys.argv[1]); k=d['roofline']['kernel_ms']; print('ms/step', round(d['ms_per_step'],4), ' '.join(f'{n}={v*1e3:.1f}' for n,v in k.items()), 'loss', d['loss_last_step'])" "$out")"
  done
done
