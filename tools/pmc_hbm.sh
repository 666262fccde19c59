#!/bin/bash
# HBM traffic of every kernel of one bench step: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one
# pass on gfx950), kernel dispatch records only.  Writes gpurun_out/pmc_hbm.txt and gpurun_out/pmc_hbm.json (decode kernel,
# corrected as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE x2 for 16 B/lane coalesced streams (checked earlier in the round
# on the stand-alone Adam launch, whose byte count is known); WRITE_SIZE uncorrected).
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
: > gpurun_out/pmc_hbm.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && rocprofv3 --pmc $c -d /tmp/pmc_$c -o run -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > /tmp/pmc_$c.log 2>&1)
  db=$(find /tmp/pmc_$c -name "*.db" | head -1)
  python $R/tools/pmc_summary.py $db nadm >> gpurun_out/pmc_hbm.txt 2>&1 || tail -5 /tmp/pmc_$c.log
done
python3 - <<'PY'
import re, json
txt = open('gpurun_out/pmc_hbm.txt').read()
vals = {}
cur = None
for line in txt.splitlines():
    if line.startswith('_Z'):
        cur = line.strip()
    else:
        m = re.match(r'\s+(\w+)\s+n=\s*(\d+)\s+mean=\s*([\d.]+)', line)
        if m and cur:
            vals.setdefault(cur, {})[m.group(1)] = (int(m.group(2)), float(m.group(3)))
dec = [k for k in vals if 'decode_bce' in k][0]
f, w = vals[dec]['FETCH_SIZE'][1], vals[dec]['WRITE_SIZE'][1]
b, M, K = 800, 500000, 8
out = {"kernel": dec.split('(')[0][:60], "fetch_size_kib_raw": f, "write_size_kib_raw": w,
       "correction": "reads x2 (gfx950 FETCH_SIZE tallies 128 B requests of 16 B/lane coalesced streams as 64 B; checked earlier in the round on "
                     "the stand-alone Adam launch over V and P, which reads 4 x 32 MiB and counted 62.6 MiB raw); writes uncorrected",
       "traffic_bytes_per_launch": (2 * f + w) * 1024.0,
       "algorithmic_bytes_per_launch": b * M / 4 + 36 * M * K}     # single-GPU step: Adam on P inside the launch (bench.py, alg_bytes)
json.dump(out, open('gpurun_out/pmc_hbm.json', 'w'), indent=1)
print(json.dumps(out))
PY
