#!/usr/bin/env python3
"""Per-kernel mean of PMC counters from a rocprofv3 rocpd sqlite db (--pmc run). Usage: pmc_summary.py db [substr]"""
import re, sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
rows = c.execute("""select s.kernel_name, p.name, e.value, d.id from rocpd_pmc_event e
  join rocpd_info_pmc p on e.pmc_id = p.id
  join rocpd_kernel_dispatch d on e.event_id = d.event_id
  join rocpd_info_kernel_symbol s on d.kernel_id = s.id""").fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for k, n, v, did in rows:
    k = re.sub(r"\(.*", "", k)
    if sub in k:
        agg[k][n].append(v)
for k, d in agg.items():
    print(k[:90])
    for n, v in sorted(d.items()):
        print(f"    {n:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
