#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace) into a per-kernel table
(name, calls, total ms, avg us, min us, max us, % of GPU kernel time).  Usage: prof_summary.py results.db [skip_first_n_calls]"""
import re
import sqlite3
import sys


def main(path, skip=0):
    c = sqlite3.connect(path)
    rows = c.execute("""
        select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d
        join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
    agg = {}
    for name, st, en in rows:
        name = re.sub(r"\(.*", "", name)
        name = re.sub(r"^void ", "", name)
        agg.setdefault(name, []).append((en - st) / 1e3)
    tot = sum(sum(v[skip:]) for v in agg.values())
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1][skip:])):
        w = v[skip:] or v
        print(f"{name[:70]:70s} {len(w):6d} {sum(w)/1e3:10.3f} {sum(w)/len(w):10.2f} {min(w):10.2f} {max(w):10.2f} {100*sum(w)/tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
