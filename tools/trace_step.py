#!/usr/bin/env python3
"""Print the kernel timeline (start offset, duration, stream/queue) of the last full step from a rocprofv3 rocpd db."""
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "queue_id" if "queue_id" in cols else None
rows = c.execute(f"""select s.kernel_name, d.start, d.end{', d.' + qcol if qcol else ''} from rocpd_kernel_dispatch d
  join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
rows = [(re.sub(r"\(.*", "", r[0]), *r[1:]) for r in rows]
# last step = from the last encode_fwd but one
starts = [i for i, r in enumerate(rows) if "encode_fwd" in r[0]]
i0, i1 = starts[-2], starts[-1]
t0 = rows[i0][1]
for r in rows[i0:i1]:
    print(f"{(r[1]-t0)/1e3:9.1f} us  +{(r[2]-r[1])/1e3:7.1f}  q={r[3] if len(r)>3 else '-'}  {r[0][9:60]}")
print(f"step span {(rows[i1][1]-t0)/1e3:.1f} us")
