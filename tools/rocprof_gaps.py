"""GPU idle gaps in a rocprofv3 kernel trace (.db): span, busy time (union of kernel intervals), and the largest gaps with the kernels around
them.  python tools/rocprof_gaps.py trace.db [min_gap_us] [skip_first_seconds]"""
import sqlite3
import sys

from rocprof_summary import short

db = sqlite3.connect(sys.argv[1])
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 500.0
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
t0 = rows[0][1]
rows = [(short(n)[:50], (s - t0) / 1e3, (e - t0) / 1e3) for n, s, e in rows if (s - t0) / 1e9 >= skip]
span = rows[-1][2] - rows[0][1]
busy, cur_end, gaps = 0.0, rows[0][1], []
prev = None
for n, s, e in rows:
    if s > cur_end:
        if s - cur_end >= min_gap:
            gaps.append((s - cur_end, cur_end, prev, n))
        busy += 0.0
        cur_start = s
    busy += max(0.0, e - max(s, cur_end))
    if e > cur_end:
        cur_end, prev = e, n
print(f"span {span / 1e3:.1f} ms, busy {busy / 1e3:.1f} ms ({100 * busy / span:.1f} %), {len(gaps)} gaps >= {min_gap:.0f} us totalling {sum(g[0] for g in gaps) / 1e3:.1f} ms")
for g, at, a, b in sorted(gaps, reverse=True)[:25]:
    print(f"  gap {g / 1e3:8.2f} ms at {at / 1e3:9.1f} ms   after {a:50s} before {b}")
