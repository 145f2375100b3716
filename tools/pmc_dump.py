"""Per-kernel averages of every counter in a rocprofv3 --pmc rocpd database:  python tools/pmc_dump.py <results.db> [name-substring ...]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, value, duration, grid_size from counters_collection").fetchall()
agg = {}
for n, c, v, d, g in rows:
    n = re.sub(r"\(.*$", "", n).replace("void amds::", "")
    if len(sys.argv) > 2 and not any(p in n for p in sys.argv[2:]):
        continue
    a = agg.setdefault((n, g), {})
    e = a.setdefault(c, [0, 0.0, 0.0])
    e[0] += 1; e[1] += v; e[2] += d
for (n, g), cs in sorted(agg.items(), key=lambda kv: -max(e[2] for e in kv[1].values())):
    any_e = next(iter(cs.values()))
    print(f"{n[:100]} (grid {g}) calls {any_e[0]} avg {any_e[2] / any_e[0] / 1e3:.1f} us")
    for c, e in sorted(cs.items()):
        print(f"    {c:32s} {e[1] / e[0]:16.0f}")
