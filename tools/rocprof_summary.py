"""Summarise a rocprofv3 rocpd (.db) kernel trace into a per-kernel table (name, calls, total/avg/min/max us, %)."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void amds::", "").replace("amds::", "")
    return name[:110]


def main(path: str) -> None:
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {namecol}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"{'kernel':110s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:110s} {a[0]:7d} {a[1]:12.1f} {a[1]/a[0]:10.2f} {a[2]:10.2f} {a[3]:10.2f} {100*a[1]/tot:6.2f}")
    print(f"{'TOTAL':110s} {sum(a[0] for a in agg.values()):7d} {tot:12.1f}")


def by_shape(path: str, alternate: int = 1) -> None:
    """per (kernel, grid) table: one row per launch SHAPE, so the dominant kernel's average per GEMM shape can be read off."""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    gcols = [c for c in ("grid_x", "grid_y", "grid_z", "grid_size_x", "grid_size_y", "grid_size_z", "grid_size") if c in cols]
    wcols = [c for c in ("workgroup_x", "workgroup_size_x", "workgroup_size") if c in cols]
    sel = ", ".join([namecol, "start", "end"] + gcols + wcols[:1])
    agg = {}
    seen = {}
    for row in db.execute(f"select {sel} from kernels order by start").fetchall():
        n, s, e = row[:3]
        key = (short(n), tuple(row[3:]))
        if alternate > 1:       # launches of one (kernel, grid) that alternate between GEMM shapes differing only in K (proj / fc2)
            i = seen[key] = seen.get(key, -1) + 1
            key = (key[0], key[1] + (f"#{i % alternate}",))
        a = agg.setdefault(key, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"columns after the name: {gcols + wcols[:1]} (grid in work-items, workgroup size)")
    print(f"{'kernel':96s} {'grid/wg':>26s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for (k, g), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:96]:96s} {str(g):>26s} {a[0]:7d} {a[1]:12.1f} {a[1]/a[0]:10.2f} {a[2]:10.2f} {a[3]:10.2f} {100*a[1]/tot:6.2f}")
    print(f"{'TOTAL':96s} {'':>26s} {sum(a[0] for a in agg.values()):7d} {tot:12.1f}")


def sequence(path: str, last: int) -> None:
    """print the last `last` dispatches in start order (one forward = a fixed launch chain)"""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {namecol}, start, end from kernels order by start").fetchall()[-last:]
    t0 = rows[0][1]
    for i, (n, s, e) in enumerate(rows):
        print(f"{i:4d} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.2f} us  {short(n)}")
    print(f"span {(rows[-1][2] - t0) / 1e3:.1f} us, busy {sum(e - s for _, s, e in rows) / 1e3:.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2] == "--seq":
        sequence(sys.argv[1], int(sys.argv[3]))
    elif len(sys.argv) > 2 and sys.argv[2] == "--by-shape":
        by_shape(sys.argv[1], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    else:
        main(sys.argv[1])
