#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max ms, share) from a rocprofv3 rocpd sqlite database,
optionally grouped by grid size.  Usage: tools/rocpd_stats.py results.db [--by-grid] > profiles/xxx.md"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    by_grid = "--by-grid" in sys.argv
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    grid = ", grid_x || 'x' || grid_y || 'x' || grid_z as grid" if by_grid and "grid_x" in cols else ""
    rows = con.execute(f"select {name}, (end - start) as dur {grid} from kernels").fetchall()
    agg = {}
    for r in rows:
        key = (r[0][:90], r[2] if grid else "")
        a = agg.setdefault(key, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += r[1]
        a[2] = min(a[2], r[1])
        a[3] = max(a[3], r[1])
    tot = sum(a[1] for a in agg.values())
    print(f"| kernel | {'grid | ' if grid else ''}calls | total ms | avg us | min us | max us | % |")
    print(f"|---|{'---|' if grid else ''}---|---|---|---|---|---|")
    for (k, g), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        gs = f"{g} | " if grid else ""
        print(f"| {k} | {gs}{a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.1f} | {a[2] / 1e3:.1f} | {a[3] / 1e3:.1f} | {100 * a[1] / tot:.1f} |")
    print(f"\ntotal kernel time {tot / 1e6:.3f} ms over {sum(a[0] for a in agg.values())} dispatches")


if __name__ == "__main__":
    main()
