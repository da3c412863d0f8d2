#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 results database (rocprofv3 --kernel-trace --stats -o NAME writes NAME_results.db
on this image when no --output-format is given).   python tools/rocprof_db_stats.py DB [steps] [--csv OUT]"""
import collections
import sqlite3
import sys


def main():
    db = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 1.0
    out = sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z from {kd} d "
                     f"join {ks} s on d.kernel_id=s.id").fetchall()
    agg = collections.defaultdict(lambda: [0, 0, 1e30, 0])
    for (n, s, e, *_g) in rows:
        a = agg[n]
        a[0] += 1
        a[1] += e - s
        a[2] = min(a[2], e - s)
        a[3] = max(a[3], e - s)
    tot = sum(a[1] for a in agg.values())
    lines = ["kernel,calls_per_step,us_per_step,avg_us,percent,min_us,max_us"]
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"\"{n}\",{a[0] / steps:.2f},{a[1] / 1e3 / steps:.1f},{a[1] / 1e3 / a[0]:.2f},{100 * a[1] / tot:.2f},"
                     f"{a[2] / 1e3:.2f},{a[3] / 1e3:.2f}")
    text = "\n".join(lines)
    if out:
        with open(out, "w") as f:
            f.write(text + "\n")
    print("\n".join(l[:170] for l in lines[:30]))
    print(f"total {tot / 1e3 / steps:.1f} us per step")


if __name__ == "__main__":
    main()
