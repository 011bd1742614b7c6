#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 rocpd SQLite) kernel trace into the per-kernel stats table
that `--stats` would print as CSV: calls, total / average / min / max duration, share.
Kernels are split by grid size so that the levels of the UNet show up separately.

    python tools/rocpd_stats.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.csv
"""
import sqlite3
import sys


def main(path, by_grid=True):
    cur = sqlite3.connect(path).cursor()
    key = "name, grid_x, grid_y, grid_z" if by_grid else "name"
    rows = cur.execute(
        f"select {key}, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        f"max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by {key} "
        f"order by sum(duration) desc").fetchall()
    total = sum(r[-7] if by_grid else r[2] for r in rows) or 1
    print("kernel,grid,calls,total_ms,avg_us,min_us,max_us,pct,vgpr,agpr,lds_bytes")
    for r in rows:
        if by_grid:
            name, gx, gy, gz, calls, tot, avg, mn, mx, vg, ag, lds = r
            grid = f"{gx}x{gy}x{gz}"
        else:
            name, calls, tot, avg, mn, mx, vg, ag, lds = r
            grid = "-"
        name = name.replace(",", ";")
        if len(name) > 110:
            name = name[:107] + "..."
        print(f"\"{name}\",{grid},{calls},{tot / 1e6:.3f},{avg / 1e3:.1f},{mn / 1e3:.1f},{mx / 1e3:.1f},"
              f"{100.0 * tot / total:.2f},{vg},{ag},{lds}")


if __name__ == "__main__":
    main(sys.argv[1], by_grid="--no-grid" not in sys.argv)
