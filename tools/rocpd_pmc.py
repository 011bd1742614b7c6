#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 PMC counters from a rocpd SQLite database
(`rocprofv3 --kernel-trace --pmc ...` on ROCm 7.2 writes *_results.db).

    python tools/rocpd_pmc.py run_results.db > summary.csv

One row per (kernel, grid, counter): dispatches, average value per dispatch, average
duration.  FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them (no correction
applied here; see DESIGN.md for the gfx950 calibration)."""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute(
        "select kernel_name, grid_size_x, grid_size_y, grid_size_z, counter_name, count(*), avg(value), "
        "avg(duration) from counters_collection group by kernel_name, grid_size_x, grid_size_y, grid_size_z, "
        "counter_name order by sum(duration) desc").fetchall()
    print("kernel,grid,counter,dispatches,avg_value,avg_duration_us")
    for name, gx, gy, gz, cname, n, val, dur in rows:
        name = name.replace(",", ";")
        if len(name) > 90:
            name = name[:87] + "..."
        print(f"\"{name}\",{gx}x{gy}x{gz},{cname},{n},{val:.1f},{dur / 1e3:.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
