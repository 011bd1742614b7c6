#!/usr/bin/env python
"""Timeline of the LAST `--ms` milliseconds of a rocprofv3 (rocpd SQLite) trace: kernels and memory copies ordered by
start time, with duration, the idle gap of the device before each, and the stream / queue.  For reading the critical
path of a launch-bound sequence (a sharded rank's block at the coarse levels).

    rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/t -- python tools/rank_step_microbench.py ...
    python tools/rocpd_timeline.py /tmp/t/.../*_results.db --ms 6 > profiles/r04_rank_timeline.txt"""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name)
    return name[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--ms", type=float, default=6.0)
    ap.add_argument("--skip-ms", type=float, default=0.0, help="end the window this long before the last event")
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    ev = []
    kcols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    sid = "stream_id" if "stream_id" in kcols else ("queue_id" if "queue_id" in kcols else "0")
    for name, start, end, gx, s in cur.execute(f"select name, start, end, grid_x, {sid} from kernels"):
        ev.append((start, end, "K", f"{short(name)} grid {gx}", s))
    for tab in ("memory_copies", "memory_copy"):
        if tab in names:
            mcols = [r[1] for r in cur.execute(f"pragma table_info({tab})")]
            sz = "size" if "size" in mcols else "0"
            sidm = "stream_id" if "stream_id" in mcols else ("queue_id" if "queue_id" in mcols else "0")
            for start, end, size, s in cur.execute(f"select start, end, {sz}, {sidm} from {tab}"):
                ev.append((start, end, "C", f"copy {size} B", s))
            break
    ev.sort()
    t_end = max(e[1] for e in ev) - args.skip_ms * 1e6
    t0 = t_end - args.ms * 1e6
    busy_until = None
    print("start_us,dur_us,idle_before_us,kind,stream,what")
    for start, end, kind, what, s in ev:
        if start < t0 or start > t_end:
            continue
        gap = 0.0 if busy_until is None else max(0.0, (start - busy_until) / 1e3)
        busy_until = end if busy_until is None else max(busy_until, end)
        print(f"{(start - t0) / 1e3:9.1f},{(end - start) / 1e3:7.1f},{gap:6.1f},{kind},{s},{what}")


if __name__ == "__main__":
    main()
