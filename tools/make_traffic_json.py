#!/usr/bin/env python
"""profiles/traffic.json (read by bench.py as `roofline.traffic`) from two rocprofv3 PMC summaries of the level-0
attention launch -- FETCH_SIZE and WRITE_SIZE, separate passes (tools/gpu_session.sh stage `pmc`, summarised by
tools/rocpd_pmc.py).  The kernel NAME and grid are taken from the CSV row, so the file cannot go stale against a renamed
kernel.  Corrections: FETCH_SIZE x 2 (gfx950 counts a wide coalesced read at half its bytes, MI355X_MICROARCH.md HBM
section; calibrated in round 1 on gather_blend_kernel: 55,057 KiB reported vs 118 MB read), WRITE_SIZE x 1
(76,802 KiB reported vs 78.6 MB written); KiB -> bytes.

    python tools/make_traffic_json.py profiles/r04_pmc_attn_FETCH_SIZE.csv profiles/r04_pmc_attn_WRITE_SIZE.csv"""
import csv
import json
import os
import sys


def dominant(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["counter"] == counter and "ext_attn" in r["kernel"]]
    rows.sort(key=lambda r: float(r["avg_duration_us"]), reverse=True)
    return rows[0]


def main(fetch_csv, write_csv):
    f, w = dominant(fetch_csv, "FETCH_SIZE"), dominant(write_csv, "WRITE_SIZE")
    assert f["kernel"] == w["kernel"] and f["grid"] == w["grid"], "the two passes must profile the same launch"
    K, S, D = 8, 4096, 320
    out = {
        "_comment": "HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (separate --kernel-trace --pmc "
                    "runs of tools/attn_microbench.py 8,4096,8,40): FETCH_SIZE x 2 (gfx950 wide-read half-count) + WRITE_SIZE "
                    "x 1, KiB -> bytes.  Written by tools/make_traffic_json.py; bench.py reports the value as "
                    "roofline.traffic with traffic_source = this file; it is NOT measured inside the bench run.",
        "cfg2": {
            "ext_attn_l0_hbm_bytes_per_launch": int((2 * float(f["avg_value"]) + float(w["avg_value"])) * 1024),
            "fetch_size_kib": float(f["avg_value"]),
            "write_size_kib": float(w["avg_value"]),
            "algorithmic_bytes_per_launch": 3 * K * S * D * 4 * 2,
            "source": f"profiles/traffic.json <- {os.path.basename(fetch_csv)}, {os.path.basename(write_csv)} (rocprofv3 --pmc, "
                      f"separate passes of tools/attn_microbench.py; not measured in this run)",
            "kernel": f"{f['kernel']} grid {f['grid']} ({f['dispatches']} dispatches, avg {f['avg_duration_us']} us under the profiler)",
        },
    }
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out["cfg2"], indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
