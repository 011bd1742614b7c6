#!/usr/bin/env python
"""A/B of the fused small-problem attention kernel (csrc/ext_attn_fused.hip) against the streaming kernels
(pre-pass + attention [+ merge]) of csrc/ext_attn.hip on the shapes it is meant for: the coarse levels of BASELINE
configs 1 and 2 on one GPU, and the two calls of a head-sharded rank of 8 (bank-only on one head group over all
keyframes; source-only on the rank's own frame, all heads).  Per shape: the streaming form (split + merge allowed), the
fused kernel in its automatic geometry, and every built geometry (query waves x key groups), with P in one value and
as hi + lo; GPU time per call from HIP-graph replays of 20 back-to-back calls (avg / min over `reps` replays).

    python tools/fused_microbench.py [--reps 20] [--only substring]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tokenflow_amd import _lib, ops  # noqa: E402


def time_it(fn, reps=20, warm=3, batch=20):
    """GPU time of one call: `batch` calls captured into ONE HIP graph and replayed `reps` times between two events --
    a single small call issued from Python is host-bound (~20 us of issue time against a 5-30 us kernel)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(batch):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / batch)
    return sum(ts) / len(ts), min(ts)


# (label, K, S, heads, head dim, part)
SHAPES = [
    ("cfg2 L2  1 GPU", 8, 256, 8, 160, "all"), ("cfg2 L3  1 GPU", 8, 64, 8, 160, "all"),
    ("cfg1 L0  1 GPU", 4, 1024, 8, 40, "all"), ("cfg1 L1  1 GPU", 4, 256, 8, 80, "all"),
    ("cfg1 L2  1 GPU", 4, 64, 8, 160, "all"), ("cfg1 L3  1 GPU", 4, 16, 8, 160, "all"),
    ("rank/8 L0 source", 1, 4096, 8, 40, "source"),
    ("rank/8 L1 bank", 8, 1024, 1, 80, "bank"), ("rank/8 L1 source", 1, 1024, 8, 80, "source"),
    ("rank/8 L2 bank", 8, 256, 1, 160, "bank"), ("rank/8 L2 source", 1, 256, 8, 160, "source"),
    ("rank/8 L3 bank", 8, 64, 1, 160, "bank"), ("rank/8 L3 source", 1, 64, 8, 160, "source"),
    ("cfg5 L2  1 GPU", 25, 256, 20, 64, "all"), ("cfg4 L3  1 GPU", 10, 144, 20, 64, "all"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    for label, K, S, h, d, part in SHAPES:
        if args.only and args.only not in label:
            continue
        D = h * d
        q, k, v = (torch.randn(3 * K, S, D, generator=g, device="cuda").bfloat16() for _ in range(3))
        nb = {"all": (1, 2), "bank": (0, 2), "source": (1, 0)}[part]
        fl = 4.0 * K * S * D * (nb[0] * S + nb[1] * K * S)
        for inj in (False, True):
            row = []

            def t(tag, **kw):
                try:
                    avg, mn = time_it(lambda: ops.ext_attn(q, k, v, h, d ** -0.5, inj, part=part, **kw), reps=args.reps, warm=3)
                    row.append(f"{tag} {avg * 1e3:6.1f}/{mn * 1e3:6.1f}")
                except Exception as e:  # noqa: BLE001
                    row.append(f"{tag} ERR {str(e)[:40]}")
            t("stream", fused=False)
            t("auto", fused=None)
            for geom in ((1, 4), (1, 8), (2, 4), (4, 2), (4, 1)):
                t("%dx%d" % geom, fused=True, hints=_lib.attn_hint(*geom) | _lib.TF_ATTN_NO_PRECISE_P)
                t("%dx%dp" % geom, fused=True, hints=_lib.attn_hint(*geom) | _lib.TF_ATTN_PRECISE_P)
            if d <= 80:
                t("1x4q2", fused=True, hints=_lib.attn_hint(1, 4, qb=2) | _lib.TF_ATTN_NO_PRECISE_P)
            print(f"{label:17s} K={K} S={S} h={h} d={d} inj={int(inj)} {fl / 1e9:7.1f} GFLOP | us avg/min: " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
