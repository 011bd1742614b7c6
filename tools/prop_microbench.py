#!/usr/bin/env python
"""Propagation of ALL chunks of one block: K calls of tf_nn_gather_blend (the reference's one chunk per UNet pass)
against ONE call of tf_nn_gather_blend_chunks, on BASELINE shapes.  TF_NN_MIN_WGS=<n> sets the grid target of the
multi-chunk search's pivot-range split.   python tools/prop_microbench.py [K,n,S,D ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenflow_amd import ops  # noqa: E402
from attn_microbench import time_it  # noqa: E402


def main():
    shapes = [(8, 5, 4096, 320), (8, 5, 1024, 640), (8, 5, 256, 1280), (8, 5, 64, 1280), (4, 2, 1024, 320), (4, 2, 16, 1280)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
    g = torch.Generator(device="cuda").manual_seed(0)
    for K, n, S, D in shapes:
        ln = torch.nn.functional.layer_norm
        piv = ln(torch.randn(K, S, D, generator=g, device="cuda"), (D,)).bfloat16()
        inv = ops.pivot_inv_norm(piv)
        kf = torch.randn(3 * K, S, D, generator=g, device="cuda").bfloat16()
        tgt = ln(torch.randn(K * n * S, D, generator=g, device="cuda"), (D,)).bfloat16()
        res = torch.randn(3, K, n, S, D, generator=g, device="cuda").bfloat16()
        resc = [res[:, j].reshape(3 * n, S, D).contiguous() for j in range(K)]
        s = torch.arange(0, n)
        w = torch.sigmoid(torch.abs(s + n - n // 2) / (torch.abs(s - n // 2) + torch.abs(s + n - n // 2))).cuda()
        nS = n * S

        def per_chunk():
            for c in range(K):
                ids = [c] if c == 0 else [c, c - 1]
                ops.propagate(tgt[c * nS:(c + 1) * nS], piv, inv, ids, kf, w if c else None, n, resc[c],
                              torch.float32 if c else torch.bfloat16)

        def batched():
            ops.propagate_chunks(tgt, piv, inv, kf, w, n, K, 0, True, res.view(3 * K * n, S, D), torch.float32)
        fl = 2.0 * n * S * S * D * (2 * K - 1)
        for name, fn in (("per-chunk", per_chunk), ("one call ", batched)):
            avg, mn = time_it(fn, reps=20, warm=3)
            print(f"propagate K={K} n={n} S={S} D={D} {name}: avg {avg * 1e3:.1f} us  min {mn * 1e3:.1f} us  "
                  f"(NN part {fl / avg / 1e9:.0f} TF/s if it were all of it)", flush=True)


if __name__ == "__main__":
    main()
