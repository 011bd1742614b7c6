#!/usr/bin/env python
"""Time tf_nn_search on BASELINE shapes (one process per library build, like attn_microbench.py):
    TOKENFLOW_HIP_LIB=<.so> python tools/nn_microbench.py [K,n,S,D ...]
Video-like targets (planted permutation + noise); prints avg/min us and TFLOP/s (2*n*S*S*D*P)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenflow_amd import ops  # noqa: E402
from fused_microbench import time_it  # noqa: E402  (HIP-graph replays: the small levels are 10-30 us kernels)


def main():
    shapes = [(8, 5, 4096, 320), (8, 5, 1024, 640), (8, 5, 256, 1280), (8, 5, 64, 1280), (4, 4, 64, 1280),
              (4, 4, 16, 1280)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
    g = torch.Generator(device="cuda").manual_seed(0)
    for K, n, S, D in shapes:
        piv = torch.nn.functional.layer_norm(torch.randn(K, S, D, generator=g, device="cuda"), (D,)).bfloat16()
        perm = torch.cat([torch.randperm(S, generator=g, device="cuda") for _ in range(n)])
        tgt = (piv[3].float()[perm] + 0.1 * torch.randn(n * S, D, generator=g, device="cuda")).bfloat16()
        inv = ops.pivot_inv_norm(piv)
        for ids in ([3, 2], [0]):
            avg, mn = time_it(lambda: ops.nn_search(tgt, piv, inv, ids), reps=10, warm=3)
            fl = 2.0 * n * S * S * D * len(ids)
            print(f"nn_search K={K} n={n} S={S} D={D} P={len(ids)}: avg {avg * 1e3:.1f} us  min {mn * 1e3:.1f} us  "
                  f"{fl / avg / 1e9:.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
