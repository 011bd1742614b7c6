#!/usr/bin/env python
"""Decision data for the default of the Dh = 40 attention (folded softmax scale vs fp32 score scaling): at the
cfg2 level-0 shape (K = 8, S = 4096, h = 8, d = 40) with logits of increasing peakedness (q scaled by `gain`:
N(0,1) inputs give logit std 1; real SD self-attention logits are peaked, gain ~ 4-8), error of both kernels
against the fp32 oracle on sampled rows, next to the plain three-term bound of tests/test_kernels_gpu.py
(2e-4 + eps |ref| + eps softmax.|V|).  Prints one JSON line per (dtype, gain, inject, kernel).
TEST/ANALYSIS TOOL: imports the oracle-style row reference, never part of the product path."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenflow_amd import ops  # noqa: E402


def rows_ref(q, k, v, K, S, h, d, b, f, head, rows, inject):
    qv, kv, vv = (t.view(3, K, S, h, d) for t in (q, k, v))
    bq = 0 if (inject and b > 0) else b
    qr = qv[bq, f, rows, head].double()
    if b == 0:
        kk, vals = kv[0, f, :, head].double(), vv[0, f, :, head].double()
    else:
        kk, vals = kv[bq, :, :, head].reshape(K * S, d).double(), vv[b, :, :, head].reshape(K * S, d).double()
    p = torch.softmax(qr @ kk.T * d ** -0.5, dim=-1)
    return (p @ vals).float(), (p @ vals.abs()).float()


def main():
    K, S, h, d = 8, 4096, 8, 40
    D = h * d
    rows = torch.arange(0, S, 97)
    probs = [(0, 0, 0), (0, 7, 7), (1, 0, 3), (1, 7, 0), (2, 3, 7), (2, 6, 5)]
    for dtype in (torch.bfloat16, torch.float16):
        eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        for gain in (1.0, 2.0, 4.0, 8.0, 16.0):
            g = torch.Generator(device="cuda").manual_seed(7)
            q = (torch.randn(3 * K, S, D, generator=g, device="cuda") * gain).to(dtype)
            k, v = (torch.randn(3 * K, S, D, generator=g, device="cuda").to(dtype) for _ in range(2))
            qc, kc, vc = q.cpu(), k.cpu(), v.cpu()
            for inject in (False, True):
                refs = [rows_ref(qc, kc, vc, K, S, h, d, b, f, hd, rows, inject) for b, f, hd in probs]
                for name, fold in (("folded", True), ("fp32-scaled", False)):
                    out = ops.ext_attn(q, k, v, h, d ** -0.5, inject, fold_scale=fold).float().cpu().view(3, K, S, h, d)
                    worst = excess = 0.0
                    mean = []
                    for (b, f, hd), (ref, ref_abs) in zip(probs, refs):
                        err = (out[b, f, rows, hd] - ref).abs()
                        bound = 2e-4 + eps * (ref.abs() + ref_abs)
                        worst = max(worst, float(err.max()))
                        excess = max(excess, float((err / bound).max()))
                        mean.append(float(err.mean()))
                    print(json.dumps(dict(dtype=str(dtype)[6:], gain=gain, inject=inject, kernel=name,
                                          max_err=round(worst, 6), mean_err=round(sum(mean) / len(mean), 7),
                                          max_err_over_plain_bound=round(excess, 3))), flush=True)


if __name__ == "__main__":
    main()
