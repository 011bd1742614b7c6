#!/usr/bin/env python
"""Yardstick only: what PyTorch-ROCm's own fused attention (the aotriton flash kernel
behind torch.nn.functional.scaled_dot_product_attention) reaches on the SAME bank problems as tf_ext_attn_fwd --
[2 branches * K query frames, h heads, S queries] x [K*S keys] at the BASELINE head dims.  Never called by the
product; tools/ only.  Prints ms and TFLOP/s of the bank part next to tf_ext_attn_fwd's whole launch."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tokenflow_amd import ops, workload  # noqa: E402
from attn_microbench import time_it  # noqa: E402


def main():
    shapes = [(8, 4096, 8, 40), (8, 1024, 8, 80), (8, 256, 8, 160), (10, 9216, 5, 64)]
    g = torch.Generator(device="cuda").manual_seed(0)
    for K, S, h, d in shapes:
        D = h * d
        q, k, v = (torch.randn(3 * K, S, D, generator=g, device="cuda").bfloat16() for _ in range(3))
        ours, _ = time_it(lambda: ops.ext_attn(q, k, v, h, d ** -0.5, False), reps=6)
        fl_all = workload.attn_flops(K, S, D)
        # bank problems only: 2 branches, every frame's queries against the branch's K*S keys, head-major copies
        # made OUTSIDE the timed region (the library kernel reads the [3K,S,D] tensors in place)
        qb = q[K:].reshape(2, K, S, h, d).permute(0, 3, 1, 2, 4).reshape(2, h, K * S, d).contiguous()
        kb = k[K:].reshape(2, K, S, h, d).permute(0, 3, 1, 2, 4).reshape(2, h, K * S, d).contiguous()
        vb = v[K:].reshape(2, K, S, h, d).permute(0, 3, 1, 2, 4).reshape(2, h, K * S, d).contiguous()
        fl_bank = 4.0 * 2 * K * S * K * S * D
        for name, backend in (("flash", torch.nn.attention.SDPBackend.FLASH_ATTENTION),
                              ("efficient", torch.nn.attention.SDPBackend.EFFICIENT_ATTENTION)):
            try:
                with torch.nn.attention.sdpa_kernel(backend):
                    F.scaled_dot_product_attention(qb, kb, vb)
                    t, _ = time_it(lambda: F.scaled_dot_product_attention(qb, kb, vb), reps=6)
                print(f"K={K} S={S} h={h} d={d}: torch SDPA[{name}] bank part {t:.3f} ms = {fl_bank / t / 1e9:.0f} TF/s | "
                      f"tf_ext_attn_fwd whole launch {ours:.3f} ms = {fl_all / ours / 1e9:.0f} TF/s", flush=True)
            except Exception as e:  # backend not available for this shape
                print(f"K={K} S={S} h={h} d={d}: torch SDPA[{name}] unavailable: {str(e)[:80]}", flush=True)


def nn_gemm():
    """The NN search of one cfg2 level-0 chunk as a PLAIN GEMM through hipBLASLt: [n*S, D] x [D, 2*S] -> the bf16
    similarity matrix the reference materialises (335 MB), no normalisation, no argmax -- next to tf_nn_search."""
    g = torch.Generator(device="cuda").manual_seed(1)
    for K, n, S, D in [(8, 5, 4096, 320), (8, 5, 1024, 640), (10, 8, 9216, 320)]:
        ln = torch.nn.functional.layer_norm
        piv = ln(torch.randn(K, S, D, generator=g, device="cuda"), (D,)).bfloat16()
        tgt = ln(torch.randn(n * S, D, generator=g, device="cuda"), (D,)).bfloat16()
        inv = ops.pivot_inv_norm(piv)
        y = piv[2:4].reshape(2 * S, D)
        fl = 2.0 * n * S * 2 * S * D
        t_gemm, _ = time_it(lambda: torch.matmul(tgt, y.T), reps=10)
        t_ours, _ = time_it(lambda: ops.nn_search(tgt, piv, inv, [3, 2]), reps=10)
        print(f"NN chunk n={n} S={S} D={D}: torch.matmul (hipBLASLt, writes the similarity matrix) {t_gemm * 1e3:.1f} us = "
              f"{fl / t_gemm / 1e9:.0f} TF/s | tf_nn_search (fused normalisation + argmax) {t_ours * 1e3:.1f} us = "
              f"{fl / t_ours / 1e9:.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
    nn_gemm()
