#!/usr/bin/env python
"""Time tf_ext_attn_fwd (and optionally tf_nn_search) on BASELINE shapes, one process per library
build so that kernel variants can be A/B-ed:  TOKENFLOW_HIP_LIB=<.so> python tools/attn_microbench.py
Prints avg/min ms over `reps` launches (HIP events on the launch stream) and TFLOP/s (algorithmic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenflow_amd import ops, workload  # noqa: E402


def time_it(fn, reps=8, warm=2):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) for a, b in ev]
    return sum(t) / len(t), min(t)


def main():
    shapes = [(8, 4096, 8, 40), (8, 1024, 8, 80), (8, 256, 8, 160), (10, 9216, 5, 64)]
    args = [a for a in sys.argv[1:] if a not in ("f16", "bf16")]
    dt = torch.float16 if "f16" in sys.argv[1:] else torch.bfloat16
    if args:
        shapes = [tuple(int(x) for x in a.split(",")) for a in args]
    g = torch.Generator(device="cuda").manual_seed(0)
    for K, S, h, d in shapes:
        D = h * d
        q, k, v = (torch.randn(3 * K, S, D, generator=g, device="cuda").to(dt) for _ in range(3))
        fl = workload.attn_flops(K, S, D)
        for inj in (False, True):
            avg, mn = time_it(lambda: ops.ext_attn(q, k, v, h, d ** -0.5, inj), reps=6 if S > 4096 else 10)
            print(f"ext_attn {str(dt)[6:]} K={K} S={S} h={h} d={d} inject={int(inj)}: avg {avg:.3f} ms  min {mn:.3f} ms  "
                  f"{fl / avg / 1e9:.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
