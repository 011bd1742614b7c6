#!/usr/bin/env python
"""Shader clock and board power WHILE the hot kernels run (is the launch time of the level-0 attention the sum of
its matrix and vector work because the chip is power-limited?).  A one-wave probe kernel on a side stream counts
shader-clock ticks over a fixed span of the constant 100 MHz counter while the main stream replays one workload;
rocm-smi is sampled once in the middle of the replay.  tools/ only.
    python tools/clock_probe.py            (needs tools/ubench/libclock_probe.so, see clock_probe.hip)"""
import ctypes
import os
import subprocess
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from tokenflow_amd import ops, workload  # noqa: E402


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        keep = [ln.strip() for ln in out.splitlines() if any(k in ln for k in ("sclk", "Power", "mclk", "fclk"))]
        return " | ".join(keep)[:400]
    except Exception as e:  # noqa: BLE001
        return f"rocm-smi unavailable: {e}"


def main():
    lib = ctypes.CDLL(os.path.join(HERE, "ubench", "libclock_probe.so"))
    lib.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p]
    side = torch.cuda.Stream()
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    K, S, h, d = 8, 4096, 8, 40
    D = h * d
    q, k, v = (torch.randn(3 * K, S, D, generator=g, device="cuda").bfloat16() for _ in range(3))
    a = torch.randn(8192, 8192, generator=g, device="cuda").bfloat16()
    b = torch.randn(8192, 8192, generator=g, device="cuda").bfloat16()
    ln = torch.nn.functional.layer_norm
    n = 5
    piv = ln(torch.randn(K, S, D, generator=g, device="cuda"), (D,)).bfloat16()
    inv = ops.pivot_inv_norm(piv)
    tgt = ln(torch.randn(n * S, D, generator=g, device="cuda"), (D,)).bfloat16()
    big = torch.randn(1 << 28, generator=g, device="cuda")

    def gemm():
        return torch.matmul(a, b)

    jobs = [
        ("idle", None, 0),
        ("level-0 attention, plain", lambda: ops.ext_attn(q, k, v, h, d ** -0.5, False), workload.attn_flops(K, S, D)),
        ("level-0 attention, inject", lambda: ops.ext_attn(q, k, v, h, d ** -0.5, True), workload.attn_flops(K, S, D)),
        ("level-0 NN search (one chunk, 2 keyframes)", lambda: ops.nn_search(tgt, piv, inv, [3, 2]),
         workload.nn_flops(n, S, D, 2)),
        ("torch.matmul bf16 8192^3 (hipBLASLt)", gemm, 2.0 * 8192 ** 3),
        ("HBM copy 1 GiB", lambda: big.clone(), 0),
    ]
    for name, fn, fl in jobs:
        ms = 400.0
        reps = 0
        if fn is not None:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            one = time.perf_counter() - t0
            reps = max(4, int(0.8 / max(one, 1e-5)))
        torch.cuda.synchronize()
        # workload first (fills the queue), probe on the side stream a moment later so that it samples the middle
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        time.sleep(0.15)
        lib.clock_probe_launch(out.data_ptr(), ms, side.cuda_stream)
        time.sleep(0.05)
        s = smi()
        torch.cuda.synchronize()
        r, c = out.tolist()
        tot = e0.elapsed_time(e1) if reps else 0.0
        rate = f"{fl * reps / tot / 1e9:.0f} TF/s, " if fl and reps else ""
        print(f"{name}: shader clock {c / (r / 100e6) / 1e9:.3f} GHz over {r / 1e5:.0f} ms "
              f"({rate}{reps} launches in {tot:.0f} ms); rocm-smi: {s}", flush=True)


if __name__ == "__main__":
    main()
