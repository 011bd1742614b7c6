#!/usr/bin/env python
"""GPU time of ONE rank's pivotal pass in the head-sharded form (FrameShard._pivotal_heads) with the wire taken
out: the two all-to-alls are replaced by local copies of the same size, so what is timed is pack + source
attention + unpack + bank attention + repack + final copy -- against the bank attention alone
(tools/rank_shard_microbench.py).  Emulates rank 0 of W (default 8) at the cfg2 levels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tokenflow_amd import sharded, workload  # noqa: E402
from attn_microbench import time_it  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8


class _Done:
    def wait(self):
        return True


def fake_a2a(recv, send, group, out_rows=None, in_rows=None, async_op=False):
    recv.view(-1)[:].copy_(send.reshape(-1)[:recv.numel()] if send.numel() >= recv.numel()
                           else send.reshape(-1).repeat((recv.numel() + send.numel() - 1) // send.numel())[:recv.numel()])
    return _Done() if async_op else None


sharded._all_to_all = fake_a2a
cfg = workload.CONFIGS["cfg2"]
K = cfg.K
sh = sharded.FrameShard.__new__(sharded.FrameShard)
sh.group, sh.world, sh.rank, sh.K = None, W, 0, K
sh.counts = [K // W + (1 if r < K % W else 0) for r in range(W)]
sh.offsets = [sum(sh.counts[:r]) for r in range(W)]
sh.even, sh.Kl, sh.kf0 = K % W == 0, sh.counts[0], 0
g = torch.Generator(device="cuda").manual_seed(0)
for lvl, (S, D, h) in enumerate(cfg.levels):
    q, k, v = (torch.randn(3 * sh.Kl, S, D, generator=g, device="cuda").bfloat16() for _ in range(3))
    for inj in (False, True):
        fn = lambda: sh._pivotal_heads(q, k, v, h, (D // h) ** -0.5, inj)
        avg, mn = time_it(fn, reps=20, warm=3)
        # host-side issue time: how long the CPU needs to enqueue the pass (GPU idle at the start, no sync inside)
        import time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        host = (time.perf_counter() - t0) / 20
        torch.cuda.synchronize()
        print(f"rank 0 of {W}, level {lvl} (S={S}, D={D}) inject={int(inj)}: pivotal pass without the wire "
              f"GPU {avg * 1e3:7.1f} us, host issue {host * 1e6:7.1f} us", flush=True)
