"""Attention time of ONE rank of a head-sharded run (part="bank", heads/N heads, all keyframes) against the
single-GPU call: shows what the small per-rank grids cost and what the split form (runs of bank frames + merge)
recovers.  TOKENFLOW_ATTN_NO_SPLIT=1 for the one-pass form."""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from tokenflow_amd import ops
from attn_microbench import time_it
g = torch.Generator(device="cuda").manual_seed(0)
for H, K, S, d, label in [(1, 8, 4096, 40, "N=8 rank, level 0"), (2, 8, 4096, 40, "N=4 rank"), (4, 8, 4096, 40, "N=2 rank"), (8, 8, 4096, 40, "N=1"),
                          (1, 8, 1024, 80, "N=8 rank, level 1"), (8, 8, 1024, 80, "N=1 level 1"),
                          (1, 8, 256, 160, "N=8 rank, level 2"), (8, 8, 256, 160, "N=1 level 2"),
                          (1, 8, 64, 160, "N=8 rank, level 3"), (8, 8, 64, 160, "N=1 level 3")]:
    q, k, v = (torch.randn(3 * K, S, H * d, generator=g, device="cuda").bfloat16() for _ in range(3))
    for inj in (False, True):
        avg, mn = time_it(lambda: ops.ext_attn(q, k, v, H, d ** -0.5, inj, part="bank"), reps=10)
        fl = 4.0 * K * S * H * d * (2 * K * S)
        print(f"{label:22s} H={H} inject={int(inj)} bank-only: {avg*1e3:7.1f} us  {fl/avg/1e9:6.0f} TF/s", flush=True)
