#!/usr/bin/env python
"""Time tf_layer_norm against torch's layer_norm (autocast sequence: cast up, fp32 norm, cast down) on the
hook path's row shapes; prints us and effective GB/s (bytes of the fused form: read + write once)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenflow_amd import ops  # noqa: E402
from attn_microbench import time_it  # noqa: E402

for rows, D in [(24 * 4096, 320), (15 * 4096, 320), (15 * 1024, 640), (15 * 256, 1280), (15 * 64, 1280)]:
    x = torch.randn(rows, D, device="cuda").bfloat16()
    ln = torch.nn.LayerNorm(D).cuda()
    avg, mn = time_it(lambda: ops.layer_norm(x, ln.weight, ln.bias, ln.eps, torch.bfloat16, True), reps=20, warm=3)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        avg_t, mn_t = time_it(lambda: ln(x).to(torch.bfloat16), reps=20, warm=3)
    gb = rows * D * 4 / 1e9
    print(f"layer_norm rows={rows} D={D}: fused {mn * 1e3:.1f} us ({gb / mn * 1e3:.0f} GB/s)   "
          f"torch autocast {mn_t * 1e3:.1f} us", flush=True)

# the other two forms of the hook path: norm of the fp32 residual stream of a chunk pass (fp32 in, 16-bit out) and
# the residual-add form of the pivotal pass (bf16 + bf16 -> bf16 sum + bf16 norm)
for rows, D in [(15 * 4096, 320), (15 * 1024, 640), (24 * 4096, 320)]:
    ln = torch.nn.LayerNorm(D).cuda()
    xf = torch.randn(rows, D, device="cuda")
    avg, mn = time_it(lambda: ops.layer_norm(xf, ln.weight, ln.bias, ln.eps, torch.bfloat16), reps=20, warm=3)
    gb = rows * D * 6 / 1e9
    print(f"layer_norm fp32->bf16 rows={rows} D={D}: {mn * 1e3:.1f} us ({gb / mn * 1e3:.0f} GB/s)", flush=True)
    a, b = torch.randn(rows, D, device="cuda").bfloat16(), torch.randn(rows, D, device="cuda").bfloat16()
    avg, mn = time_it(lambda: ops.add_layer_norm(a, b, ln.weight, ln.bias, ln.eps, torch.bfloat16), reps=20, warm=3)
    gb = rows * D * 8 / 1e9
    print(f"add_layer_norm bf16+bf16 rows={rows} D={D}: {mn * 1e3:.1f} us ({gb / mn * 1e3:.0f} GB/s)", flush=True)
    avg, mn = time_it(lambda: xf + a, reps=20, warm=3)
    print(f"torch fp32 + bf16 add rows={rows} D={D}: {mn * 1e3:.1f} us ({rows * D * 10 / 1e9 / mn * 1e3:.0f} GB/s)", flush=True)
