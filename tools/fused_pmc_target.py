#!/usr/bin/env python
"""Target of the PMC passes over the fused attention kernel (tools/gpu_session.sh stage `fusedpmc`): the bank call of a
head-sharded rank of 8 at cfg2 level 1 (K=8, S=1024, one head of 80) in the wave-private form, 12 plain launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokenflow_amd import _lib, ops  # noqa: E402

K, S, h, d = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8,1024,1,80").split(","))
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn(3 * K, S, h * d, generator=g, device="cuda").bfloat16() for _ in range(3))
for _ in range(12):
    ops.ext_attn(q, k, v, h, d ** -0.5, False, part="bank", fused=True,
                 hints=_lib.attn_hint(1, 4) | _lib.TF_ATTN_NO_PRECISE_P)
torch.cuda.synchronize()
