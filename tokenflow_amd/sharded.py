"""Frame-sharded multi-GPU execution of the hot path (one process per GPU).

The reference is single-process (no parallelism of any kind); this is new.
Rank r of W owns the contiguous chunks [r*C/W, (r+1)*C/W) of the video and the keyframes
drawn from them (K = C keyframes, one per chunk, run_tokenflow_pnp.py:224).  Per block there
are exactly two exchange steps, both through torch.distributed (backend "nccl" = RCCL over
xGMI on MI355X; "gloo" in the CPU tests):

 1. pivotal pass -- all-gather of the key/value bank: every query of a local keyframe
    attends to the keys/values of ALL K keyframes of its branch (tokenflow_utils.py:133-138).
    Only what is read remotely travels: without injection K and V of uncond and cond
    (4 slabs of [K/W,S,D]); with q/k injection (124-130) the key bank is the SOURCE branch's
    for both, so 3 slabs.  Each slab is gathered straight into its place in the
    [3,K,S,D] bank the kernel reads (no re-layout copy); the source branch's own frames are
    copied locally.  `ops.ext_attn(q_local, k_bank, v_bank, q_frame0=...)` then computes only
    the local keyframes' queries.
 2. propagation passes -- chunk c needs keyframes c and c-1 (331-333): the first local chunk's
    left neighbour lives on rank r-1, so each rank sends its LAST keyframe's pivot features,
    inverse norms and attention output to rank r+1 (one point-to-point message per block).

Work is partitioned, not re-associated: every output element is produced by exactly the same
kernel arithmetic as on one GPU, so sharded results equal single-process results bit for bit.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from . import ops


class FrameShard:
    def __init__(self, K: int, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if K % self.world:
            raise ValueError(f"{K} keyframes do not divide over {self.world} ranks "
                             "(uneven frame shards are not supported yet)")
        self.K = K
        self.Kl = K // self.world            # local keyframes == local chunks
        self.kf0 = self.rank * self.Kl       # first global keyframe / chunk of this rank

    # ------------------------------------------------------------------ pivotal pass
    def gather_bank(self, k_local: torch.Tensor, v_local: torch.Tensor, inject: bool
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
        """k_local, v_local: [3*Kl, S, D] -> banks [3*K, S, D] holding every slab the kernel reads."""
        if self.world == 1:
            return k_local, v_local
        B, S, D = k_local.shape
        Kl, K = self.Kl, self.K
        kl, vl = k_local.contiguous().view(3, Kl, S, D), v_local.contiguous().view(3, Kl, S, D)
        kb = torch.empty(3, K, S, D, dtype=k_local.dtype, device=k_local.device)
        vb = torch.empty(3, K, S, D, dtype=v_local.dtype, device=v_local.device)
        sl = slice(self.kf0, self.kf0 + Kl)
        works = []
        for b in range(3):
            k_remote = (b == 0) if inject else (b > 0)     # key bank read across frames?
            v_remote = b > 0
            for need, bank, loc in ((k_remote, kb, kl), (v_remote, vb, vl)):
                if need:
                    works.append(dist.all_gather_into_tensor(bank[b], loc[b], group=self.group, async_op=True))
                else:
                    bank[b, sl].copy_(loc[b])               # only this rank's own frames are read
        for w in works:
            w.wait()
        return kb.view(3 * K, S, D), vb.view(3 * K, S, D)

    def pivotal_attention(self, q_local, k_local, v_local, heads: int, scale: float, inject: bool):
        """Extended attention for the local keyframes against the all-gathered bank -> [3*Kl,S,D]."""
        kb, vb = self.gather_bank(k_local, v_local, inject)
        return ops.ext_attn(q_local, kb, vb, heads, scale, inject, q_frame0=self.kf0)

    # ------------------------------------------------------------------ halo for propagation
    def exchange_halo(self, pivots_local: torch.Tensor, inv_local: torch.Tensor, kf_out_local: torch.Tensor):
        """pivots_local [Kl,S,D], inv_local [Kl,S], kf_out_local [3*Kl,S,D] (this rank's keyframes).
        Returns the same three with ONE extra leading keyframe slot = the previous rank's last
        keyframe (unused zeros on rank 0: global chunk 0 matches a single keyframe, 331-333)."""
        Kl = self.Kl
        if self.world == 1:
            return pivots_local, inv_local, kf_out_local      # no halo slot: ids are [c, c-1] directly
        _, S, D = pivots_local.shape
        piv = torch.zeros(Kl + 1, S, D, dtype=pivots_local.dtype, device=pivots_local.device)
        inv = torch.zeros(Kl + 1, S, dtype=inv_local.dtype, device=inv_local.device)
        kfo = torch.zeros(3, Kl + 1, S, D, dtype=kf_out_local.dtype, device=kf_out_local.device)
        piv[1:].copy_(pivots_local)
        inv[1:].copy_(inv_local)
        kfo[:, 1:].copy_(kf_out_local.view(3, Kl, S, D))
        if self.world > 1:
            sends, recvs, opsl = [], [], []
            if self.rank + 1 < self.world:
                sends = [pivots_local[-1].contiguous(), inv_local[-1].contiguous(),
                         kf_out_local.view(3, Kl, S, D)[:, -1].contiguous()]
                opsl += [dist.P2POp(dist.isend, t, self._peer(self.rank + 1), self.group) for t in sends]
            if self.rank > 0:
                recvs = [torch.empty_like(piv[0]), torch.empty_like(inv[0]),
                         torch.empty(3, S, D, dtype=kfo.dtype, device=kfo.device)]
                opsl += [dist.P2POp(dist.irecv, t, self._peer(self.rank - 1), self.group) for t in recvs]
            for req in (dist.batch_isend_irecv(opsl) if opsl else []):
                req.wait()
            if recvs:
                piv[0].copy_(recvs[0])
                inv[0].copy_(recvs[1])
                kfo[:, 0].copy_(recvs[2])
        return piv, inv, kfo.view(3 * (Kl + 1), S, D)

    def _peer(self, group_rank: int) -> int:
        return dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank

    # ------------------------------------------------------------------ propagation pass
    def propagate(self, j: int, tgt: torch.Tensor, residual: torch.Tensor, piv_ext, inv_ext, kf_out_ext,
                  w: torch.Tensor, n: int, out_dtype_two: torch.dtype = torch.float32):
        """Local chunk j (global chunk kf0 + j): NN search + gather/blend/residual.
        tgt [n*S, D] 16-bit source-branch features, residual [3n,S,D]."""
        c = self.kf0 + j
        o = 1 if self.world > 1 else 0                   # halo slot offset
        ids = [j + o] if c == 0 else [j + o, j + o - 1]  # slots of keyframes [c, c-1] (tokenflow_utils.py:331-333)
        idx = ops.nn_search(tgt, piv_ext, inv_ext, ids)
        blend_dtype = out_dtype_two if len(ids) == 2 else kf_out_ext.dtype
        out_dtype = torch.promote_types(blend_dtype, residual.dtype) if residual is not None else blend_dtype
        return ops.gather_blend(kf_out_ext, idx, w if len(ids) == 2 else None, ids, n, residual, out_dtype)
