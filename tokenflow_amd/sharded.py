"""Frame-sharded multi-GPU execution of the hot path (one process per GPU).

The reference is single-process (no parallelism of any kind); this is new.
Rank r of W owns the contiguous chunks [r*C/W, (r+1)*C/W) of the video and the keyframes
drawn from them (K = C keyframes, one per chunk, run_tokenflow_pnp.py:224).  Per block there
are two exchange steps, both through torch.distributed (backend "nccl" = RCCL over xGMI on
MI355X; "gloo" in the CPU tests):

 1. pivotal pass -- every query of a keyframe attends to the keys/values of ALL K keyframes
    of its branch (tokenflow_utils.py:133-138).  Two exchange patterns, same results:

    "heads" (default when heads % W == 0): the bank branches are re-sharded from frames to
        heads for the attention and back.  Rank r sends, to every rank w, head group w of its
        keyframes' q, k, v (uncond + cond; with q/k injection, 124-130, the source q, k and
        the uncond/cond v), receives head group r of everybody's keyframes, runs
        `ops.ext_attn(part="bank")` on [3,K,S,D/W], and returns the outputs the same way.
        Two all-to-alls per block; a rank sends (6+2)*(W-1)/W (injection: (4+2)*(W-1)/W) of
        its local [K/W,S,D] slabs -- under 8 local slabs whatever W -- and on a fully
        connected xGMI mesh every pair uses its own link.  The source branch (own-frame
        keys only, 173/177) never leaves the rank: `ops.ext_attn(part="source")` runs on the
        local frames while the first exchange is in flight.
    "bank": all-gather of the key/value bank.  Only what is read remotely travels: without
        injection K and V of uncond and cond (4 slabs of [K/W,S,D] per rank, gathered to
        [K,S,D]); with injection 3 slabs.  Each slab is gathered straight into its place in
        the [3,K,S,D] bank the kernel reads; `ops.ext_attn(q_local, k_bank, v_bank,
        q_frame0=...)` then computes only the local keyframes' queries.  A rank receives
        4*(W-1) local slabs: 4x the "heads" volume at W = 8, and ring-bound.

 2. propagation passes -- chunk c needs keyframes c and c-1 (331-333): the first local chunk's
    left neighbour lives on rank r-1, so each rank sends its LAST keyframe's pivot features,
    inverse norms and attention output to rank r+1 (one point-to-point message per block).

Work is partitioned, not re-associated: every output element is produced by exactly the same
kernel arithmetic as on one GPU (the attention of one (query, head) visits the K frames in the
same order, whoever computes it), so sharded results equal single-process results bit for bit --
with one exception: the small grid of a rank makes `tf_ext_attn_fwd` split the bank over extra
workgroups and merge, which re-associates fp32 sums; TOKENFLOW_ATTN_NO_SPLIT=1
turns that off and restores bit-identical results.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from . import ops


class _Done:
    def wait(self):
        return True


def _all_to_all(recv: torch.Tensor, send: torch.Tensor, group, out_rows=None, in_rows=None, async_op: bool = False):
    """dist.all_to_all_single over dim 0 (row counts per peer; None = equal).  gloo (development boxes, the
    single-GPU tests) moves host memory only, so device tensors are staged through the host there.  RCCL
    takes the device buffers directly."""
    if send.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(host, send.cpu(), out_rows, in_rows, group=group)
        recv.copy_(host)
        return _Done() if async_op else None
    return dist.all_to_all_single(recv, send, out_rows, in_rows, group=group, async_op=async_op)


class FrameShard:
    """K keyframes (= chunks) over the ranks of `group` in contiguous runs; the first K % W ranks hold one more
    (cfg5's 25 chunks over 8 ranks -> 4,3,3,3,3,3,3,3)."""

    def __init__(self, K: int, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if K < self.world:
            raise ValueError(f"{K} keyframes cannot be sharded over {self.world} ranks (a rank would own none)")
        self.K = K
        self.counts = [K // self.world + (1 if r < K % self.world else 0) for r in range(self.world)]
        self.offsets = [sum(self.counts[:r]) for r in range(self.world)]
        self.even = K % self.world == 0
        self.Kl = self.counts[self.rank]       # local keyframes == local chunks
        self.kf0 = self.offsets[self.rank]     # first global keyframe / chunk of this rank

    # ------------------------------------------------------------------ pivotal pass
    def gather_bank(self, k_local: torch.Tensor, v_local: torch.Tensor, inject: bool
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
        """k_local, v_local: [3*Kl, S, D] -> banks [3*K, S, D] holding every slab the kernel reads."""
        if self.world == 1:
            return k_local, v_local
        B, S, D = k_local.shape
        Kl, K, W = self.Kl, self.K, self.world
        kl, vl = k_local.contiguous().view(3, Kl, S, D), v_local.contiguous().view(3, Kl, S, D)
        kb = torch.empty(3, K, S, D, dtype=k_local.dtype, device=k_local.device)
        vb = torch.empty(3, K, S, D, dtype=v_local.dtype, device=v_local.device)
        sl = slice(self.kf0, self.kf0 + Kl)
        works, pads = [], []
        for b in range(3):
            k_remote = (b == 0) if inject else (b > 0)     # key bank read across frames?
            v_remote = b > 0
            for need, bank, loc in ((k_remote, kb, kl), (v_remote, vb, vl)):
                if not need:
                    bank[b, sl].copy_(loc[b])               # only this rank's own frames are read
                elif self.even:                             # straight into place, no re-layout
                    works.append(dist.all_gather_into_tensor(bank[b], loc[b], group=self.group, async_op=True))
                else:                                       # uneven runs: equal-size padded contributions
                    Km = self.counts[0]
                    mine = torch.zeros(Km, S, D, dtype=loc.dtype, device=loc.device)
                    mine[:Kl].copy_(loc[b])
                    allp = torch.empty(W * Km, S, D, dtype=loc.dtype, device=loc.device)
                    works.append(dist.all_gather_into_tensor(allp, mine, group=self.group, async_op=True))
                    pads.append((bank[b], allp.view(W, Km, S, D)))
        for w in works:
            w.wait()
        for dst, allp in pads:
            for r in range(W):
                dst[self.offsets[r]:self.offsets[r] + self.counts[r]].copy_(allp[r, :self.counts[r]])
        return kb.view(3 * K, S, D), vb.view(3 * K, S, D)

    def pivotal_attention(self, q_local, k_local, v_local, heads: int, scale: float, inject: bool,
                          mode: Optional[str] = None):
        """Extended attention for the local keyframes against all K keyframes -> [3*Kl,S,D].
        mode: "heads" | "bank" | None (= "heads" when the heads divide over the ranks)."""
        if self.world == 1:
            return ops.ext_attn(q_local, k_local, v_local, heads, scale, inject)
        if mode is None:
            mode = "heads" if heads % self.world == 0 else "bank"
        if mode == "heads":
            return self._pivotal_heads(q_local, k_local, v_local, heads, scale, inject)
        kb, vb = self.gather_bank(k_local, v_local, inject)
        return ops.ext_attn(q_local, kb, vb, heads, scale, inject, q_frame0=self.kf0)

    def _pivotal_heads(self, q_local, k_local, v_local, heads: int, scale: float, inject: bool):
        W, Kl, K = self.world, self.Kl, self.K
        B, S, D = q_local.shape
        if heads % W:
            raise ValueError(f"{heads} heads do not divide over {W} ranks")
        hd, dev, dt = D // W, q_local.device, q_local.dtype
        q3, k3, v3 = (t.contiguous().view(3, Kl, S, W, hd) for t in (q_local, k_local, v_local))
        even = self.even
        # ---- pack, frame-major: send[w, f] = head group w of every slab of local keyframe f that the bank
        #      branches read.  The ranks' runs are contiguous and in rank order, so what arrives is already
        #      [K (global frame), ns, S, hd] whatever the run lengths.
        ns = 4 if inject else 6
        send = torch.empty(W, Kl, ns, S, hd, dtype=dt, device=dev)
        if inject:       # source q, k (what uncond and cond use, 124-130) and the two value banks
            send[:, :, 0].copy_(q3[0].permute(2, 0, 1, 3))
            send[:, :, 1].copy_(k3[0].permute(2, 0, 1, 3))
            send[:, :, 2:4].copy_(v3[1:3].permute(3, 1, 0, 2, 4))
        else:
            sv = send.view(W, Kl, 3, 2, S, hd)
            for t, x in enumerate((q3, k3, v3)):
                sv[:, :, t].copy_(x[1:3].permute(3, 1, 0, 2, 4))
        recv = torch.empty(K, ns, S, hd, dtype=dt, device=dev)
        work = _all_to_all(recv.view(K, -1), send.view(W * Kl, -1), self.group,
                           None if even else self.counts, None if even else [Kl] * W, async_op=True)
        # ---- source branch: own-frame keys, all heads, stays local (overlaps the exchange)
        out = torch.empty(3, Kl, S, D, dtype=dt, device=dev)
        ops.ext_attn(q_local, k_local, v_local, heads, scale, inject, out=out.view(3 * Kl, S, D), part="source")
        work.wait()
        # ---- unpack into the [3,K,S,hd] q / k / v the kernel reads; slabs it does not read stay unset
        bank = torch.empty(3, 3, K, S, hd, dtype=dt, device=dev)     # [q|k|v][branch][frame]
        if inject:
            bank[0, 0].copy_(recv[:, 0])
            bank[1, 0].copy_(recv[:, 1])
            bank[2, 1:3].copy_(recv[:, 2:4].permute(1, 0, 2, 3))
        else:
            bank[:, 1:3].copy_(recv.view(K, 3, 2, S, hd).permute(1, 2, 0, 3, 4))
        oh = ops.ext_attn(bank[0].view(3 * K, S, hd), bank[1].view(3 * K, S, hd), bank[2].view(3 * K, S, hd),
                          heads // W, scale, inject, part="bank")
        # ---- outputs back to the frame owners, frame-major again: rows of rank w's run go to rank w
        send2 = oh.view(3, K, S, hd)[1:3].permute(1, 0, 2, 3).contiguous()          # [K, 2, S, hd]
        recv2 = torch.empty(W, Kl, 2, S, hd, dtype=dt, device=dev)                   # [head group, my frames]
        _all_to_all(recv2.view(W * Kl, -1), send2.view(K, -1), self.group,
                    None if even else [Kl] * W, None if even else self.counts)
        out.view(3, Kl, S, W, hd)[1:3].copy_(recv2.permute(2, 1, 3, 0, 4))
        return out.view(3 * Kl, S, D)

    # ------------------------------------------------------------------ halo for propagation
    def exchange_halo(self, pivots_local: torch.Tensor, inv_local: torch.Tensor, kf_out_local: torch.Tensor):
        """pivots_local [Kl,S,D], inv_local [Kl,S], kf_out_local [3*Kl,S,D] (this rank's keyframes).
        Returns the same three with ONE extra leading keyframe slot = the previous rank's last
        keyframe (unset and unread on rank 0: global chunk 0 matches a single keyframe, 331-333)."""
        Kl = self.Kl
        if self.world == 1:
            return pivots_local, inv_local, kf_out_local      # no halo slot: ids are [c, c-1] directly
        _, S, D = pivots_local.shape
        # slot 0 = the left neighbour's last keyframe; on rank 0 it stays unset and is never read
        # (global chunk 0 matches keyframe 0 alone)
        piv = torch.empty(Kl + 1, S, D, dtype=pivots_local.dtype, device=pivots_local.device)
        inv = torch.empty(Kl + 1, S, dtype=inv_local.dtype, device=inv_local.device)
        kfo = torch.empty(3, Kl + 1, S, D, dtype=kf_out_local.dtype, device=kf_out_local.device)
        kf3 = kf_out_local.view(3, Kl, S, D)
        piv[1:].copy_(pivots_local)
        inv[1:].copy_(inv_local)
        kfo[:, 1:].copy_(kf3)
        # one grouped point-to-point exchange, no staging copies: every message is a contiguous view
        # (the attention output travels as one message per branch)
        opsl = []
        if self.rank + 1 < self.world:
            peer = self._peer(self.rank + 1)
            opsl += [dist.P2POp(dist.isend, t, peer, self.group)
                     for t in (pivots_local[-1], inv_local[-1], kf3[0, -1], kf3[1, -1], kf3[2, -1])]
        if self.rank > 0:
            peer = self._peer(self.rank - 1)
            opsl += [dist.P2POp(dist.irecv, t, peer, self.group)
                     for t in (piv[0], inv[0], kfo[0, 0], kfo[1, 0], kfo[2, 0])]
        for req in (dist.batch_isend_irecv(opsl) if opsl else []):
            req.wait()
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
        blend_dtype = out_dtype_two if len(ids) == 2 else kf_out_ext.dtype
        out_dtype = torch.promote_types(blend_dtype, residual.dtype) if residual is not None else blend_dtype
        return ops.propagate(tgt, piv_ext, inv_ext, ids, kf_out_ext, w if len(ids) == 2 else None, n, residual,
                             out_dtype)

    def propagate_all(self, tgt_all: torch.Tensor, residual_all: torch.Tensor, piv_ext, inv_ext, kf_out_ext,
                      w: torch.Tensor, n: int, out_dtype: torch.dtype = torch.float32):
        """ALL local chunks in one call (tf_nn_gather_blend_chunks): tgt_all [Kl*n*S, D] chunk-major, residual_all
        [3*Kl*n, S, D] (frames chunk-major inside each branch).  Same results as Kl calls of `propagate`, bit for
        bit; the one-keyframe chunk 0 of the video (rank 0) is rounded to the dtype its own call would produce."""
        o = 1 if self.world > 1 else 0
        return ops.propagate_chunks(tgt_all, piv_ext, inv_ext, kf_out_ext, w, n, self.Kl, o, self.kf0 == 0,
                                    residual_all, out_dtype)
