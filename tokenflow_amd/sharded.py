"""Frame-sharded multi-GPU execution of the hot path (one process per GPU).

The reference is single-process (no parallelism of any kind); this is new.
Rank r of W owns the contiguous chunks [r*C/W, (r+1)*C/W) of the video and the keyframes
drawn from them (K = C keyframes, one per chunk, run_tokenflow_pnp.py:224).  Per block there
are two exchange steps, both through torch.distributed (backend "nccl" = RCCL over xGMI on
MI355X; "gloo" in the CPU tests) -- or, with `FrameShard(K, comm=HipComm(...))`, through the library's own C-ABI
exchange entry points (tokenflow_amd/comm.py: RCCL without torch.distributed on the data path; the calls run on a
side stream ordered against the compute stream with events):

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
    "bank": ONE all-gather of the key/value slabs of the local keyframes (6 slabs of [K/W,S,D]
        per rank; 4 with injection), packed by one launch and gathered straight into the
        [K,slabs,S,D] buffer that ONE `ops.ext_attn_views(q_local, k_bank, v_bank, q_frame0=...)`
        call reads in place, computing only the local keyframes' queries.  A rank receives
        6*(W-1) local slabs -- several times the "heads" volume at W = 8 -- for one collective,
        one pack and one attention call per block instead of two collectives, pack, unpack and two
        calls: chosen per block (`auto_mode`) where the exchange is latency-bound (the mid block),
        and whenever the heads do not divide over the ranks.

 2. propagation passes -- chunk c needs keyframes c and c-1 (331-333): the first local chunk's
    left neighbour lives on rank r-1, so each rank sends its LAST keyframe's pivot features,
    inverse norms and attention output to rank r+1: one grouped point-to-point exchange per block
    (`pivotal_block`, the in-place form: the block's propagation state lives in halo-extended buffers
    from `ext_alloc` that the producers write directly) or two (`halo_start` before the attention,
    `halo_finish` after it), on a communicator / process group and side stream of its own when the
    host provides one (`halo_comm` / `halo_group`).

Three hosts run this sequence with identical results: `FrameShard` over torch.distributed, `FrameShard`
over the library's exchange entry points (`comm=`), and `NativeShard` (bottom of this file), whose
pivotal pass of a block is ONE library call (tf_rank_pivotal, csrc/rank_exec.hip).  The hook API
reaches them through `tokenflow_amd.hooks.register_frame_shard`.

Work is partitioned, not re-associated: every output element is produced by exactly the same
kernel arithmetic as on one GPU IN THE BIT-STABLE MODE (the attention of one (query, head) visits the
K frames in the same order, whoever computes it).  That is the default: `FrameShard` asks the attention
for its one-pass form (TF_ATTN_NO_SPLIT), in which kernel choice and in-workgroup key split are functions
of the shape alone; a sharded run then equals a single-GPU run with TOKENFLOW_ATTN_NO_SPLIT=1 (and a
world-1 `FrameShard`) bit for bit.  The single-GPU DEFAULT mode is free to choose per grid (large grids of
<= 256-token frames run without the key split: cfg2 level 2, +26 % faster there, profiles/r05_attn_nosplit_ab.txt)
and agrees with the sharded results within the attention's parity bound, like any two correct launches.
`FrameShard(..., attn_split=True)` (or TOKENFLOW_SHARD_ATTN_SPLIT=1) lets the small grid of a rank split
the bank over extra workgroups and merge: faster, held to the ORACLE's bound.
"""
import ctypes
import os
from typing import Optional

import torch
import torch.distributed as dist

from . import ops


class _Done:
    def wait(self):
        return True


class _EventRing:
    """A few reusable events for ordering the exchange stream against the compute stream.  Creating a
    `torch.cuda.Event` costs tens of microseconds on this stack (`Stream.wait_stream` creates one per call: it was
    half of a rank's host time per block, profiles/r03_rank_step_v1.txt); recording an existing one costs ~1 us.  A
    wait captures the record that is current when it is issued, so an event may be re-recorded while earlier waits on
    it are still queued; a handle that is waited on only after its event has been re-recorded (the ring wrapped)
    waits for that LATER point of the same in-order stream -- still correct, never early.  That argument needs every
    event of a ring to be recorded on ONE stream: there is one ring per recording stream (the compute stream's "start"
    events, each side stream's "done" events), never a shared one.  256 events cover more than two passes over the 16
    blocks (<= 6 per block), longer than any handle lives (a halo is waited for in the propagation pass that follows
    its pivotal pass)."""

    def __init__(self, n: int = 256):
        self._ev = [torch.cuda.Event() for _ in range(n)]
        self._i = 0

    def next(self):
        e = self._ev[self._i]
        self._i = (self._i + 1) % len(self._ev)
        return e


class _StreamWork:
    """Completion handle of exchanges issued on the side stream: wait() orders the CURRENT stream behind them
    (no host blocking -- the semantics of a c10d work object on the NCCL backend)."""

    def __init__(self, event, device):
        self.event, self.device = event, device

    def wait(self):
        # current_stream(device): the device-less form resolves the device through torch.cuda.is_available(),
        # ~30 us per call on this stack -- it was two thirds of a rank's host time per block
        torch.cuda.current_stream(self.device).wait_event(self.event)
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

    def __init__(self, K: int, group: Optional[dist.ProcessGroup] = None, comm=None,
                 attn_split: Optional[bool] = None, halo_group: Optional[dist.ProcessGroup] = None, halo_comm=None):
        # attn_split: let the attention split a rank's small grid over extra workgroups and merge (faster: -15..40 %
        # on a rank's attention at 8 GPUs; results then agree with the single-GPU ones within the
        # output rounding).  Default False: one pass per bank problem, arithmetic independent of the grid, sharded
        # results equal to single-GPU results bit for bit.  None reads TOKENFLOW_SHARD_ATTN_SPLIT.
        # A WORLD-1 shard follows the same rule (it equals a world-W shard bit for bit), which means it runs the
        # bit-stable mode and loses the single-GPU default's per-grid kernel choice (+26 % at cfg2 level 2): pass
        # attn_split=True when a world-1 shard is registered for speed, not for bit stability.
        if attn_split is None:
            attn_split = os.environ.get("TOKENFLOW_SHARD_ATTN_SPLIT", "0") not in ("", "0")
        self.attn_split = bool(attn_split)
        # TOKENFLOW_SHARD_SRC_AUX=1: the source-branch attention of the head re-sharding on an auxiliary stream, beside
        # the bank attention (in-place form only).  Measured on a rank of 8 (profiles/r03_rank_step_v3.txt): no gain at
        # the coarse levels and -4 % at level 0 (its workgroups take slots from the chip-filling bank grid) -> off.
        self.src_aux = os.environ.get("TOKENFLOW_SHARD_SRC_AUX", "0") not in ("", "0")
        self.group = group
        self.comm = comm                       # tokenflow_amd.comm.HipComm: exchanges through the C ABI instead
        self._cs = None                        # its side stream
        # The neighbour halo on a communicator (and side stream) of its own: collectives of ONE RCCL communicator run
        # in issue order, so on the pivotal pass's communicator the ~10 MB halo message of block b (one xGMI link) would
        # sit in front of block b+1's first all-to-all.  None = share the pivotal pass's.
        self.halo_group, self.halo_comm = halo_group, halo_comm
        self._hs = None
        if comm is not None:
            self.world, self.rank = comm.world, comm.rank
        else:
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
        self._bufs = {}                        # exchange buffers by (tag, shape, dtype, device): reused block after block

    def _buf(self, tag, shape, dtype, device):
        """Receive / send scratch of the exchanges.  Every use is complete (the collective waited for, the consumer
        kernel enqueued on the same stream) before the next block touches the buffer again, so one buffer per shape
        serves all blocks -- and saves an allocator round trip per buffer and block on the host."""
        key = (tag, tuple(shape), dtype, device)
        b = getattr(self, "_bufs", None)
        if b is None:
            b = self._bufs = {}
        t = b.get(key)
        if t is None:
            t = b[key] = torch.empty(shape, dtype=dtype, device=device)
        return t

    def _side(self, tensors, fn, halo: bool = False):
        """HipComm path: run `fn(stream)` (RCCL calls through the C ABI, asynchronous on the stream they are handed)
        on the exchange stream, ordered after everything already enqueued on the compute stream; returns the handle
        to wait on.  No stream context manager and no device-less torch.cuda query on this path: both cost tens of
        microseconds of host time per call, several times per block."""
        if not tensors[0].is_cuda:             # host tensors (the CPU tests' stand-in comm): no streams to order
            fn(None)
            return _Done()
        dev = tensors[0].device
        cur = torch.cuda.current_stream(dev)
        if getattr(self, "_cs", None) is None:
            self._cs = torch.cuda.Stream(device=dev)
            self._ring = _EventRing()          # "start" events: recorded on the compute stream only
            self._cs_done = _EventRing()       # "done" events of the exchange stream
        side, done_ring = self._cs, self._cs_done
        if halo and getattr(self, "halo_comm", None) is not None:
            if getattr(self, "_hs", None) is None:
                self._hs = torch.cuda.Stream(device=dev)
                self._hs_done = _EventRing()   # "done" events of the halo stream
            side, done_ring = self._hs, self._hs_done
        e = self._ring.next()
        e.record(cur)
        side.wait_event(e)
        fn(side.cuda_stream)
        for t in tensors:
            t.record_stream(side)              # allocator: not reusable before the exchange stream is done with it
        done = done_ring.next()
        done.record(side)
        return _StreamWork(done, dev)

    def _on_aux(self, dev, tensors, fn):
        """Run `fn(raw stream)` on this rank's auxiliary compute stream, ordered after everything already enqueued on
        the current stream; returns the handle whose wait() orders the current stream behind it."""
        cur = torch.cuda.current_stream(dev)
        if getattr(self, "_aux", None) is None:
            self._aux = torch.cuda.Stream(device=dev)
            self._aux_ring = _EventRing(64)    # recorded on the compute stream
            self._aux_done = _EventRing(64)    # recorded on the auxiliary stream
        e = self._aux_ring.next()
        e.record(cur)
        self._aux.wait_event(e)
        fn(self._aux.cuda_stream)
        for t in tensors:
            t.record_stream(self._aux)
        done = self._aux_done.next()
        done.record(self._aux)
        return _StreamWork(done, dev)

    def _a2a(self, recv: torch.Tensor, send: torch.Tensor, out_rows=None, in_rows=None, async_op: bool = False):
        """Row all-to-all over dim 0: out_rows[p] rows arrive from peer p, in_rows[p] rows go to peer p (None = equal)."""
        comm = getattr(self, "comm", None)
        if comm is None:
            return _all_to_all(recv, send, self.group, out_rows, in_rows, async_op=async_op)
        work = self._side([recv, send], lambda st: comm.all_to_all_rows(send, recv, in_rows, out_rows, stream=st))
        if async_op:
            return work
        work.wait()
        return None

    # ------------------------------------------------------------------ pivotal pass
    def _bank_gather(self, k_local: torch.Tensor, v_local: torch.Tensor, inject: bool):
        """ONE collective per block: the slabs of the local keyframes that the attention reads across frames --
        without injection k and v of all three branches ([k0,k1,k2,v0,v1,v2]: the source slabs ride along so that the
        gathered buffer is the whole [3,K,S,D] bank of ONE attention call), with injection [k0,v0,v1,v2] -- are packed
        frame-major by one `tf_head_pack` launch (W = 1: a plain slab pack), gathered straight into
        [K (global frame order), slabs, S, D], and read there in place through strided views.  Runs of different
        lengths (K % W != 0) use the row form of the all-gather (no padding, no compaction copies).
        Returns (k view [3 or 1, K, S, D], v view [3, K, S, D])."""
        B, S, D = k_local.shape
        Kl, K, W = self.Kl, self.K, self.world
        dev, dt = k_local.device, k_local.dtype

        def frames(t):     # [3*Kl, S, D] (token stride free) -> [3, Kl, S, D] view
            if t.stride(2) != 1 or t.stride(0) != S * t.stride(1):
                t = t.contiguous()
            return t.view(3, Kl, S, D) if t.is_contiguous() else t.unflatten(0, (3, Kl))
        k3, v3 = frames(k_local), frames(v_local)
        if k3.stride(2) != v3.stride(2):
            k3, v3 = k3.contiguous(), v3.contiguous()
        slabs = [k3[0], v3[0], v3[1], v3[2]] if inject else [k3[0], k3[1], k3[2], v3[0], v3[1], v3[2]]
        ns = len(slabs)
        send = ops.head_pack(slabs, 1, out=self._buf("bank_send", (1, Kl, ns, S, D), dt, dev)).view(Kl, ns * S * D)
        recv = self._buf("bank_recv", (K, ns * S * D), dt, dev)
        comm = getattr(self, "comm", None)
        if self.even:
            if comm is None:
                if send.is_cuda and dist.get_backend(self.group) == "gloo":     # development boxes: host staging
                    host = torch.empty(recv.shape, dtype=dt)
                    dist.all_gather_into_tensor(host, send.cpu(), group=self.group)
                    recv.copy_(host)
                else:
                    dist.all_gather_into_tensor(recv, send, group=self.group)
            else:
                self._side([recv, send], lambda st: comm.allgather(send, recv, stream=st)).wait()
        elif comm is None:       # one grouped point-to-point exchange: my rows to every peer, theirs into place
            staged = send.is_cuda and dist.get_backend(self.group) == "gloo"
            src = send.cpu() if staged else send
            dst = torch.empty(recv.shape, dtype=dt) if staged else recv
            opsl = []
            for r in range(W):
                if r == self.rank:
                    continue
                peer = self._peer(r)
                opsl.append(dist.P2POp(dist.isend, src, peer, self.group))
                opsl.append(dist.P2POp(dist.irecv, dst[self.offsets[r]:self.offsets[r] + self.counts[r]], peer,
                                       self.group))
            reqs = dist.batch_isend_irecv(opsl)
            dst[self.kf0:self.kf0 + Kl].copy_(src)
            for r in reqs:
                r.wait()
            if staged:
                recv.copy_(dst)
        else:
            self._side([recv, send], lambda st: comm.allgather_rows(send, recv, self.counts, stream=st)).wait()
        rp = recv.view(K, ns, S, D).permute(1, 0, 2, 3)        # [ns, K, S, D] views: frame stride ns*S*D
        return (rp[0:1], rp[1:4]) if inject else (rp[0:3], rp[3:6])

    def auto_mode(self, heads: int, S: int) -> str:
        """Exchange pattern of the pivotal pass for one block.  "heads" moves the least data (under 8 local slabs
        per rank whatever W) and is the choice wherever the block has real work; it needs heads % W == 0.  "bank"
        is ONE collective and ONE attention call instead of two collectives, a pack, an unpack and two calls: the
        choice where a block is a few tens of microseconds of work (S <= 64: the mid block) and the exchange is
        latency-, not volume-bound -- and the only one when the heads do not divide over the ranks."""
        if heads % self.world:
            return "bank"
        return "bank" if S <= 64 else "heads"

    def pivotal_attention(self, q_local, k_local, v_local, heads: int, scale: float, inject: bool,
                          mode: Optional[str] = None, out4: Optional[torch.Tensor] = None):
        """Extended attention for the local keyframes against all K keyframes -> [3*Kl,S,D].
        mode: "heads" | "bank" | None (= `auto_mode`, chosen per block).
        out4: a [3,Kl,S,D] view (dense frames, free branch stride) the result is written into in place -- the
        keyframe slots 1.. of a halo-extended buffer (`ext_alloc`); returned as is."""
        if self.world == 1:
            # the same mode as a rank of a larger world: a world-1 shard equals a world-W shard bit for bit by default.
            # COST: with attn_split unset this is the bit-stable mode (no_split), which gives up the single-GPU default's
            # per-grid kernel choice (cfg2 level 2: 124 against 98 us per block, profiles/r05_attn_nosplit_ab.txt).  A
            # single-GPU user who registers a world-1 shard for speed passes attn_split=True.
            ns = not self.attn_split
            if out4 is None:
                return ops.ext_attn(q_local, k_local, v_local, heads, scale, inject, no_split=ns)
            return ops.ext_attn(q_local, k_local, v_local, heads, scale, inject, out=out4.view(q_local.shape), no_split=ns)
        if mode is None:
            mode = self.auto_mode(heads, q_local.shape[1])
        if mode == "heads":
            return self._pivotal_heads(q_local, k_local, v_local, heads, scale, inject, out4)
        return self._pivotal_bank(q_local, k_local, v_local, heads, scale, inject, out4)

    def _pivotal_bank(self, q_local, k_local, v_local, heads: int, scale: float, inject: bool, out4=None):
        """One gather (`_bank_gather`), one attention call on the gathered buffer in place; q keeps its own layout
        (its token stride is independent of the bank's)."""
        B, S, D = q_local.shape
        Kl = self.Kl
        kv, vv = self._bank_gather(k_local, v_local, inject)
        q = q_local
        if q.stride(2) != 1 or q.stride(0) != S * q.stride(1):
            q = q.contiguous()
        q4 = q.view(3, Kl, S, D) if q.is_contiguous() else q.unflatten(0, (3, Kl))
        out = torch.empty(3, Kl, S, D, dtype=q.dtype, device=q.device) if out4 is None else out4
        ops.ext_attn_views(q4, kv, vv, out, heads, scale, inject, "all", q_frame0=self.kf0,
                           no_split=not self.attn_split)
        return out.view(3 * Kl, S, D) if out4 is None else out4

    def _pivotal_heads(self, q_local, k_local, v_local, heads: int, scale: float, inject: bool, out4=None):
        """Frames <-> heads re-sharding.  Launches per block on this rank: ONE pack kernel, the source-branch
        attention (overlaps the first all-to-all), the bank attention reading the received buffer IN PLACE and
        writing the second all-to-all's send buffer IN PLACE (strided views, no re-layout copies), ONE unpack
        kernel.  Buffers are laid out for the collectives: what a rank sends to peer w is one contiguous
        [Kl, slabs, S, D/W] piece, so what arrives is [K (global frame order), slabs, S, D/W] whatever the run
        lengths."""
        W, Kl, K = self.world, self.Kl, self.K
        B, S, D = q_local.shape
        if heads % W:
            raise ValueError(f"{heads} heads do not divide over {W} ranks")
        hd, dev, dt = D // W, q_local.device, q_local.dtype
        src_done = None

        def frames(t):     # [3*Kl, S, D] (token stride free) -> [3, Kl, S, D] view
            if t.stride(2) != 1 or t.stride(0) != S * t.stride(1):
                t = t.contiguous()
            return t.view(3, Kl, S, D) if t.is_contiguous() else t.unflatten(0, (3, Kl))
        q3, k3, v3 = frames(q_local), frames(k_local), frames(v_local)
        if not (q3.stride(2) == k3.stride(2) == v3.stride(2)):      # one token stride for the pack kernel
            q3, k3, v3 = (t.contiguous() for t in (q3, k3, v3))
        even = self.even
        # ---- pack: head group w of every slab the bank branches read, frame-major, slab order [q.., k.., v..]
        if inject:       # source q, k (what uncond and cond use, 124-130) and the two value banks
            slabs = [q3[0], k3[0], v3[1], v3[2]]
        else:
            slabs = [q3[1], q3[2], k3[1], k3[2], v3[1], v3[2]]
        ns = len(slabs)
        send = ops.head_pack(slabs, W, out=self._buf("send", (W, Kl, ns, S, hd), dt, dev))
        recv = self._buf("recv", (K, ns, S, hd), dt, dev)
        work = self._a2a(recv.view(K, -1), send.view(W * Kl, -1),
                         None if even else self.counts, None if even else [Kl] * W, async_op=True)
        # ---- source branch: own-frame keys, all heads, stays local (overlaps the exchange)
        if out4 is None:
            out = torch.empty(3, Kl, S, D, dtype=dt, device=dev)
            ops.ext_attn(q_local, k_local, v_local, heads, scale, inject, out=out.view(3 * Kl, S, D), part="source",
                         no_split=not self.attn_split)
        else:            # straight into the caller's (strided) slots: the strided entry point
            out = out4
            if q3.is_cuda and getattr(self, "src_aux", False):
                # on its own stream: the source problems of a rank are a fraction of a chip-filling grid (one frame's
                # queries: 32..128 workgroups), so they run BESIDE the exchange and the bank attention that follows it,
                # not in front of it; joined before the block's results leave (below)
                src_done = self._on_aux(dev, [q3, k3, v3, out], lambda st: ops.ext_attn_views(
                    q3, k3, v3, out, heads, scale, inject, "source", no_split=not self.attn_split, stream=st))
            else:
                ops.ext_attn_views(q3, k3, v3, out, heads, scale, inject, "source", no_split=not self.attn_split)
        work.wait()
        # ---- bank branches on this rank's head group, all K frames: read `recv`, write `send2`, both in place
        rp = recv.permute(1, 0, 2, 3)                                   # [ns, K, S, hd] view
        send2 = self._buf("send2", (K, 2, S, hd), dt, dev)            # [frame][uncond|cond]: rows of rank w's run -> w
        o4 = send2.permute(1, 0, 2, 3)                                  # [2, K, S, hd] view = branches 1, 2
        if inject:
            ops.ext_attn_views(rp[0:1], rp[1:2], rp[2:4], o4, heads // W, scale, True, "bank", branch0=(0, 0, 1, 1),
                               no_split=not self.attn_split)
        else:
            ops.ext_attn_views(rp[0:2], rp[2:4], rp[4:6], o4, heads // W, scale, False, "bank", branch0=(1, 1, 1, 1),
                               no_split=not self.attn_split)
        # ---- outputs back to the frame owners
        recv2 = self._buf("recv2", (W, Kl, 2, S, hd), dt, dev)        # [head group][my frames][uncond|cond]
        self._a2a(recv2.view(W * Kl, -1), send2.view(K, -1),
                  None if even else [Kl] * W, None if even else self.counts)
        ops.head_unpack(recv2, [out[1], out[2]])
        if src_done is not None:
            src_done.wait()
        return out.view(3 * Kl, S, D) if out4 is None else out4

    # ------------------------------------------------------------------ pivotal pass of one block, in place
    def ext_alloc(self, S: int, D: int, dtype: torch.dtype, device):
        """Per-block state of the propagation, halo slot included: (pivots [Kl+o,S,D], inverse norms [Kl+o,S] fp32,
        cached attention output [3,Kl+o,S,D]) with o = 1 when there is a left neighbour to hear from (world > 1), else
        0.  The producers write the local keyframes straight into slots o.. (norm1 -> pivots and inverse norms, the
        attention -> its output): no staging copy on either side of the halo exchange."""
        o = 1 if self.world > 1 else 0
        Kl = self.Kl
        return (torch.empty(Kl + o, S, D, dtype=dtype, device=device),
                torch.empty(Kl + o, S, dtype=torch.float32, device=device),
                torch.empty(3, Kl + o, S, D, dtype=dtype, device=device))

    def pivotal_block(self, q_local, k_local, v_local, heads: int, scale: float, inject: bool, ext,
                      mode: Optional[str] = None, inv_norm: bool = False):
        """The pivotal pass of one block on this rank, for the reference's call order (one pivotal pass over all
        blocks, then the chunk passes): `ext` = `ext_alloc(...)` whose pivot / inverse-norm slots o.. the caller has
        filled.  The attention writes its output into ext's slots o.. in place, then ONE grouped neighbour exchange
        carries the last local keyframe's pivots, inverse norms and attention output to slot 0 of rank r+1 (it has the
        rest of the pivotal pass to arrive; nothing waits for it before the propagation).
        inv_norm=True: the inverse-norm slots o.. are filled HERE from the pivot slots (callers whose norm1 is not the
        fused LayerNorm producer; the native executor does it inside its pack launch).
        Returns (pivots ext, inverse norms ext, attention output ext [3(Kl+o),S,D], pending requests) -- the
        arguments of `propagate_all(..., halo_reqs=)`."""
        piv, inv, kfo = ext
        o = 1 if self.world > 1 else 0
        Kl = self.Kl
        S, D = piv.shape[1:]
        if inv_norm:
            ops.pivot_inv_norm(piv[o:], out=inv[o:])
        self.pivotal_attention(q_local, k_local, v_local, heads, scale, inject, mode=mode, out4=kfo[:, o:])
        reqs = []
        if self.world > 1:
            reqs = self._p2p([piv[-1], inv[-1], kfo[0, -1], kfo[1, -1], kfo[2, -1]],
                             [piv[0], inv[0], kfo[0, 0], kfo[1, 0], kfo[2, 0]])
        return piv, inv, kfo.view(3 * (Kl + o), S, D), reqs

    def halo_block(self, piv_ext: torch.Tensor, inv_ext: torch.Tensor, kfo_ext: torch.Tensor):
        """ONE grouped neighbour exchange on halo-extended per-block state the producers have written in place
        (`ext_alloc` layout: slot 0 = the left neighbour's last keyframe): the last local keyframe's pivots, inverse
        norms and attention output go to rank r+1, the neighbour's arrive in slot 0.  Returns the pending requests
        (`halo_wait`).  What `pivotal_block` issues behind its attention; the hook path calls it behind `to_out`."""
        if self.world == 1:
            return []
        return self._p2p([piv_ext[-1], inv_ext[-1], kfo_ext[0, -1], kfo_ext[1, -1], kfo_ext[2, -1]],
                         [piv_ext[0], inv_ext[0], kfo_ext[0, 0], kfo_ext[1, 0], kfo_ext[2, 0]])

    # ------------------------------------------------------------------ halo for propagation
    def exchange_halo(self, pivots_local: torch.Tensor, inv_local: torch.Tensor, kf_out_local: torch.Tensor):
        """pivots_local [Kl,S,D], inv_local [Kl,S], kf_out_local [3*Kl,S,D] (this rank's keyframes).
        Returns the same three with ONE extra leading keyframe slot = the previous rank's last
        keyframe (unset and unread on rank 0: global chunk 0 matches a single keyframe, 331-333)."""
        h = self.halo_start(pivots_local, inv_local)
        return self.halo_finish(h, kf_out_local)

    def halo_start(self, pivots_local: torch.Tensor, inv_local: torch.Tensor):
        """First half of the halo exchange: the pivot features and inverse norms of the last local keyframe go to
        rank r+1.  They exist as soon as norm1 has run, BEFORE the attention, so this is issued first and travels
        under the attention; `halo_finish` then sends the attention output."""
        if self.world == 1:
            return (pivots_local, inv_local, None)
        Kl = self.Kl
        _, S, D = pivots_local.shape
        # slot 0 = the left neighbour's last keyframe; on rank 0 it stays unset and is never read
        piv = torch.empty(Kl + 1, S, D, dtype=pivots_local.dtype, device=pivots_local.device)
        inv = torch.empty(Kl + 1, S, dtype=inv_local.dtype, device=inv_local.device)
        piv[1:].copy_(pivots_local)
        inv[1:].copy_(inv_local)
        reqs = self._p2p([pivots_local[-1], inv_local[-1]], [piv[0], inv[0]])
        return (piv, inv, reqs)

    def halo_finish(self, handle, kf_out_local: torch.Tensor, wait: bool = True):
        """Second half: the attention output of the last local keyframe to rank r+1.  wait=False returns the
        pending requests as a 4th element (call `halo_wait`): the propagation of the local chunks 1.. does not read
        the halo slot and can be issued before."""
        piv, inv, reqs = handle
        if self.world == 1:
            return (piv, inv, kf_out_local) if wait else (piv, inv, kf_out_local, [])
        Kl = self.Kl
        S, D = kf_out_local.shape[1:]
        kf3 = kf_out_local.view(3, Kl, S, D)
        kfo = torch.empty(3, Kl + 1, S, D, dtype=kf_out_local.dtype, device=kf_out_local.device)
        kfo[:, 1:].copy_(kf3)
        # no staging copies: every message is a contiguous view (the attention output travels as one message per branch)
        reqs = list(reqs) + self._p2p([kf3[0, -1], kf3[1, -1], kf3[2, -1]], [kfo[0, 0], kfo[1, 0], kfo[2, 0]])
        res = (piv, inv, kfo.view(3 * (Kl + 1), S, D))
        if not wait:
            return res + (reqs,)
        self.halo_wait(reqs)
        return res

    @staticmethod
    def halo_wait(reqs):
        for r in reqs:
            r.wait()

    def _p2p(self, send_tensors, recv_tensors):
        """One grouped point-to-point exchange: `send_tensors` to rank r+1, `recv_tensors` from rank r-1."""
        comm = getattr(self, "halo_comm", None) or getattr(self, "comm", None)
        if comm is not None:
            to = self.rank + 1 if self.rank + 1 < self.world else -1
            frm = self.rank - 1 if self.rank > 0 else -1
            if to < 0 and frm < 0:
                return []
            send_tensors = [t.contiguous() for t in send_tensors]

            def go(st):     # one grouped call per element type (the C entry point takes one dtype per call)
                for dt in dict.fromkeys(t.dtype for t in list(send_tensors) + list(recv_tensors)):
                    comm.sendrecv([t for t in send_tensors if t.dtype == dt], to,
                                  [t for t in recv_tensors if t.dtype == dt], frm, stream=st)
            return [self._side(list(send_tensors) + list(recv_tensors), go, halo=True)]
        opsl = []
        grp = getattr(self, "halo_group", None) or self.group
        if self.rank + 1 < self.world:
            peer = self._peer(self.rank + 1)
            opsl += [dist.P2POp(dist.isend, t, peer, grp) for t in send_tensors]
        if self.rank > 0:
            peer = self._peer(self.rank - 1)
            opsl += [dist.P2POp(dist.irecv, t, peer, grp) for t in recv_tensors]
        return dist.batch_isend_irecv(opsl) if opsl else []

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
                      w: torch.Tensor, n: int, out_dtype: torch.dtype = torch.float32, halo_reqs=None):
        """ALL local chunks (tf_nn_gather_blend_chunks): tgt_all [Kl*n*S, D] chunk-major, residual_all
        [3*Kl*n, S, D] (frames chunk-major inside each branch).  Same results as Kl calls of `propagate`, bit for
        bit; the one-keyframe chunk 0 of the video (rank 0) is rounded to the dtype its own call would produce.
        halo_reqs (from `halo_finish(wait=False)`): the local chunks 1.. read no halo slot and are issued FIRST,
        the first local chunk after the halo has landed; returns (first chunk [3n,S,D], rest [3(Kl-1)n,S,D] or None)."""
        o = 1 if self.world > 1 else 0
        if halo_reqs is None:
            return ops.propagate_chunks(tgt_all, piv_ext, inv_ext, kf_out_ext, w, n, self.Kl, o, self.kf0 == 0,
                                        residual_all, out_dtype)
        Kl, S, D = self.Kl, piv_ext.shape[1], piv_ext.shape[2]
        nS = n * S
        res = residual_all.view(3, Kl, n, S, D)
        rest = None
        if Kl > 1:
            rest = ops.propagate_chunks(tgt_all[nS:], piv_ext, inv_ext, kf_out_ext, w, n, Kl - 1, o + 1, False,
                                        res[:, 1:].reshape(3 * (Kl - 1) * n, S, D), out_dtype)
        self.halo_wait(halo_reqs)
        first = self.propagate(0, tgt_all[:nS], res[:, 0].reshape(3 * n, S, D), piv_ext, inv_ext, kf_out_ext, w, n)
        return first, rest


class _SlotWait:
    """Pending neighbour halo of one block of a `NativeShard`: wait() orders the CURRENT stream behind it."""

    def __init__(self, shard, slot, device):
        self.shard, self.slot, self.device = shard, slot, device

    def wait(self):
        from . import _lib
        rc = _lib.load().tf_rank_halo_wait(self.shard._rk, self.slot, torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            _lib.check(rc, "tf_rank_halo_wait")
        return True


class NativeShard(FrameShard):
    """`FrameShard` whose pivotal pass of a block is ONE call into the library (tf_rank_pivotal, csrc/rank_exec.hip):
    pack, exchanges, source and bank attention, unpack and the neighbour halo are issued by native code on the
    library's own streams -- the host cost of a block drops from a dozen Python-level op calls (200-300 us at a rank of
    8, more than the GPU needs for the block at the coarse levels) to one foreign call.  Same results as `FrameShard`
    bit for bit (same kernels, same buffers' layouts).  `comm` / `halo_comm`: `tokenflow_amd.comm.HipComm` objects (RCCL
    through the C ABI; a second communicator lets the halo of one block travel beside the exchanges of the next).
    Only the in-place two-pass API (`ext_alloc`, `pivotal_block`, `propagate_all(..., halo_reqs=)`) goes native; the
    other methods are `FrameShard`'s own on the same communicator."""

    def __init__(self, K: int, comm, halo_comm=None, attn_split: Optional[bool] = None):
        super().__init__(K, comm=comm, attn_split=attn_split, halo_comm=halo_comm)
        from . import _lib
        lib = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(lib.tf_rank_create(comm._h if comm is not None else None,
                                      halo_comm._h if halo_comm is not None else None, K, ctypes.byref(h)),
                   "tf_rank_create")
        self._rk, self._halo_comm = h, halo_comm
        assert lib.tf_rank_local_keyframes(h) == self.Kl and lib.tf_rank_first_keyframe(h) == self.kf0
        self._slot = 0
        self._nws = {}

    def close(self):
        rk, self._rk = getattr(self, "_rk", None), None
        if rk is not None:
            from . import _lib
            _lib.load().tf_rank_destroy(rk)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def pivotal_attention(self, q_local, k_local, v_local, heads: int, scale: float, inject: bool,
                          mode: Optional[str] = None, out4: Optional[torch.Tensor] = None):
        """`FrameShard.pivotal_attention` as one library call (tf_rank_pivotal with TF_RANK_NO_HALO): what the hook path
        calls from `attn1` (its cached attention output is the to_out-projected one, so the halo stays the block's)."""
        if self.world == 1 or out4 is not None or not q_local.is_cuda or q_local.dtype not in (torch.bfloat16,
                                                                                                torch.float16):
            return super().pivotal_attention(q_local, k_local, v_local, heads, scale, inject, mode=mode, out4=out4)
        B, S, D = q_local.shape
        out = torch.empty(3, self.Kl, S, D, dtype=q_local.dtype, device=q_local.device)
        self._native(q_local, k_local, v_local, heads, scale, inject, mode, None, None, out, no_halo=True)
        return out.view(B, S, D)

    def pivotal_block(self, q_local, k_local, v_local, heads: int, scale: float, inject: bool, ext,
                      mode: Optional[str] = None, inv_norm: bool = False):
        piv, inv, kfo = ext
        B, S, D = q_local.shape
        Kl = self.Kl
        o = 1 if self.world > 1 else 0
        slot = self._native(q_local, k_local, v_local, heads, scale, inject, mode, piv, inv, kfo, no_halo=False,
                            inv_norm=inv_norm)
        reqs = [_SlotWait(self, slot, q_local.device)] if self.world > 1 else []
        return piv, inv, kfo.view(3 * (Kl + o), S, D), reqs

    def _native(self, q_local, k_local, v_local, heads, scale, inject, mode, piv, inv, kfo, no_halo, inv_norm=False):
        from . import _lib
        lib = _lib.load()
        B, S, D = q_local.shape
        Kl = self.Kl
        if mode is None:
            mode = self.auto_mode(heads, S)
        dt = ops._DT.get(q_local.dtype)
        if dt is None or dt == _lib.TF_F32 or not (q_local.is_cuda and kfo.is_contiguous()
                                                    and (no_halo or (piv.is_contiguous() and inv.is_contiguous()))):
            raise TypeError("NativeShard: 16-bit GPU tensors, contiguous (halo-extended) buffers")

        def frames(t):     # [3*Kl, S, D] (token stride free) -> [3, Kl, S, D] view
            if t.stride(2) != 1 or t.stride(0) != S * t.stride(1):
                t = t.contiguous()
            return t.view(3, Kl, S, D) if t.is_contiguous() else t.unflatten(0, (3, Kl))
        q3, k3, v3 = frames(q_local), frames(k_local), frames(v_local)
        if k3.stride(2) != v3.stride(2) or (mode == "heads" and q3.stride(2) != k3.stride(2)):
            q3, k3, v3 = (t.contiguous() for t in (q3, k3, v3))
        strides = (ctypes.c_int64 * 8)(q3.stride(0), q3.stride(1), k3.stride(0), k3.stride(1), v3.stride(0), v3.stride(1),
                                       q3.stride(2), k3.stride(2))
        dh = D // heads
        key = (S, heads, dh, dt, q_local.device)
        ws = self._nws.get(key)
        if ws is None:
            nbytes = lib.tf_rank_pivotal_workspace_bytes(self._rk, S, heads, dh, dt)
            ws = self._nws[key] = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=q_local.device)
        flags = (_lib.TF_ATTN_INJECT if inject else 0) | (0 if self.attn_split else _lib.TF_ATTN_NO_SPLIT)
        if ops.FOLD_SCALE:
            flags |= _lib.TF_ATTN_FOLD_SCALE
        slot = self._slot
        if not no_halo:
            self._slot = (slot + 1) % _lib.TF_RANK_SLOTS
        m = ((_lib.TF_RANK_HEADS if mode == "heads" else _lib.TF_RANK_BANK) | (_lib.TF_RANK_NO_HALO if no_halo else 0)
             | (_lib.TF_RANK_INV_NORM if inv_norm else 0))
        rc = lib.tf_rank_pivotal(self._rk, q3.data_ptr(), k3.data_ptr(), v3.data_ptr(), strides,
                                 None if piv is None else piv.data_ptr(), None if inv is None else inv.data_ptr(),
                                 kfo.data_ptr(), S, heads, dh, float(scale), flags, dt, m, slot, ws.data_ptr(),
                                 ws.numel(), torch.cuda.current_stream(q_local.device).cuda_stream)
        if rc:
            _lib.check(rc, "tf_rank_pivotal")
        return slot
