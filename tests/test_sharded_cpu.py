"""Frame-sharded path on 2 and on 8 CPU processes (gloo): the exchange logic of
tokenflow_amd/sharded.py with the oracle-backed FakeOps standing in for the HIP ops.
Sharded results must equal the single-process results bit for bit (work is partitioned,
not re-associated).  World 8 runs BASELINE configs 3 and 5 at their own rank geometry
(toy token counts): K = 8 -> one keyframe and one chunk per rank, K = 25 -> runs of
4,3,3,3,3,3,3,3 with head counts that do not divide over the ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tokenflow_oracle as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _data(K, n, S, h, d, seed=0):
    g = torch.Generator().manual_seed(seed)
    D = h * d
    q, k, v = (torch.randn(3 * K, S, D, generator=g) for _ in range(3))
    piv = torch.randn(K, S, D, generator=g)
    kf_out = torch.randn(3 * K, S, D, generator=g)
    tgt = torch.randn(K, n * S, D, generator=g)            # per chunk
    res = torch.randn(K, 3 * n, S, D, generator=g)
    return q, k, v, piv, kf_out, tgt, res


class GlooComm:
    """Stand-in for tokenflow_amd.comm.HipComm with the same interface and argument meaning, carried by gloo: lets
    the CPU tests drive FrameShard's C-ABI-comm branch (argument order of the row all-to-all, the per-dtype grouping
    of the halo messages, the padded all-gather of uneven runs) with a world of two."""

    def __init__(self):
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allgather(self, local, bank, stream=None):
        assert bank.numel() == self.world * local.numel() and local.is_contiguous() and bank.is_contiguous()
        dist.all_gather_into_tensor(bank.view(-1), local.reshape(-1))
        return bank

    def allgather_rows(self, local, bank, rows, stream=None):
        assert local.is_contiguous() and bank.is_contiguous() and sum(rows) == bank.shape[0]
        assert local.shape[0] == rows[self.rank] and len(rows) == self.world
        parts = list(bank.split(list(rows)))
        opsl = []
        for p in range(self.world):
            if p == self.rank:
                parts[p].copy_(local)
            else:
                opsl += [dist.P2POp(dist.isend, local, p), dist.P2POp(dist.irecv, parts[p], p)]
        for r in dist.batch_isend_irecv(opsl):
            r.wait()
        return bank

    def all_to_all_rows(self, send, recv, send_rows=None, recv_rows=None, stream=None):
        assert send.is_contiguous() and recv.is_contiguous()
        if send_rows is not None:
            assert sum(send_rows) == send.shape[0] and sum(recv_rows) == recv.shape[0]
        dist.all_to_all_single(recv, send, recv_rows, send_rows)
        return recv

    def sendrecv(self, send, send_peer, recv, recv_peer, stream=None):
        ts = list(send if send_peer >= 0 else []) + list(recv if recv_peer >= 0 else [])
        if not ts:
            return
        assert len({t.dtype for t in ts}) == 1 and all(t.is_contiguous() for t in ts)   # one dtype per C call
        opsl = [dist.P2POp(dist.isend, t, send_peer) for t in (send if send_peer >= 0 else [])]
        opsl += [dist.P2POp(dist.irecv, t, recv_peer) for t in (recv if recv_peer >= 0 else [])]
        for r in dist.batch_isend_irecv(opsl):
            r.wait()


def _worker(rank, world, port, K, n, S, h, d, inject, mode, ret, use_comm=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.fake_ops import FakeOps
        from tokenflow_amd import sharded
        fake = FakeOps()
        sharded.ops = fake
        q, k, v, piv, kf_out, tgt, res = _data(K, n, S, h, d)
        D = h * d
        h_ = h
        # ---- single-process reference (same fake ops, full data)
        full_attn = fake.ext_attn(q, k, v, h, d ** -0.5, inject)
        inv = fake.pivot_inv_norm(piv)
        w = orc.blend_weights(n, 1)
        full_prop = []
        for c in range(K):
            ids = orc.keyframe_ids(c)
            idx = fake.nn_search(tgt[c], piv, inv, ids)
            dt = torch.float32
            full_prop.append(fake.gather_blend(kf_out, idx, w if len(ids) == 2 else None, ids, n, res[c], dt))
        # ---- sharded
        sh = sharded.FrameShard(K, comm=GlooComm() if use_comm else None)
        Kl, f0 = sh.Kl, sh.kf0
        loc = lambda t: t.view(3, K, S, D)[:, f0:f0 + Kl].reshape(3 * Kl, S, D)
        fake.calls.clear()
        out = sh.pivotal_attention(loc(q), loc(k), loc(v), h, d ** -0.5, inject, mode=mode)
        ok = torch.equal(out, loc(full_attn))
        parts = [c[3] if len(c) > 3 else "all" for c in fake.calls if c[0] == "ext_attn"]
        want = ["source", "bank"] if (mode or sh.auto_mode(h, S)) == "heads" else ["all"]
        ok = ok and parts == want
        piv_e, inv_e, kfo_e = sh.exchange_halo(piv[f0:f0 + Kl], inv[f0:f0 + Kl], loc(kf_out))
        for j in range(Kl):
            y = sh.propagate(j, tgt[f0 + j], res[f0 + j], piv_e, inv_e, kfo_e, w, n)
            ok = ok and torch.equal(y, full_prop[f0 + j])
        # ---- all local chunks in one call, halo split in two halves, the first chunk deferred behind the halo
        tgt_all = torch.cat([tgt[f0 + j] for j in range(Kl)])
        res_all = torch.stack([res[f0 + j].view(3, n, S, D) for j in range(Kl)], dim=1).reshape(3 * Kl * n, S, D)
        want = torch.stack([full_prop[f0 + j].view(3, n, S, D) for j in range(Kl)], dim=1).reshape(3 * Kl * n, S, D)
        got = sh.propagate_all(tgt_all, res_all, piv_e, inv_e, kfo_e, w, n)
        ok = ok and torch.equal(got, want)
        h = sh.halo_start(piv[f0:f0 + Kl], inv[f0:f0 + Kl])
        pe, ie, ke, reqs = sh.halo_finish(h, loc(kf_out), wait=False)
        first, rest = sh.propagate_all(tgt_all, res_all, pe, ie, ke, w, n, halo_reqs=reqs)
        ok = ok and torch.equal(first, full_prop[f0])
        if Kl > 1:
            ok = ok and torch.equal(rest, want.view(3, Kl, n, S, D)[:, 1:].reshape(3 * (Kl - 1) * n, S, D))
        # ---- the in-place form of the two-pass order: producers write into the halo-extended buffers, ONE grouped
        #      neighbour exchange per block, the propagation waits for it
        ext = sh.ext_alloc(S, D, q.dtype, q.device)
        o = 1 if world > 1 else 0
        ext[0][o:].copy_(piv[f0:f0 + Kl])
        fake.pivot_inv_norm(ext[0][o:], out=ext[1][o:])
        fake.calls.clear()
        pe, ie, ke, reqs = sh.pivotal_block(loc(q), loc(k), loc(v), h_, d ** -0.5, inject, ext, mode=mode)
        parts = [c[3] if len(c) > 3 else "all" for c in fake.calls if c[0] == "ext_attn"]
        ok = ok and parts == (["source", "bank"] if (mode or sh.auto_mode(h_, S)) == "heads" else ["all"])
        ok = ok and torch.equal(ke.view(3, Kl + o, S, D)[:, o:].reshape(3 * Kl, S, D), loc(full_attn))
        ke.view(3, Kl + o, S, D)[:, o:].copy_(loc(kf_out).view(3, Kl, S, D))   # the propagation data of this test
        reqs2 = sh._p2p([ke.view(3, Kl + o, S, D)[b, -1] for b in range(3)],
                        [ke.view(3, Kl + o, S, D)[b, 0] for b in range(3)]) if world > 1 else []
        sh.halo_wait(reqs2)
        first, rest = sh.propagate_all(tgt_all, res_all, pe, ie, ke, w, n, halo_reqs=reqs)
        ok = ok and torch.equal(first, full_prop[f0])
        if Kl > 1:
            ok = ok and torch.equal(rest, want.view(3, Kl, n, S, D)[:, 1:].reshape(3 * (Kl - 1) * n, S, D))
        if rank > 0:          # the halo slot holds the left neighbour's last keyframe: pivots, inverse norms, output
            ok = ok and torch.equal(pe[0], piv[f0 - 1]) and torch.equal(ie[0], inv[f0 - 1])
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("inject", [False, True])
@pytest.mark.parametrize("mode,h", [("heads", 2), ("bank", 2), (None, 4), (None, 3)])
def test_sharded_equals_single_process(inject, mode, h):
    """Both exchange patterns of the pivotal pass (head re-sharding, bank all-gather) and the default choice
    (heads when they divide over the ranks: h = 4 -> heads, h = 3 -> bank)."""
    world, K, n, S, d = 2, 4, 2, 12, 8
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, K, n, S, h, d, inject, mode, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


@pytest.mark.parametrize("inject", [False, True])
@pytest.mark.parametrize("mode", ["heads", "bank"])
def test_uneven_shards(inject, mode):
    """K = 5 keyframes over 2 ranks -> runs of 3 and 2 (cfg5 is 4,3,3,3,3,3,3,3)."""
    world, K, n, S, h, d = 2, 5, 2, 12, 2, 8
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, K, n, S, h, d, inject, mode, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


@pytest.mark.parametrize("K,mode,inject", [(4, "heads", True), (5, "heads", False), (4, "bank", False), (5, "bank", True)])
def test_sharded_over_comm_interface(K, mode, inject):
    """The same comparisons with FrameShard on its C-ABI-comm branch (`comm=`), a gloo-backed stand-in with HipComm's
    interface doing the moving: even and uneven runs, both exchange patterns, halo in two halves."""
    world, n, S, h, d = 2, 2, 12, 2, 8
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, K, n, S, h, d, inject, mode, ret, True), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


@pytest.mark.parametrize("inject", [False, True])
@pytest.mark.parametrize("K,h,mode", [(8, 8, "heads"), (8, 8, None), (8, 8, "bank"), (25, 5, None)])
def test_world8_baseline_geometries(K, h, mode, inject):
    """BASELINE config 3 (K = 8 keyframes over 8 ranks: Kl = 1, the rank's ONLY chunk sits behind the halo, so
    `propagate_all(..., halo_reqs)` runs with no local chunk to issue first; 8 heads -> head re-sharding, also the
    per-block default and the single-collective bank form) and config 5 (K = 25: runs 4,3,3,3,3,3,3,3; SD2.1's 5 heads
    (10 and 20 at the coarser levels: the same branch) do not divide over 8 ranks -> bank form through the row
    all-gather), both injection states: every rank
    equal to the single-process result bit for bit."""
    world, n, S, d = 8, 2, 6, 4
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, K, n, S, h, d, inject, mode, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_world8_over_comm_interface():
    """Config 5's uneven runs at world 8 on FrameShard's C-ABI-comm branch (`allgather_rows`, uneven `all_to_all_rows`,
    per-dtype halo groups)."""
    world, K, n, S, h, d = 8, 25, 2, 6, 5, 4
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, K, n, S, h, d, False, None, ret, True), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_shard_runs():
    from tokenflow_amd import sharded
    sh = sharded.FrameShard(5)          # world 1
    assert (sh.Kl, sh.kf0, sh.counts, sh.even) == (5, 0, [5], True)
    one = sharded.FrameShard.__new__(sharded.FrameShard)
    assert [K // 8 + (1 if r < K % 8 else 0) for K in (25,) for r in range(8)] == [4, 3, 3, 3, 3, 3, 3, 3]


def _bootstrap_worker(rank, world, port, fail_rank, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tokenflow_amd import comm as tfcomm
        closed = []

        class Fake:
            def __init__(self, uid, r, w):
                if r == fail_rank and len(closed) == 0 and Fake.made == 1:   # the SECOND communicator fails on one rank
                    raise RuntimeError("no device for you")
                Fake.made += 1
                self.rank, self.world = r, w

            def close(self):
                closed.append(self)
        Fake.made = 0
        comms, why = tfcomm.bootstrap(rank, world, 2, make=Fake)
        if fail_rank < 0:
            ret[rank] = comms is not None and len(comms) == 2 and why is None and not closed
        else:   # every rank gets the same verdict and the reason; whatever a rank had created is closed again
            ret[rank] = comms is None and "no device for you" in why and len(closed) == (1 if rank == fail_rank else 2)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [-1, 1])
def test_comm_bootstrap_agrees_on_every_rank(fail_rank):
    """bench.py's default N > 1 path creates the library's communicators through `comm.bootstrap`: all ranks come
    back with a full set, or all come back with None and the failing rank's reason (nobody left in a collective)."""
    world = 3
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bootstrap_worker, args=(world, port, fail_rank, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def _hooks_worker(rank, world, port, K, inject_t, mode, one_pass_chunks, ret):
    """The drop-in hook layer sharded over `world` processes (register_frame_shard) against the same hooks in one
    process: every rank runs the pivotal pass on ITS keyframes and the chunk passes of ITS chunks."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import tokenflow_utils as tfu
        from oracle import golden_cases as gc
        from tests import fake_diffusers as fd
        from tests.fake_ops import FakeOps
        from tokenflow_amd import hooks, sharded
        fake = FakeOps()
        hooks.ops = fake
        sharded.ops = fake
        cfg = gc.BLOCKS_CFG

        def pipe():
            torch.manual_seed(cfg["seed"])
            p = fd.FakePipeline(dims=cfg["dims"], heads=cfg["heads"], cross_dim=cfg["cross_dim"]).eval()
            tfu.register_extended_attention_pnp(p, [1])
            tfu.set_tokenflow(p.unet)
            tfu.register_time(p, inject_t)
            return p
        n, S = 2, 12
        g = torch.Generator().manual_seed(11)
        ok = True
        for lvl, blk_of in ((0, lambda p: p.unet.up_blocks[3].attentions[1].transformer_blocks[0]),
                            (1, lambda p: p.unet.down_blocks[1].attentions[0].transformer_blocks[0])):
            D = cfg["dims"][lvl]
            x_piv = torch.randn(3, K, S, D, generator=g)
            enc = torch.randn(3, K, 7, cfg["cross_dim"], generator=g)
            chunks = [torch.randn(3 * n, S, D, generator=g) for _ in range(K)]
            enc_n = torch.randn(3 * n, 7, cfg["cross_dim"], generator=g)
            with torch.no_grad():
                # ---- one process: all K keyframes, all K chunks
                ref_p = pipe()
                blk = blk_of(ref_p)
                tfu.register_pivotal(ref_p, True)
                piv_out = blk(x_piv.reshape(3 * K, S, D), encoder_hidden_states=enc.reshape(3 * K, 7, -1)).view(3, K, S, D)
                tfu.register_pivotal(ref_p, False)
                want = []
                for c in range(K):
                    tfu.register_batch_idx(ref_p, c)
                    want.append(blk(chunks[c], encoder_hidden_states=enc_n))
                # ---- this rank: its keyframes, its chunks
                sh = sharded.FrameShard(K)
                my_p = pipe()
                tfu.register_frame_shard(my_p, sh)
                blk = blk_of(my_p)
                lo, hi = sh.kf0, sh.kf0 + sh.Kl
                tfu.register_pivotal(my_p, True)
                if mode is not None:
                    sh.auto_mode = lambda heads, S_: mode
                got_p = blk(x_piv[:, lo:hi].reshape(3 * sh.Kl, S, D),
                            encoder_hidden_states=enc[:, lo:hi].reshape(3 * sh.Kl, 7, -1)).view(3, sh.Kl, S, D)
                ok = ok and torch.equal(got_p, piv_out[:, lo:hi])
                if rank % 2 == 1:      # what a captured pivotal pass ends with: the halos waited for here, not in the
                    from tokenflow_amd import hooks as _hooks       # first chunk pass -- same results either way
                    _hooks.join_frame_shard(my_p)
                    ok = ok and blk.__dict__["_tf_halo"][3] == []
                tfu.register_pivotal(my_p, False)
                if one_pass_chunks and sh.Kl > 1:      # the rank's chunks in ONE pass (batch_idx = a run)
                    tfu.register_batch_idx(my_p, range(lo, hi))
                    x_all = torch.stack([chunks[c].view(3, n, S, D) for c in range(lo, hi)], dim=1).reshape(-1, S, D)
                    e_all = enc_n.view(3, n, 7, -1).repeat(1, sh.Kl, 1, 1).reshape(3 * sh.Kl * n, 7, -1)
                    got = blk(x_all, encoder_hidden_states=e_all).view(3, sh.Kl, n, S, D)
                    for j, c in enumerate(range(lo, hi)):
                        ok = ok and torch.allclose(got[:, j].reshape(3 * n, S, D).float(), want[c].float(), atol=1e-6, rtol=0)
                else:
                    for c in range(lo, hi):
                        tfu.register_batch_idx(my_p, c)
                        ok = ok and torch.equal(blk(chunks[c], encoder_hidden_states=enc_n), want[c])
                # a chunk of another rank is refused, not silently matched against the wrong keyframes
                other = (hi % K) if world > 1 else None
                if other is not None and not (lo <= other < hi):
                    tfu.register_batch_idx(my_p, other)
                    try:
                        blk(chunks[other], encoder_hidden_states=enc_n)
                        ok = False
                    except ValueError:
                        pass
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,K,inject_t,mode,one_pass", [(2, 4, 1, None, False), (2, 5, 0, "bank", False),
                                                            (2, 4, 0, "heads", True), (4, 4, 1, None, False)])
def test_hooks_sharded_equal_single_process(world, K, inject_t, mode, one_pass):
    """`register_frame_shard`: the reference's hook API with the keyframes and chunks sharded over ranks -- pivotal
    pass on the local keyframes (extended attention through the exchange, halo to the right neighbour), chunk passes
    with GLOBAL chunk indices reading keyframes c and c-1 from the halo-extended caches -- equals the one-process
    hooks bit for bit, with and without q/k injection (t = 1 is on the schedule), even and uneven runs, one keyframe
    per rank (world 4), per-chunk and one-pass chunk order."""
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_hooks_worker, args=(world, port, K, inject_t, mode, one_pass, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}
