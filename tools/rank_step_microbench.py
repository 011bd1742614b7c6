#!/usr/bin/env python
"""One rank's share of a cfg2 step at W GPUs (default: rank 1 of 8 -- one keyframe, one chunk, a left neighbour),
with the wire taken out: `FrameShard` runs on a stand-in comm with HipComm's interface whose exchanges are local
copies of the same size on the exchange stream, so every launch, buffer, stream hand-over and host call of the real
rank is there and only the xGMI transfer time is missing.  The step is bench.py's own `run_step` (pivotal pass over
the 16 blocks, then their propagation: the reference's call order).

Prints, per configuration (attention split on / off, exchange pattern):
  * the whole rank step: GPU time (events around the asynchronously issued step: what the rank's step costs when the
    host can run ahead) and host issue time, median / min / max over the repetitions;
  * per level, in isolation: pivotal pass and propagation of ONE block, GPU and host time, median / max -- isolated
    blocks of the coarse levels are host-bound (the GPU waits for the next launch), inside a step the host runs ahead
    during the level-0 blocks;
  * the bytes a rank would put on the wire per block, and the time they take at an assumed per-link rate.
Target: step <= single-GPU step / 6."""
import argparse
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tokenflow_amd import sharded, workload  # noqa: E402


class LocalComm:
    """HipComm's interface; every exchange is a same-size device-to-device hipMemcpyAsync on the stream it is handed
    (one foreign call per message, like the RCCL calls it stands for)."""

    def __init__(self, rank, world):
        import ctypes
        self.rank, self.world = rank, world
        self.bytes = 0
        self._hip = ctypes.CDLL("libamdhip64.so")
        self._hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                             ctypes.c_void_p]

    def _copy(self, dst, src, nbytes, stream):
        if stream is None:
            stream = torch.cuda.current_stream(dst.device).cuda_stream
        rc = self._hip.hipMemcpyAsync(dst.data_ptr(), src.data_ptr(), nbytes, 3, stream)   # 3 = device to device
        assert rc == 0, rc

    def allgather(self, local, bank, stream=None):
        nb = local.numel() * local.element_size()
        for p in range(self.world):
            self._hip.hipMemcpyAsync(bank.data_ptr() + p * nb, local.data_ptr(), nb, 3,
                                     stream if stream is not None else torch.cuda.current_stream(bank.device).cuda_stream)
        self.bytes += nb * (self.world - 1)
        return bank

    def allgather_rows(self, local, bank, rows, stream=None):
        rb = bank[0].numel() * bank.element_size()
        st = stream if stream is not None else torch.cuda.current_stream(bank.device).cuda_stream
        off = 0
        for p, r in enumerate(rows):
            self._hip.hipMemcpyAsync(bank.data_ptr() + off * rb, local.data_ptr(), min(r, local.shape[0]) * rb, 3, st)
            off += r
        self.bytes += local.numel() * local.element_size() * (self.world - 1)
        return bank

    def all_to_all_rows(self, send, recv, send_rows=None, recv_rows=None, stream=None):
        n = min(send.numel(), recv.numel()) * send.element_size()
        self._copy(recv, send, n, stream)
        self.bytes += send.numel() * send.element_size() * (self.world - 1) // self.world
        return recv

    def sendrecv(self, send, send_peer, recv, recv_peer, stream=None):
        if recv_peer >= 0:
            for s, r in zip(send, recv):
                self._copy(r, s, r.numel() * r.element_size(), stream)
        if send_peer >= 0:
            self.bytes += sum(t.numel() * t.element_size() for t in send)


def latest_driver_ms(default=26.0):
    """ms_per_step of the newest driver bench record (BENCH_rNN.json at the repository root)."""
    import glob
    import json
    for path in sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json")), reverse=True):
        try:
            return float(json.load(open(path))["parsed"]["ms_per_step"])
        except Exception:  # noqa: BLE001
            continue
    return default


def measure(fn, reps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    gpu, host = [], []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        t0 = time.perf_counter()
        fn()
        host.append((time.perf_counter() - t0) * 1e6)
        e1.record()
        torch.cuda.synchronize()
        gpu.append(e0.elapsed_time(e1) * 1e3)
    return gpu, host


def fmt(xs):
    return "%8.1f (min %8.1f, max %8.1f)" % (statistics.median(xs), min(xs), max(xs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=1)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--link-gbs", type=float, default=50.0, help="assumed achieved rate of one xGMI link, GB/s")
    ap.add_argument("--single-ms", type=float, default=latest_driver_ms(),
                    help="single-GPU step to quote ratios against (default: ms_per_step of the newest BENCH_r*.json the "
                         "driver left at the repository root)")
    ap.add_argument("--profile", action="store_true", help="cProfile of the host side of 5 steps (first configuration)")
    ap.add_argument("--only", default="", help="e.g. 'split,auto': run one configuration (split|onepass , auto|heads|bank)")
    ap.add_argument("--no-levels", action="store_true", help="skip the per-level isolated blocks (profiling runs)")
    ap.add_argument("--no-copies", action="store_true",
                    help="--native: the loopback exchanges move nothing (tf_comm_loopback_copies(0)): the rank step with the "
                         "stand-in copies excluded from the GPU time as well")
    ap.add_argument("--wire-model", default="", metavar="LAT_US,GBPS",
                    help="--native: the loopback transport's wire MODEL (tf_comm_loopback_wire): every exchange also holds "
                         "its stream for LAT_US + bytes on its busiest link / GBPS GB/s (e.g. 25,50), so the overlap of the "
                         "exchanges with compute is executed; prints the step under the model and the exchange time it "
                         "leaves exposed (against the same step with the wire switched off)")
    ap.add_argument("--native", action="store_true",
                    help="sharded.NativeShard: the pivotal pass of a block as ONE library call (tf_rank_pivotal) on the "
                         "library's loopback transport (tf_comm_init_loopback: the same wire-less stand-in, in C)")
    args = ap.parse_args()
    cfg = workload.CONFIGS[args.config]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    w = bench.blend_w(cfg.chunk, dev)
    for split in (True, False):
        for mode in (None, "heads", "bank"):
            if args.only and args.only != f"{'split' if split else 'onepass'},{mode or 'auto'}":
                continue
            wire = tuple(float(x) for x in args.wire_model.split(",")) if args.wire_model else None
            if args.native:
                from tokenflow_amd import _lib
                from tokenflow_amd.comm import HipComm
                comm = HipComm.loopback(args.rank, args.world, copies=not args.no_copies, wire=wire)
                comm.bytes = 0
                hcomm = HipComm.loopback(args.rank, args.world, copies=not args.no_copies, wire=wire)
                shard = sharded.NativeShard(cfg.K, comm, hcomm, attn_split=split)
            else:
                comm = LocalComm(args.rank, args.world)
                shard = sharded.FrameShard(cfg.K, comm=comm, attn_split=split)
            if mode == "heads" and any(l[2] % args.world for l in cfg.levels):
                continue
            gen = torch.Generator(device=dev).manual_seed(1234 + args.rank)
            blocks = [bench.Block(cfg, lvl, inj, shard, gen, dev) for lvl, inj in workload.BLOCKS]
            modes = [mode or shard.auto_mode(l[2], l[0]) for l in cfg.levels]
            print(f"=== {'NATIVE executor, ' if args.native else ''}{'stand-in copies EXCLUDED, ' if args.no_copies else ''}rank {args.rank} of {args.world}, {cfg.name}: Kl={shard.Kl}, attention "
                  f"{'split+merge' if split else 'one-pass (bit-exact)'}, exchange per level {modes}", flush=True)
            if args.profile:
                import cProfile
                import pstats
                for _ in range(3):
                    bench.run_step(cfg, blocks, shard, False, w, exchange=mode)
                torch.cuda.synchronize()
                pr = cProfile.Profile()
                pr.enable()
                for _ in range(5):
                    bench.run_step(cfg, blocks, shard, False, w, exchange=mode)
                pr.disable()
                torch.cuda.synchronize()
                pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
                args.profile = False
            for inj in (False, True):
                gpu, host = measure(lambda: bench.run_step(cfg, blocks, shard, inj, w, exchange=mode), args.reps, 3)
                med = statistics.median(gpu)
                tag = f"wire model {wire[0]:g} us + bytes / {wire[1]:g} GB/s per link" if wire else "wire-less"
                print(f"  step inject={int(inj)}: GPU {fmt(gpu)} us   host issue {fmt(host)} us   -> "
                      f"{args.single_ms * 1e3 / med:4.2f}x of the {args.single_ms} ms single-GPU step ({tag})",
                      flush=True)
                if wire and args.native:   # the same step, wire off: the difference is the exchange time left exposed
                    for c in (comm, hcomm):
                        _lib.check(_lib.load().tf_comm_loopback_wire(c._h, 0.0, 0.0), "tf_comm_loopback_wire")
                    gpu0, _ = measure(lambda: bench.run_step(cfg, blocks, shard, inj, w, exchange=mode), args.reps, 3)
                    for c in (comm, hcomm):
                        _lib.check(_lib.load().tf_comm_loopback_wire(c._h, wire[0], wire[1]), "tf_comm_loopback_wire")
                    med0 = statistics.median(gpu0)
                    print(f"    wire off: GPU {fmt(gpu0)} us -> exposed exchange time {med - med0:7.1f} us per step "
                          f"({(med - med0) / 16:5.1f} us per block)", flush=True)
            comm.bytes = 0
            bench.run_step(cfg, blocks, shard, False, w, exchange=mode)
            torch.cuda.synchronize()
            wire_us = comm.bytes / (args.link_gbs * 1e3) / min(args.world - 1, 7)
            if comm.bytes:
                print(f"  wire: {comm.bytes / 1e6:7.1f} MB sent per rank and step; at {args.link_gbs:.0f} GB/s per link over "
                      f"{min(args.world - 1, 7)} links {wire_us:7.0f} us if nothing overlapped (halo: one link only)")
            for lvl in range(0 if not args.no_levels else len(cfg.levels), len(cfg.levels)):
                blk = next(b for b in blocks if b.lvl == lvl)
                for inj in ((False, True) if blk.injected or any(b.injected for b in blocks if b.lvl == lvl) else (False,)):
                    b1 = next((b for b in blocks if b.lvl == lvl and b.injected), blk) if inj else blk
                    state = {}

                    def pivotal():
                        scale = (b1.D // b1.h) ** -0.5
                        state["h"] = shard.pivotal_block(b1.q, b1.k, b1.v, b1.h, scale, inj, b1.ext, mode=mode, inv_norm=True)

                    def prop():
                        pe, ie, ke, reqs = state["h"]
                        shard.propagate_all(b1.tgt, b1.res, pe, ie, ke, w, cfg.chunk, halo_reqs=reqs)
                    g1, h1 = measure(pivotal, args.reps, 3)
                    g2, h2 = measure(prop, args.reps, 3)
                    print(f"  level {lvl} (S={b1.S}, D={b1.D}) inject={int(inj)} isolated block: pivotal GPU "
                          f"{statistics.median(g1):7.1f} (max {max(g1):7.1f}) host {statistics.median(h1):6.1f} | "
                          f"propagation GPU {statistics.median(g2):7.1f} (max {max(g2):7.1f}) host "
                          f"{statistics.median(h2):6.1f} us", flush=True)


if __name__ == "__main__":
    main()
