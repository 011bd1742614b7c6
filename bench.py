#!/usr/bin/env python
"""Hot-path benchmark: denoising-step frames/sec of TokenFlow's hook layer on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A *step* = one pass of the hot path over one synthetic video: for each of
the 16 transformer blocks of the SD UNet, one extended attention over the 3K-keyframe batch
(pivotal pass: tf_ext_attn_fwd + tf_pivot_inv_norm) and, per frame chunk, one NN search and
one gather/blend/residual (propagation passes: tf_nn_search + tf_gather_blend).  Inputs are
synthetic post-projection tensors resident in HBM before the timed region; the Linear /
LayerNorm / conv layers of the UNet are diffusers' and are not part of the path.  q/k
injection is on for every other step (the reference injects during the first 50 % of the
timesteps, config_pnp.yaml:21).

N > 1: frames are sharded over ranks (tokenflow_amd/sharded.py): the SAME video is split,
so scaling is "strong"; the pivotal-pass exchange (frames <-> heads all-to-all, or the bank
all-gather with --pivotal-exchange bank) and the neighbour halo exchange run through
torch.distributed (RCCL) inside the timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the head-dim-40
extended attention of level 0), timed with HIP events on the launch stream inside the timed
region; `cpu_baseline` is the CPU oracle ("port" of the reference hook path, fp32 torch CPU)
timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tokenflow_amd import ops, sharded, workload  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg2", choices=list(workload.CONFIGS))
    ap.add_argument("--pivotal-exchange", default="auto", choices=["auto", "heads", "bank"],
                    help="N > 1: how the pivotal pass is exchanged (sharded.py); auto = heads when they divide")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N > 1: nccl (= RCCL); gloo lets several ranks share one GPU on a development box "
                         "(functional check of the N > 1 path, its timing means nothing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-levels", default="0,1,2,3")
    return ap.parse_args()


class Block:
    """Synthetic inputs of one transformer block on this rank."""

    def __init__(self, cfg, lvl, injected, shard, gen, dev):
        S, D, h = cfg.levels[lvl]
        self.S, self.D, self.h, self.lvl, self.injected = S, D, h, lvl, injected
        Kl, n = shard.Kl, cfg.chunk
        bf = torch.bfloat16

        def rnd(*shape):
            return torch.randn(*shape, generator=gen, device=dev, dtype=torch.float32).to(bf)
        self.q, self.k, self.v = rnd(3 * Kl, S, D), rnd(3 * Kl, S, D), rnd(3 * Kl, S, D)
        ln = torch.nn.functional.layer_norm
        self.pivots = ln(torch.randn(Kl, S, D, generator=gen, device=dev), (D,)).to(bf)
        # video-like targets: permuted pivot rows + noise; residual ~ N(0,1)
        self.tgt, self.res = [], []
        for j in range(Kl):
            perm = torch.stack([torch.randperm(S, generator=gen, device=dev) for _ in range(n)])
            t = self.pivots[j].float()[perm.reshape(-1)] + 0.1 * torch.randn(n * S, D, generator=gen, device=dev)
            self.tgt.append(t.to(bf))
            self.res.append(rnd(3 * n, S, D))
        self.attn_flops = workload.attn_flops(cfg.K, S, D) * Kl / cfg.K


def run_step(cfg, blocks, shard, inject_on, w, events=None, exchange=None):
    n = cfg.chunk
    for blk in blocks:
        inj = inject_on and blk.injected and cfg.pnp
        if events is not None and blk.lvl == 0:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if shard.world == 1:
            kf_out = ops.ext_attn(blk.q, blk.k, blk.v, blk.h, (blk.D // blk.h) ** -0.5, inj)
        else:
            kf_out = shard.pivotal_attention(blk.q, blk.k, blk.v, blk.h, (blk.D // blk.h) ** -0.5, inj, mode=exchange)
        if events is not None and blk.lvl == 0:
            e1.record()
            events.append((e0, e1, blk.attn_flops))
        inv = ops.pivot_inv_norm(blk.pivots)
        piv_e, inv_e, kfo_e = shard.exchange_halo(blk.pivots, inv, kf_out)
        for j in range(shard.Kl):
            shard.propagate(j, blk.tgt[j], blk.res[j], piv_e, inv_e, kfo_e, w, n)


def blend_w(n, dev):
    s = torch.arange(0, n)
    d1, d2 = torch.abs(s - n // 2), torch.abs(s + n - n // 2)
    return torch.sigmoid(d2 / (d1 + d2)).to(dev)


def usable_cores():
    """Host cores this process may actually use: CPU affinity capped by the cgroup CPU quota
    (the GPU box exposes 256 logical CPUs but grants a 16-CPU quota; 256 threads run 4x slower)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(cfg, levels):
    """Time the CPU oracle on a bounded sample and extrapolate to one full step.
    Sample: per level, ONE frame (all heads unless the score matrices pass 16 GB, batched as the reference's bmm is) of the bank problem
    (its queries against the full K*S-key bank) and of the source problem, ONE chunk of NN search (two
    keyframes) and ONE chunk of gather/blend; scaled by heads x frames x branches, chunk count and block
    count.  About 10 s of CPU work at cfg2 (each piece runs three times: warm-up + best of two)."""
    from oracle import tokenflow_oracle as orc
    torch.set_num_threads(usable_cores())
    K, n, C = cfg.K, cfg.chunk, cfg.K
    g = torch.Generator().manual_seed(0)
    total, t_spent, parts = 0.0, 0.0, []

    def timed(fn, reps=2):
        """min wall time over `reps` runs after one untimed warm-up (thread pool, allocator)."""
        fn()
        best = float("inf")
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            best = min(best, time.perf_counter() - t0)
        return best, r

    for lvl in range(4):
        S, D, h = cfg.levels[lvl]
        d = D // h
        nblk = sum(1 for l, _ in workload.BLOCKS if l == lvl)
        if lvl not in levels:
            continue
        # heads in the sample: all of them while scores + softmax (2 x hs*S*K*S fp32) stay under 16 GB of host
        # memory (8.6 GB at cfg2 level 0; the reference materialises the same matrices, tokenflow_utils.py:173-179)
        hs = max(1, min(h, int(16e9 // (8.0 * S * K * S))))
        q = torch.randn(hs, S, d, generator=g)
        kb, vb = torch.randn(hs, K * S, d, generator=g), torch.randn(hs, K * S, d, generator=g)

        def attn(kk, vv):
            sim = torch.bmm(q, kk.transpose(-1, -2)) * d ** -0.5        # tokenflow_utils.py:173-175
            return torch.bmm(sim.softmax(dim=-1), vv)                   # :177-179
        t_bank, _ = timed(lambda: attn(kb, vb))
        t_src, _ = timed(lambda: attn(kb[:, :S], vb[:, :S]))
        t_attn = K * (h / hs) * (2 * t_bank + t_src)
        piv = torch.randn(K, S, D, generator=g)
        tgt = torch.randn(n, S, D, generator=g)
        kf_out = torch.randn(3 * K, S, D, generator=g)
        res = torch.randn(3 * n, S, D, generator=g)
        t_nn2, (idx, _) = timed(lambda: orc.nn_search(tgt, piv, 1))    # util.py:61-69 + :335-343
        t_gb2, _ = timed(lambda: orc.gather_blend(kf_out, idx, 1, n, residual=res))   # :362-397
        t_prop = (C - 0.5) * t_nn2 + (C - 0.5) * t_gb2                  # chunk 0 matches one keyframe (~half)
        total += nblk * (t_attn + t_prop)
        t_spent += t_bank + t_src + t_nn2 + t_gb2
        parts.append(f"L{lvl}: bank {t_bank:.2f}s src {t_src:.2f}s nn {t_nn2:.2f}s gather {t_gb2:.2f}s")
    return dict(value=cfg.frames / total, unit="frames/s", cores=torch.get_num_threads(), kind="port",
                sample=("oracle (fp32 torch CPU restatement of the reference hooks) timed per level on one "
                        "frame (all heads, memory permitting) of the bank+source attention, one 2-keyframe NN-search chunk and one "
                        "gather/blend chunk, extrapolated by heads*frames*branches, chunks and blocks to a full "
                        f"step ({total:.1f} s/step extrapolated from {t_spent:.1f} s of best-of-2 samples; " + "; ".join(parts) + ")"))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched through torch.distributed.run (one rank per GPU)")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    cfg = workload.CONFIGS[args.config]
    shard = sharded.FrameShard(cfg.K)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    blocks = [Block(cfg, lvl, inj, shard, gen, dev) for lvl, inj in workload.BLOCKS]
    w = blend_w(cfg.chunk, dev)
    exchange = None if args.pivotal_exchange == "auto" else args.pivotal_exchange
    n_heads_ok = sum(1 for l in cfg.levels if l[2] % world == 0)
    names = {"heads": "frames<->heads all-to-all", "bank": "K/V bank all-gather"}
    exch_name = (names[exchange] if exchange else names["heads"] if n_heads_ok == len(cfg.levels)
                 else names["bank"] if n_heads_ok == 0
                 else "frames<->heads all-to-all on the levels whose heads divide over the ranks, K/V bank all-gather on the others")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        run_step(cfg, blocks, shard, i % 2 == 0, w, exchange=exchange)
    events = []
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_step(cfg, blocks, shard, i % 2 == 0, w, events, exchange=exchange)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = cfg.frames * args.steps / elapsed
    fa, fn, gb = workload.step_work(cfg)
    # dominant kernel: level-0 extended attention (head dim 40 for SD1.5): algorithmic flops / launch time
    durs = [e0.elapsed_time(e1) * 1e-3 for e0, e1, _ in events]
    flops = events[0][2] if events else 0.0
    avg = sum(durs) / max(len(durs), 1)
    achieved = flops / avg / 1e12 if avg > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.isfile(tpath):
        try:
            traffic = json.load(open(tpath)).get(args.config, {}).get("ext_attn_l0_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "denoising-step hot-path frames/sec", "value": round(value, 2), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": cfg.name + " (hot path: 16 blocks x [ext-attn + %d x (NN-search + gather/blend)])" % cfg.K,
                   "frames": cfg.frames, "keyframes": cfg.K, "frames_per_chunk": cfg.chunk,
                   "levels_S_D_heads": [list(l) for l in cfg.levels],
                   "parallelism": "1 GPU" if world == 1 else
                   "frames sharded over %d GPUs; pivotal pass: %s" % (world, exch_name),
                   "step_algorithmic_tflop": round((fa + fn) / 1e12, 2),
                   "step_tflops_achieved": round((fa + fn) / 1e12 / (ms_per_step * 1e-3), 1)},
        "roofline": {"kernel": "ext_attn_kernel<bf16, Dh=%d> level 0 (+ V^T pre-pass inside the event bracket)"
                               % (cfg.levels[0][1] // cfg.levels[0][2]),
                     "bound": "mfma", "achieved": round(achieved, 1), "peak": workload.MFMA_BF16_PEAK / 1e12,
                     "unit": "TFLOP/s", "frac": round(achieved * 1e12 / workload.MFMA_BF16_PEAK, 4),
                     "avg_launch_ms": round(avg * 1e3, 3), "launches_timed": len(durs),
                     "algorithmic_gflop_per_launch": round(flops / 1e9, 1), "traffic": traffic},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            lv = [int(x) for x in args.cpu_sample_levels.split(",") if x != ""]
            out["cpu_baseline"] = cpu_baseline(cfg, lv)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
