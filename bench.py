#!/usr/bin/env python
"""Hot-path benchmark: denoising-step frames/sec of TokenFlow's hook layer on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2] [--graph] [--per-chunk]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches itself through
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU, RCCL);
under torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE as usual.

A *step* = one pass of the hot path over one synthetic video: for each of the 16
transformer blocks of the SD UNet, one extended attention over the 3K-keyframe batch (pivotal pass:
tf_ext_attn_fwd + tf_pivot_inv_norm) and the token propagation of every frame chunk (NN search + gather / blend /
residual: tf_nn_gather_blend_chunks over all chunks of the block in one call; `--per-chunk` issues the reference's
one call per chunk, tf_nn_gather_blend).  Inputs are synthetic post-projection tensors resident in HBM before the
timed region; the Linear / LayerNorm / conv layers of the UNet are diffusers' and are not part of the path.  q/k
injection is on for every other step (the reference injects during the first 50 % of the timesteps,
config_pnp.yaml:21); the two states are also timed separately (`ms_per_step_inject_on/off`).

N > 1: frames are sharded over ranks (tokenflow_amd/sharded.py): the SAME video is split, so scaling is "strong";
the pivotal-pass exchange (frames <-> heads all-to-all or the single-collective bank all-gather, chosen per block;
--pivotal-exchange forces one) and the neighbour halo exchange run through torch.distributed (RCCL) inside the
timed region.  Two forms of the rank's attention are timed in two regions of K steps each: the one-pass form, whose
results equal the single-GPU run in its bit-stable mode (TOKENFLOW_ATTN_NO_SPLIT=1) bit for bit
(`ms_per_step_bit_identical`), and the split form (`ms_per_step_split`:
small grids split the key sequence and merge in fp32; held to the oracle's bound like every other launch).  `value` /
`ms_per_step` report the faster of the two and `value_form` names it.  The step then follows the reference's call order: the pivotal pass over all 16 blocks, then the
propagation of all blocks (the halo of a block travels under the rest of the pivotal pass).

Prints ONE JSON line (rank 0):
  roofline      the dominant kernel: the head-dim-40 extended attention of level 0 WITHOUT q/k injection
                (ext_attn_kernel<.., MODE_ALL>), HIP events on the launch stream around every such call inside the
                timed region (the bracket includes the V^T pre-pass launch); `roofline_inject` the same for the
                injected calls (dual-V kernel + source launch), with the flops it actually executes next to the
                algorithmic figure.  `traffic` is the HBM byte count of one plain launch from rocprofv3 PMC passes
                (profiles/traffic.json; collected separately, never in this run) or null.
  parity        in-run check against the oracle (after the timed region, rank 0, N = 1), one block per level, bf16 and
                f16: `attn_linf_fp32_out` (the normalised fp32 accumulator) against the tolerance 1e-3, the boolean
                `attn_16bit_within_half_ulp` for the 16-bit output tensor, and the tie-aware NN index mismatch / raw
                index-diff rates of a chunk for BOTH target flavours of SURVEY 8(d): video-like and iid (`*_iid`).
  input_sets    step i runs on input set i % n; set s, block b is seeded 1234 + 16 s + b; all sets are
                generated before the timed region (up to 8, at most 64 GB).
  roofline_other  the level-1 attention call (head dim 80 at cfg2), plain and with q/k injection (HIP events around every such
                call INSIDE the timed region, like `roofline`); NN search (MFMA roof) and gather/blend (HBM roof) on one
                level-0 chunk, HIP events, after the timed region (rank 0, N = 1); since round 6 also ONE level-0
                attention launch of BASELINE configs 4 and 5 (head dim 64) against the MFMA roof.
  other_configs   ms per step of BASELINE config 1's geometry (3 eager steps after the timed region, rank 0, N = 1).
  value_bit_identical / value_split   N > 1: frames/s of each timed form, whatever `value` picked (`value_form`).
  yardstick     same box, same run, after the timed region (rank 0, N = 1): what the vendor libraries reach -- hipBLASLt
                (torch.matmul, bf16 8192^3) and PyTorch-ROCm's fused attention (aotriton flash behind
                scaled_dot_product_attention) on the level-0 bank problems.  Comparison points for the roofline
                fraction, never part of the product path.
  cpu_baseline  the CPU oracle ("port" of the reference hook path, fp32 torch CPU, the reference's own
                bmm -> *scale -> softmax -> bmm structure) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import socket
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
    ap.add_argument("--backend", default="nccl", choices=["auto", "nccl", "gloo", "hip", "native"],
                    help="N > 1: nccl (default) = RCCL through torch.distributed -- the plainest path, and the default until "
                         "the library's own communicators have run on a multi-GPU node (no such node was available to "
                         "any round; tools/scale.sh measures both); auto = native, falling back to nccl if the library's "
                         "communicators cannot be created; hip = the exchange steps through "
                         "the library's C ABI (tf_comm_*: RCCL without torch.distributed on the data path; gloo carries "
                         "only the barrier and the unique id); native = hip plus the pivotal pass of a block as ONE library "
                         "call (tf_rank_pivotal: pack, exchanges, attention, unpack, halo issued by native code; a second "
                         "communicator carries the halo); gloo lets several ranks share one GPU on a development "
                         "box (functional check of the N > 1 path, its timing means nothing)")
    ap.add_argument("--no-attn-split", action="store_true",
                    help="N > 1: time ONLY the bit-identical form.  By default the timed region runs the rank's attention "
                         "in the form whose results equal the bit-stable single-GPU run bit for bit (`ms_per_step_bit_identical`), and a "
                         "second timed region of the same length runs the split form (small grids split the key "
                         "sequence inside / over workgroups and merge: equal within the output rounding) and reports it "
                         "as `ms_per_step_split`")
    ap.add_argument("--per-chunk", action="store_true",
                    help="issue the propagation one call per chunk (the reference's granularity) instead of one "
                         "call per block over all chunks")
    ap.add_argument("--graph", action="store_true",
                    help="N = 1: capture the two step variants (injection on / off) into HIP graphs and time the "
                         "replays (launch-bound configurations: cfg1); the roofline bracket is then measured in a "
                         "separate eager pass after the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-ref-l0", action="store_true",
                    help="--cpu-baseline-only with the reference mounted: run level 0 for real (minutes, ~13 GB)")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="run ONLY the cpu_baseline leg and print it (no GPU needed): with the reference tree mounted "
                         "($TOKENFLOW_REFERENCE, default /root/reference) this times the verbatim reference hooks "
                         "(kind = 'reference'), else the oracle port")
    ap.add_argument("--input-sets", type=int, default=0,
                    help="distinct synthetic input sets the steps cycle through (SURVEY 8d: inputs of step s, block b come "
                         "from manual_seed(1234 + 16 s + b)); 0 = auto: one per step, at most 8, at most 64 GB in total")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-yardstick", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the cfg4 / cfg5 level-0 launches and the cfg1 steps that follow the headline's timed region")
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
        # N > 1: the propagation state of the block lives in halo-extended buffers (slot 0 = the left neighbour's last
        # keyframe); the pivots sit in slots 1.. from the start, as the block's norm1 would leave them there
        self.ext = shard.ext_alloc(S, D, bf, dev)
        self.pivots = self.ext[0][1 if shard.world > 1 else 0:]
        self.pivots.copy_(ln(torch.randn(Kl, S, D, generator=gen, device=dev), (D,)).to(bf))
        # video-like targets: permuted pivot rows + noise; residual ~ N(0,1).
        # All local chunks in one tensor, chunk-major: tgt [Kl*n*S, D], res [3, Kl*n, S, D]
        tgt, self.perm = [], []
        for j in range(Kl):
            perm = torch.stack([torch.randperm(S, generator=gen, device=dev) for _ in range(n)]).reshape(-1)
            tgt.append(self.pivots[j].float()[perm] + 0.1 * torch.randn(n * S, D, generator=gen, device=dev))
            self.perm.append(perm)
        self.tgt = torch.cat(tgt).to(bf)
        self.res = rnd(3 * Kl * n, S, D)
        self.attn_flops = workload.attn_flops(cfg.K, S, D) * Kl / cfg.K
        # flops the dual-V (injection) launches execute: QK^T of the bank branches once instead of twice
        self.attn_flops_inject = self.attn_flops - 2.0 * Kl * S * D * (cfg.K * S)


def run_step(cfg, blocks, shard, inject_on, w, events=None, exchange=None, per_chunk=False):
    n = cfg.chunk
    outs = None
    two_pass = shard.world > 1 and not per_chunk
    pending = []
    for blk in blocks:
        inj = inject_on and blk.injected and cfg.pnp
        timed = events is not None and blk.lvl <= 1      # level 0 = `roofline`, level 1 = its `roofline_other` entries
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        scale = (blk.D // blk.h) ** -0.5
        if shard.world == 1:
            kf_out = ops.ext_attn(blk.q, blk.k, blk.v, blk.h, scale, inj)
            halo = None
        elif two_pass:
            # N > 1, the reference's own call order (run_tokenflow_pnp.py:222-231): ONE pivotal UNet pass over all 16
            # blocks, then the chunk passes.  In place: inverse norms and attention output go straight into the block's
            # halo-extended buffers, ONE grouped neighbour exchange per block (pivots, inverse norms, attention output of
            # the last local keyframe), which has the rest of the pivotal pass to arrive.
            pending.append((blk, shard.pivotal_block(blk.q, blk.k, blk.v, blk.h, scale, inj, blk.ext, mode=exchange,
                                                     inv_norm=True)))
        else:
            # the pivots' halo (features + inverse norms of the last local keyframe -> rank r+1) does not depend on
            # the attention: issued first, it travels under it
            halo = shard.halo_start(blk.pivots, ops.pivot_inv_norm(blk.pivots))
            kf_out = shard.pivotal_attention(blk.q, blk.k, blk.v, blk.h, scale, inj, mode=exchange)
        if timed:
            e1.record()
            events.append((e0, e1, inj, blk.lvl))
        if two_pass:
            continue
        if per_chunk or shard.world == 1:
            if halo is None:
                piv_e, inv_e, kfo_e = shard.exchange_halo(blk.pivots, ops.pivot_inv_norm(blk.pivots), kf_out)
            else:
                piv_e, inv_e, kfo_e = shard.halo_finish(halo, kf_out)
        if per_chunk:
            nS = n * blk.S
            res = blk.res.view(3, shard.Kl, n, blk.S, blk.D)
            for j in range(shard.Kl):
                outs = shard.propagate(j, blk.tgt[j * nS:(j + 1) * nS], res[:, j].reshape(3 * n, blk.S, blk.D),
                                       piv_e, inv_e, kfo_e, w, n)
        else:
            outs = shard.propagate_all(blk.tgt, blk.res, piv_e, inv_e, kfo_e, w, n)
    for blk, (piv_e, inv_e, kfo_e, reqs) in pending:
        outs, _ = shard.propagate_all(blk.tgt, blk.res, piv_e, inv_e, kfo_e, w, n, halo_reqs=reqs)
    return outs


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


def other_rooflines(cfg, blocks, w, l1_events=None):
    """The two other kernels of the path on a level-0 chunk (rank 0, N = 1, after the timed region; HIP events on
    the launch stream): NN search against the MFMA roof, gather/blend against the HBM roof."""
    blk = next(b for b in blocks if b.lvl == 0)
    n, S, D, K = cfg.chunk, blk.S, blk.D, cfg.K
    c = min(3, K - 1)
    ids = [c, c - 1] if c > 0 else [c]
    nS = n * S
    tgt = blk.tgt[c * nS:(c + 1) * nS]
    res = blk.res.view(3, K, n, S, D)[:, c].reshape(3 * n, S, D).contiguous()
    kf_out = ops.ext_attn(blk.q, blk.k, blk.v, blk.h, (D // blk.h) ** -0.5, False)
    inv = ops.pivot_inv_norm(blk.pivots)
    idx = ops.nn_search(tgt, blk.pivots, inv, ids)

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    P = len(ids)
    t_nn = timed(lambda: ops.nn_search(tgt, blk.pivots, inv, ids))
    out_dt = torch.float32 if P == 2 else kf_out.dtype
    t_gb = timed(lambda: ops.gather_blend(kf_out, idx, w if P == 2 else None, ids, n, res, out_dt))
    fl = workload.nn_flops(n, S, D, P)
    e_in, e_out = kf_out.element_size(), torch.empty(0, dtype=out_dt).element_size()
    by = 3.0 * nS * D * (P * e_in + res.element_size() + e_out) + P * nS * 4
    # what the step actually issues: ONE call per block over all K chunks (search of every chunk in one launch + the
    # gather); its flops are the searches' (chunk 0 matches one keyframe), its time includes the HBM-bound gather
    t_all = timed(lambda: ops.propagate_chunks(blk.tgt, blk.pivots, inv, kf_out, w, n, K, 0, True, blk.res,
                                               torch.float32), reps=10)
    fl_all = workload.nn_flops(n, S, D, 2) * (K - 0.5)
    # the level-1 attention calls (cfg2: head dim 80), plain and with q/k injection: the other streaming kernel of the step,
    # timed INSIDE the timed region like the level-0 call (HIP events around each call; replayed back to back in a loop
    # the same launch runs ~10 % slower -- the chip clocks down under a pure MFMA load)
    lvl1 = []
    blk1 = next((b for b in blocks if b.lvl == 1), None)
    if blk1 is not None and l1_events:
        d1 = blk1.D // blk1.h
        for inj in (False, True):
            durs = [e0.elapsed_time(e1) for e0, e1, i_, l_ in l1_events if i_ == inj and l_ == 1]
            if not durs:
                continue
            t1 = sum(durs) / len(durs)
            lvl1.append({"kernel": "tf_ext_attn_fwd, level 1 (S = %d, %d heads of %d), %s, one call (+ V^T pre-pass%s), "
                                   "HIP events inside the timed region"
                                   % (blk1.S, blk1.h, d1, "q/k injection on" if inj else "no q/k injection",
                                      " + source launch; algorithmic flops, the call executes fewer" if inj else ""),
                         "bound": "mfma", "achieved": round(blk1.attn_flops / t1 / 1e9, 1), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(blk1.attn_flops / t1 / 1e9 / 2500.0, 4), "avg_launch_ms": round(t1, 4),
                         "launches_timed": len(durs), "algorithmic_gflop_per_launch": round(blk1.attn_flops / 1e9, 1)})
    return lvl1 + [
        {"kernel": "tf_nn_gather_blend_chunks (level 0, all %d chunks of a block: the launch pair the step issues -- "
                   "batched NN search + gather/blend/residual; flops = the searches', time = both launches)" % K,
         "bound": "mfma", "achieved": round(fl_all / t_all / 1e9, 1), "peak": 2500.0, "unit": "TFLOP/s",
         "frac": round(fl_all / t_all / 1e9 / 2500.0, 4), "avg_launch_ms": round(t_all, 4),
         "algorithmic_gflop_per_launch": round(fl_all / 1e9, 1)},
        {"kernel": "nn_search (level 0, ONE chunk against %d keyframes, search + finalize launches: GEMM + fused "
                   "normalisation and argmax; the step itself searches all chunks of a block in one launch, "
                   "profiles/r04_kernel_stats.csv)" % P,
         "bound": "mfma", "achieved": round(fl / t_nn / 1e9, 1), "peak": 2500.0, "unit": "TFLOP/s",
         "frac": round(fl / t_nn / 1e9 / 2500.0, 4), "avg_launch_ms": round(t_nn, 4),
         "algorithmic_gflop_per_launch": round(fl / 1e9, 1)},
        {"kernel": "gather_blend (level 0, one chunk: two gathered keyframe rows + residual -> fp32 result)",
         "bound": "hbm", "achieved": round(by / t_gb / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
         "frac": round(by / t_gb / 1e6 / 8000.0, 4), "avg_launch_ms": round(t_gb, 4),
         "algorithmic_mbytes_per_launch": round(by / 1e6, 1)},
    ]


def other_config_rooflines(dev):
    """Driver-visible numbers for the configurations that are not the headline, after the timed
    region (rank 0, N = 1), on seeded synthetic tensors: ONE level-0 extended-attention launch of BASELINE configs 4 and 5
    (head dim 64: K = 10, S = 9216 and K = 25, S = 4096, 5 heads) against the MFMA roof with SURVEY 8(d)'s algorithmic flops,
    HIP events on the launch stream around the second of two launches (pre-pass included).  ~0.3 s of box time."""
    res = []
    for name in ("cfg4", "cfg5"):
        cfg = workload.CONFIGS[name]
        S, D, h = cfg.levels[0]
        K = cfg.K
        try:
            g = torch.Generator(device=dev).manual_seed(4321)
            q, k, v = (torch.randn(3 * K, S, D, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
                       for _ in range(3))
            fn = lambda: ops.ext_attn(q, k, v, h, (D // h) ** -0.5, False)  # noqa: E731
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            fl = workload.attn_flops(K, S, D)
            res.append({"kernel": "tf_ext_attn_fwd, %s level 0 (K = %d, S = %d, %d heads of %d), no q/k injection, one launch "
                                  "(+ V^T pre-pass)" % (name, K, S, h, D // h),
                        "bound": "mfma", "achieved": round(fl / ms / 1e9, 1), "peak": 2500.0, "unit": "TFLOP/s",
                        "frac": round(fl / ms / 1e9 / 2500.0, 4), "avg_launch_ms": round(ms, 3), "launches_timed": 1,
                        "algorithmic_gflop_per_launch": round(fl / 1e9, 1)})
            del q, k, v
        except Exception as e:  # noqa: BLE001
            res.append({"kernel": "tf_ext_attn_fwd, %s level 0" % name, "error": str(e)[:160]})
    torch.cuda.empty_cache()
    return res


def other_config_steps(dev, w_of):
    """ms per step of BASELINE config 1's geometry (8 frames, 256^2, 4 keyframes: launch-bound), 2 warm-up + 3 timed eager
    steps, same `run_step` as the headline."""
    out = {}
    for name, steps in (("cfg1", 3),):
        cfg = workload.CONFIGS[name]
        try:
            sh = sharded.FrameShard(cfg.K, attn_split=False)
            gen = torch.Generator(device=dev).manual_seed(1234)
            blocks = [Block(cfg, lvl, inj, sh, gen, dev) for lvl, inj in workload.BLOCKS]
            w = w_of(cfg.chunk)
            for i in range(2):
                run_step(cfg, blocks, sh, i % 2 == 0, w)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(steps):
                run_step(cfg, blocks, sh, i % 2 == 0, w)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / steps
            out[name] = {"ms_per_step": round(ms, 4), "frames_per_s": round(cfg.frames / (ms * 1e-3), 1), "steps": steps,
                         "workload": cfg.name, "launch": "eager"}
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": str(e)[:160]}
    return out


def power_state(blocks):
    """Shader clock and socket power while the dominant kernel runs: replay the level-0 plain attention for ~0.6 s
    (enqueued asynchronously) and read `rocm-smi` once in the middle.  On this box the launch is power-limited
    (~1.87 GHz at ~1.2 kW against 2.4 GHz nominal), which is what this records next to `roofline`."""
    import re
    import subprocess
    blk = next(b for b in blocks if b.lvl == 0)
    d = blk.D // blk.h
    fn = lambda: ops.ext_attn(blk.q, blk.k, blk.v, blk.h, d ** -0.5, False)  # noqa: E731
    try:
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        one = time.perf_counter() - t0
        for _ in range(max(4, int(1.5 / max(one, 1e-4)))):
            fn()
        time.sleep(0.3)
        txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True,
                             timeout=30).stdout
        torch.cuda.synchronize()
        sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)
        pw = re.search(r"Power \(W\): ([0-9.]+)", txt)
        return {"kernel": "level-0 attention, no injection, replayed back to back", "sclk_mhz": sclk and int(sclk.group(1)),
                "socket_power_w": pw and float(pw.group(1)), "nominal_sclk_mhz": 2400, "source": "rocm-smi, one sample mid-replay"}
    except Exception as e:  # noqa: BLE001
        torch.cuda.synchronize()
        return {"error": str(e)[:120]}


def yardstick(cfg):
    """Vendor-library comparison points measured on this box right after the timed region (SURVEY.md appendix C:
    SDPA 'as a yardstick only').  torch ops on purpose -- nothing here is on the product path."""
    import torch.nn.functional as F

    def timed(fn, reps):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    res = {}
    g = torch.Generator(device="cuda").manual_seed(7)
    try:
        a = torch.randn(8192, 8192, generator=g, device="cuda").bfloat16()
        b = torch.randn(8192, 8192, generator=g, device="cuda").bfloat16()
        ms = timed(lambda: torch.matmul(a, b), 20)
        res["hipblaslt_gemm_bf16_8192_tflops"] = round(2.0 * 8192 ** 3 / ms / 1e9, 1)
        del a, b
    except Exception as e:  # noqa: BLE001
        res["hipblaslt_gemm_bf16_8192_tflops"] = None
        res["gemm_error"] = str(e)[:120]
    try:
        S, D, h = cfg.levels[0]
        K, d = cfg.K, D // h
        q, k, v = (torch.randn(2, h, K * S, d, generator=g, device="cuda").bfloat16() for _ in range(3))
        with torch.nn.attention.sdpa_kernel(torch.nn.attention.SDPBackend.FLASH_ATTENTION):
            ms = timed(lambda: F.scaled_dot_product_attention(q, k, v), 3)
        res["torch_sdpa_flash_level0_bank_tflops"] = round(4.0 * 2 * K * S * K * S * D / ms / 1e9, 1)
        res["torch_sdpa_note"] = ("uncond + cond bank problems of level 0 (head dim %d) on head-major copies made "
                                  "outside the timed bracket" % d)
    except Exception as e:  # noqa: BLE001
        res["torch_sdpa_flash_level0_bank_tflops"] = None
        res["sdpa_error"] = str(e)[:120]
    return res


def cpu_baseline(cfg, levels):
    """Time the CPU oracle on a bounded sample and extrapolate to one full step.
    Sample: per level, ONE query frame (all heads, all three branches: the source problem and the two bank
    problems against the full K*S-key bank) through `oracle.ext_attn_core_bmm` -- the reference's own
    bmm -> *scale -> softmax -> bmm per head (tokenflow_utils.py:172-179) -- ONE chunk of NN search (two
    keyframes) and ONE chunk of gather/blend; scaled by frames, chunk count and block count.  About 10 s of CPU
    work at cfg2 (each piece runs three times: warm-up + best of two)."""
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
        q, kb, vb = (torch.randn(3 * K, S, D, generator=g) for _ in range(3))
        t_frame, _ = timed(lambda: orc.ext_attn_core_bmm(q, kb, vb, h, d ** -0.5, False, frames=[K // 2]))
        t_attn = K * t_frame
        piv = torch.randn(K, S, D, generator=g)
        tgt = torch.randn(n, S, D, generator=g)
        kf_out = torch.randn(3 * K, S, D, generator=g)
        res = torch.randn(3 * n, S, D, generator=g)
        t_nn2, (idx, _) = timed(lambda: orc.nn_search(tgt, piv, 1))    # util.py:61-69 + :335-343
        t_gb2, _ = timed(lambda: orc.gather_blend(kf_out, idx, 1, n, residual=res))   # :362-397
        t_prop = (C - 0.5) * t_nn2 + (C - 0.5) * t_gb2                  # chunk 0 matches one keyframe (~half)
        total += nblk * (t_attn + t_prop)
        t_spent += t_frame + t_nn2 + t_gb2
        parts.append(f"L{lvl}: attn frame {t_frame:.2f}s nn {t_nn2:.2f}s gather {t_gb2:.2f}s")
    return dict(value=cfg.frames / total, unit="frames/s", cores=torch.get_num_threads(), kind="port",
                sample_seconds=round(t_spent, 2), step_seconds_extrapolated=round(total, 1),
                extrapolation="per level: one query frame of the attention, one 2-keyframe chunk of the search and of the "
                              "gather, scaled by frames / chunks / blocks",
                sample=("oracle (fp32 torch CPU restatement of the reference hooks; attention = "
                        "oracle.ext_attn_core_bmm, the reference's per-head bmm/softmax/bmm) timed per level on one "
                        "query frame (all heads, source + two bank problems), one 2-keyframe NN-search chunk and "
                        "one gather/blend chunk, extrapolated by frames, chunks and blocks to a full "
                        f"step ({total:.1f} s/step extrapolated from {t_spent:.1f} s of best-of-2 samples; "
                        + "; ".join(parts) + ")"))


def cpu_baseline_reference(cfg, levels, full_l0=False):
    """`cpu_baseline` with kind = "reference": the VERBATIM hooks of omerbt/TokenFlow (tokenflow_utils.py, loaded unmodified
    by oracle/ref_loader.py from $TOKENFLOW_REFERENCE, default /root/reference) on this host's cores, fp32 -- only where
    the reference tree is mounted (the build container; a GPU box ships the repository alone and reports the port).
    Per sampled level ONE block: the pivotal pass's attention through the reference's own `sa_forward` closure on the
    whole 3K-frame batch (stand-in projections that return the pre-generated q / k / v, as oracle/make_golden.py drives
    it), and ONE chunk pass of the reference's `TokenFlowBlock.forward` propagation branch (chunk 1: two keyframes).
    Level 0's attention (minutes on 8 cores, ~13 GB of score matrices) is extrapolated from level 1 by the ratio of the
    materialised score matrices (S^2: the reference's bmm -> softmax -> bmm is bound by them, not by the flops) unless
    --cpu-ref-l0 runs it for real; the sample says which."""
    from oracle import make_golden as mg
    from oracle import ref_loader
    from tests import fake_diffusers as fd
    tfu, _util = ref_loader.load()
    torch.set_num_threads(usable_cores())
    K, n, C = cfg.K, cfg.chunk, cfg.K
    g = torch.Generator().manual_seed(0)
    total, t_spent, parts, t_attn_by_level = 0.0, 0.0, [], {}
    for lvl in sorted(set(levels) | {1}, reverse=True):
        S, D, h = cfg.levels[lvl]
        nblk = sum(1 for l, _ in workload.BLOCKS if l == lvl)
        if lvl == 0 and not full_l0:
            S1, D1, h1 = cfg.levels[1]
            t_attn = t_attn_by_level[1] * (S / S1) ** 2 * h / h1
            note = "attn %.1fs extrapolated from level 1 by score-matrix size (S^2)" % t_attn
        else:
            q, k, v = (torch.randn(3 * K, S, D, generator=g) for _ in range(3))
            blk = fd.BasicTransformerBlock(D, h)
            blk.attn1 = mg.CoreAttention(q, k, v, h)
            tfu.register_extended_attention_pnp(mg._OneBlock(blk), [])
            blk.attn1.t = 1
            with torch.no_grad():
                t0 = time.perf_counter()
                blk.attn1.forward(torch.zeros(3 * K, S, D))
                t_attn = time.perf_counter() - t0
            t_spent += t_attn
            note = "attn %.2fs" % t_attn
        t_attn_by_level[lvl] = t_attn
        if lvl not in levels:
            continue
        Sp = S if (lvl or full_l0) else cfg.levels[1][0]   # the chunk pass is timed at the level's own size except level 0 ...
        piv = torch.randn(3, K, Sp, D, generator=g)
        kf_out = torch.randn(3 * K, Sp, D, generator=g)
        hidden = torch.randn(3 * n, Sp, D, generator=g)
        pblk = mg._IdBlock(D)
        pblk.__class__ = tfu.make_tokenflow_attention_block(pblk.__class__)
        pblk.pivot_hidden_states, pblk.kf_attn_output, pblk.pivotal_pass, pblk.batch_idx = piv, kf_out, False, 1
        with torch.no_grad():
            t0 = time.perf_counter()
            pblk.forward(hidden.clone())
            t_chunk = time.perf_counter() - t0
        t_spent += t_chunk
        if lvl == 0 and not full_l0:              # ... whose NN search scales with S^2 (4x S: 16x), the gather with S
            t_chunk *= (S / Sp) ** 2
            note += ", chunk pass extrapolated from level 1's size by S^2"
        total += nblk * (t_attn + (C - 0.5) * t_chunk)
        parts.append(f"L{lvl}: {note}, chunk pass {t_chunk:.2f}s")
    return dict(value=cfg.frames / total, unit="frames/s", cores=torch.get_num_threads(), kind="reference",
                sample_seconds=round(t_spent, 2), step_seconds_extrapolated=round(total, 1),
                extrapolated_levels=[] if full_l0 else [l for l in levels if l == 0],
                sample=("VERBATIM reference hooks (tokenflow_utils.py of omerbt/TokenFlow through oracle/ref_loader.py; fp32 "
                        "torch CPU): per level one block -- sa_forward on the whole 3K-frame batch and one two-keyframe chunk "
                        "pass of TokenFlowBlock.forward -- scaled by blocks and chunks to a full step "
                        f"({total:.1f} s/step from {t_spent:.1f} s of single runs; " + "; ".join(parts) + ")"))


def parity_check(cfg, blocks, w):
    """In-run parity, ONE BLOCK PER LEVEL (rank 0, N = 1): attention L_inf against the fp32 oracle on sampled
    query rows of six (branch, frame, head) problems, with and without injection; tie-aware NN index mismatch
    rate of one chunk on sampled targets (tolerance 1e-5 on the fp32 cosine similarity).  The headline figures
    are the worst over the levels; `by_level` keeps each."""
    from oracle import tokenflow_oracle as orc
    by_level, f16_levels, attn_rows, nn_total = [], [], 0, 0
    worst_state, worst_state32, worst_ratio = {}, {}, [0.0]
    nn_bad = nn_diff = 0
    iid_bad = iid_diff = iid_total = 0
    within_half_ulp = True      # every sampled 16-bit output within 1e-3 + half an ulp of its reference value
    K, n = cfg.K, cfg.chunk
    for lvl in range(len(cfg.levels)):
        cands = [b for b in blocks if b.lvl == lvl]
        if not cands:
            continue
        blk = next((b for b in cands if b.injected), cands[0])
        S, D, h = blk.S, blk.D, blk.h
        d = D // h
        rows = torch.arange(min(3, S - 1), S, max(S // 24, 1))
        worst, worst32, floor16, lvl_ratio = {}, {}, 0.0, 0.0
        f16_16, f16_32 = 0.0, 0.0     # the same problems in f16, the reference's own autocast dtype (run_tokenflow_pnp.py:220)
        for dt in (torch.bfloat16, torch.float16):
            dq, dk, dv = ((blk.q, blk.k, blk.v) if dt == torch.bfloat16 else (blk.q.half(), blk.k.half(), blk.v.half()))
            qc, kc, vc = (t.float().cpu().view(3, K, S, h, d) for t in (dq, dk, dv))
            for inject in ((False, True) if cfg.pnp else (False,)):
                out = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject).float().cpu().view(3, K, S, h, d)
                # the same launch with TF_ATTN_OUT_F32: the normalised fp32 accumulator, no 16-bit output rounding
                out32 = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, out_dtype=torch.float32).cpu().view(3, K, S, h, d)
                err = err32 = ratio = 0.0
                for b, f, head in [(0, 0, 0), (0, K - 1, h - 1), (1, 0, h // 2), (1, K - 1, 0), (2, K // 2, h - 1), (2, K - 2, 1)]:
                    bq = 0 if (inject and b > 0) else b
                    qr = qc[bq, f, rows, head]
                    if b == 0:
                        kk, vv = kc[0, f, :, head], vc[0, f, :, head]
                    else:
                        kk, vv = kc[bq, :, :, head].reshape(K * S, d), vc[b, :, :, head].reshape(K * S, d)
                    pm = torch.softmax(qr @ kk.T * d ** -0.5, dim=-1)
                    ref = pm @ vv                                                    # tokenflow_utils.py:173-179
                    e16 = (out[b, f, rows, head] - ref).abs()
                    err = max(err, float(e16.max()))
                    # what ANY tensor of this 16-bit type holding `ref` is off by at worst: half an ulp
                    half_ulp = torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(1e-30)))
                                          - (8 if dt == torch.bfloat16 else 11))
                    within_half_ulp = within_half_ulp and bool((e16 <= 1e-3 + half_ulp).all())
                    err32 = max(err32, float((out32[b, f, rows, head] - ref).abs().max()))
                    if dt == torch.bfloat16:
                        # the bound the parity tests assert for 16-bit P and output (tests/test_kernels_gpu.py)
                        bound = 2e-4 + 2.0 ** -8 * (ref.abs() + pm @ vv.abs())
                        ratio = max(ratio, float((e16 / bound).max()))
                        # what ANY bf16 tensor holding `ref` is off by at worst: half an ulp = 2^(exponent - 8)
                        floor16 = max(floor16, float(torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 8).max()))
                if dt == torch.float16:
                    f16_16, f16_32 = max(f16_16, err), max(f16_32, err32)
                    continue
                state = "inject" if inject else "plain"
                worst[state], worst32[state] = err, err32
                worst_state[state] = max(worst_state.get(state, 0.0), err)
                worst_state32[state] = max(worst_state32.get(state, 0.0), err32)
                worst_ratio[0] = max(worst_ratio[0], ratio)
                lvl_ratio = max(lvl_ratio, ratio)
                attn_rows += int(len(rows)) * 6
        f16_levels.append({"level": lvl, "attn_linf": round(f16_16, 6), "attn_linf_fp32_out": round(f16_32, 6)})
        # NN search of chunk c on sampled targets
        c = min(3, K - 1)
        nS = n * S
        inv = ops.pivot_inv_norm(blk.pivots)
        ids = [c, c - 1] if c > 0 else [c]
        tgt = blk.tgt[c * nS:(c + 1) * nS]
        idx = ops.nn_search(tgt, blk.pivots, inv, ids).cpu()
        sample = torch.randperm(nS, generator=torch.Generator().manual_seed(0))[:2048]
        sim = orc.batch_cosine_sim(tgt[sample.to(tgt.device)].float().cpu(),
                                   blk.pivots[ids].float().cpu().reshape(-1, D))
        n_diff = n_bad = 0
        for p_, s_ in enumerate(sim.chunk(len(ids), dim=1)):
            a, b_ = orc.nn_mismatch_tie_aware(s_, s_.argmax(-1), idx[p_][sample], 1e-5)
            n_diff += a
            n_bad += b_
        total = len(ids) * len(sample)
        nn_bad, nn_diff, nn_total = nn_bad + n_bad, nn_diff + n_diff, nn_total + total
        # flavour (i) of SURVEY 8(d): iid LayerNorm(N(0,1)) targets -- the near-tie worst case of the argmax
        gi = torch.Generator(device=tgt.device).manual_seed(4321 + lvl)
        tgt_iid = torch.nn.functional.layer_norm(torch.randn(nS, D, generator=gi, device=tgt.device), (D,)).to(tgt.dtype)
        idx_i = ops.nn_search(tgt_iid, blk.pivots, inv, ids).cpu()
        sim_i = orc.batch_cosine_sim(tgt_iid[sample.to(tgt.device)].float().cpu(),
                                     blk.pivots[ids].float().cpu().reshape(-1, D))
        i_diff = i_bad = 0
        for p_, s_ in enumerate(sim_i.chunk(len(ids), dim=1)):
            a, b_ = orc.nn_mismatch_tie_aware(s_, s_.argmax(-1), idx_i[p_][sample], 1e-5)
            i_diff += a
            i_bad += b_
        iid_bad, iid_diff, iid_total = iid_bad + i_bad, iid_diff + i_diff, iid_total + total
        by_level.append({"level": lvl, "S": S, "D": D, "head_dim": d,
                         "attn_linf": {k: round(v, 6) for k, v in worst.items()},
                         "attn_linf_fp32_out": {k: round(v, 6) for k, v in worst32.items()},
                         "attn_err_over_bf16_bound": round(lvl_ratio, 4),
                         "bf16_half_ulp_of_largest_ref": round(floor16, 6),
                         "nn_mismatch_rate": n_bad / total, "nn_index_diff_rate": n_diff / total,
                         "nn_mismatch_rate_iid": i_bad / total, "nn_index_diff_rate_iid": i_diff / total})
    linf32 = max(worst_state32.values())
    return {"tolerance": 1e-3,
            "attn_linf_fp32_out": round(linf32, 6),
            "attn_fp32_out_within_tolerance": bool(linf32 < 1e-3),
            "attn_16bit_within_half_ulp": within_half_ulp,
            "attn_linf_16bit_out": round(max(worst_state.values()), 6),
            "attn_linf": round(max(worst_state.values()), 6),
            "attn_linf_by_state": {k: round(v, 6) for k, v in worst_state.items()},
            "attn_err_over_bf16_bound": round(worst_ratio[0], 4),
            "attn_rows_checked": attn_rows,
            "tolerance_note": "north_star's absolute 1e-3 is held by attn_linf_fp32_out (the normalised fp32 accumulator, "
                              "TF_ATTN_OUT_F32) at every level; the 16-bit output tensor adds its own format rounding: "
                              "attn_16bit_within_half_ulp = every sampled 16-bit output within 1e-3 + half an ulp of its "
                              "reference value (what the parity tests assert).  attn_linf (= attn_linf_16bit_out) is absolute.  With P and the output in bf16 (8 significand bits; the reference's "
                              "autocast rounds at the same two points in its 16-bit type) the deviation is relative: "
                              "bound = 2e-4 + 2^-8 (|ref| + softmax.|V|), the one the parity tests assert; "
                              "attn_err_over_bf16_bound <= 1 means inside it.  attn_linf_fp32_out = the same launches with "
                              "TF_ATTN_OUT_F32 (no output rounding): below 1e-3 at EVERY level since round 4 -- frames of "
                              "<= 256 tokens run in the fused kernel, which carries P as hi + lo bf16 (the rounding of P was "
                              "what exceeded 1e-3 where a handful of keys is averaged).  What is left in attn_linf at the "
                              "coarse levels is the rounding of the bf16 OUTPUT itself (half an ulp of |out| ~ 0.5-1: "
                              "bf16_half_ulp_of_largest_ref), which no bf16 tensor can avoid",
            "nn_mismatch_rate": nn_bad / nn_total, "nn_index_diff_rate": nn_diff / nn_total,
            "nn_mismatch_rate_iid": iid_bad / iid_total, "nn_index_diff_rate_iid": iid_diff / iid_total,
            "nn_note": "nn_mismatch_rate = sampled (target, keyframe) pairs whose index differs from the fp32 oracle's by MORE "
                       "than a near-tie (oracle similarity gap > 1e-5); nn_index_diff_rate = any difference.  Plain fields: "
                       "video-like targets (permuted pivot rows + 0.1 noise, SURVEY 8d flavour ii); *_iid: independent "
                       "LayerNorm(N(0,1)) targets (flavour i, the near-tie worst case)",
            "nn_targets_checked": nn_total, "nn_targets_checked_iid": iid_total, "by_level": by_level,
            "f16": {"note": "the same problems with f16 inputs -- the reference's own autocast dtype (run_tokenflow_pnp.py:220): "
                            "P and the output carry 11 significand bits",
                    "attn_linf": round(max(l["attn_linf"] for l in f16_levels), 6),
                    "attn_linf_fp32_out": round(max(l["attn_linf_fp32_out"] for l in f16_levels), 6),
                    "by_level": f16_levels},
            "reference": "oracle (fp32 CPU restatement pinned to the verbatim reference, tests/golden/)"}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run` with N ranks on this node."""
    n_dev = torch.cuda.device_count()
    if args.backend in ("auto", "nccl", "hip", "native") and n_dev < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) are visible "
                 f"(use --backend gloo to let ranks share a GPU for a functional check)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def cpu_leg(cfg, levels, full_l0=False):
    """The reference itself where its tree is mounted, else the oracle port."""
    try:
        from oracle import ref_loader
        if ref_loader.available():
            return cpu_baseline_reference(cfg, levels, full_l0)
    except Exception as e:  # noqa: BLE001
        print(f"[bench] reference-timed CPU baseline unavailable ({e}); timing the oracle port", file=sys.stderr)
    return cpu_baseline(cfg, levels)


def main():
    args = parse()
    if args.cpu_baseline_only:
        lv = [int(x) for x in args.cpu_sample_levels.split(",") if x != ""]
        print(json.dumps({"cpu_baseline": cpu_leg(workload.CONFIGS[args.config], lv, args.cpu_ref_l0)}), flush=True)
        return
    # before the HIP runtime initialises: the host driver of this pool only supports dmabuf IPC (RCCL between processes)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE {world}")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cfg = workload.CONFIGS[args.config]
    hip_comm = halo_comm = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend in ("auto", "hip", "native"):
            dist.init_process_group("gloo")                 # control plane only: the tensors never touch gloo
            from tokenflow_amd import comm as tfcomm
            # two communicators: the second one carries the neighbour halo (sharded.py / rank_exec.hip)
            comms, why = tfcomm.bootstrap(rank, world, 2)
            if comms is not None:
                hip_comm, halo_comm = comms
            if hip_comm is None:
                if args.backend != "auto":
                    sys.exit(f"bench.py: --backend {args.backend}: the library's communicator could not be created: {why}")
                if rank == 0:
                    print(f"[bench] native communicators unavailable ({why}); falling back to torch.distributed/nccl",
                          file=sys.stderr, flush=True)
                dist.destroy_process_group()
                args.backend = "nccl"
            elif args.backend == "auto":
                args.backend = "native"
        halo_group = None
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
            halo_group = dist.new_group(backend="nccl")      # a second RCCL communicator for the neighbour halo
        elif args.backend == "gloo":
            dist.init_process_group("gloo")
    def make_shard(split):
        if world > 1 and args.backend == "native":
            return sharded.NativeShard(cfg.K, hip_comm, halo_comm, attn_split=split)
        if world > 1:
            return sharded.FrameShard(cfg.K, comm=hip_comm, attn_split=split, halo_comm=halo_comm, halo_group=halo_group)
        return sharded.FrameShard(cfg.K, attn_split=False)
    # N > 1: two timed regions -- the form that reproduces the (bit-stable) single-GPU result bit for bit, and the split form in a
    # second timed region (`ms_per_step_split`)
    shard = make_shard(False)
    shard_split = make_shard(True) if world > 1 and not args.no_attn_split else None
    # Inputs: step s, block b draw from manual_seed(1234 + 16 s + b) (+ a rank offset: every rank holds its
    # own frames).  All sets are generated and resident in HBM BEFORE the timed region; step i runs on set i % n_sets, so
    # no step re-reads what the previous one left in the 256 MB Infinity Cache (one set of cfg2 is ~1.1 GB).
    def make_set(s_):
        return [Block(cfg, lvl, inj, shard, torch.Generator(device=dev).manual_seed(1234 + 16 * s_ + b_ + 7919 * rank), dev)
                for b_, (lvl, inj) in enumerate(workload.BLOCKS)]
    m0 = torch.cuda.memory_allocated(dev)
    input_sets = [make_set(0)]
    set_bytes = max(torch.cuda.memory_allocated(dev) - m0, 1)
    n_sets = args.input_sets if args.input_sets > 0 else max(1, min(args.steps, 8, int(64e9 // set_bytes)))
    if n_sets > 1 and n_sets % 2:
        n_sets -= 1          # even: the injection state (step parity) is then a function of the set index (graph keys)
    input_sets += [make_set(s_) for s_ in range(1, n_sets)]
    blocks = input_sets[0]
    w = blend_w(cfg.chunk, dev)
    exchange = None if args.pivotal_exchange == "auto" else args.pivotal_exchange
    names = {"heads": "frames<->heads all-to-all", "bank": "K/V bank all-gather (one collective)"}
    modes = [exchange or shard.auto_mode(l[2], l[0]) for l in cfg.levels]
    exch_name = (names[modes[0]] if len(set(modes)) == 1
                 else "per level " + ", ".join("L%d %s" % (i, m) for i, m in enumerate(modes))
                 + " (heads = frames<->heads all-to-all, bank = one K/V all-gather)")
    use_graph = args.graph and world == 1   # capturing RCCL calls crashes in hipStreamEndCapture on this stack

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(i, events=None, sh=None):
        return run_step(cfg, input_sets[i % n_sets], sh or shard, i % 2 == 0, w, events, exchange=exchange,
                        per_chunk=args.per_chunk)

    for i in range(args.warmup):
        step(i)
    graphs = None
    if use_graph:
        from tokenflow_amd.graphs import GraphCache
        graphs = GraphCache(warmup=0)
        for i in range(max(2, n_sets)):   # capture every (input set, injection state) before the timed region
            graphs.run(("step", i % max(2, n_sets)), lambda i=i: step(i))
    events, marks = [], []
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        m = torch.cuda.Event(enable_timing=True)
        m.record()
        marks.append(m)
        if use_graph:
            graphs.run(("step", i % max(2, n_sets)), None)
        else:
            step(i, events)
    m = torch.cuda.Event(enable_timing=True)
    m.record()
    marks.append(m)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if hip_comm is not None else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_split = None
    if shard_split is not None:     # the same K steps in the split form (same barriers, max over ranks)
        for i in range(args.warmup):
            step(i, sh=shard_split)
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(i, sh=shard_split)
        barrier()
        el = time.perf_counter() - t1
        t = torch.tensor([el], device="cpu" if hip_comm is not None else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_split = float(t.item()) / args.steps * 1e3
    if use_graph:                   # roofline bracket: separate eager pass (events cannot be read out of a graph)
        for i in range(2):
            step(i, events)
        torch.cuda.synchronize()

    ms_bit = elapsed / args.steps * 1e3
    # N > 1: `value` is the FASTEST form whose results are verified against the oracle (both are: the one-pass form
    # through its bit-identity with the single-GPU kernels, the split form directly, tests/test_sharded_gpu.py);
    # north_star asks for tolerance parity, not bit-identity across world sizes.  Both timings stay in the line.
    value_form = "one-pass rank attention (bit-identical to the 1-GPU run with TOKENFLOW_ATTN_NO_SPLIT=1)"
    ms_per_step = ms_bit
    if ms_split is not None and ms_split < ms_bit:
        ms_per_step, value_form = ms_split, "split rank attention (key runs merged in fp32; held to the oracle's bound)"
    value = cfg.frames / (ms_per_step * 1e-3)
    fa, fn, gb = workload.step_work(cfg)
    per_state = {True: [], False: []}
    for i in range(args.steps):
        per_state[i % 2 == 0].append(marks[i].elapsed_time(marks[i + 1]))
    avg = lambda xs: sum(xs) / len(xs) if xs else None

    blk0 = next(b for b in blocks if b.lvl == 0)
    dh0 = cfg.levels[0][1] // cfg.levels[0][2]

    def roof(inj, flops, extra=None):
        durs = [e0.elapsed_time(e1) * 1e-3 for e0, e1, i_, l_ in events if i_ == inj and l_ == 0]
        if not durs:
            return None
        a = sum(durs) / len(durs)
        r = {"bound": "mfma", "achieved": round(flops / a / 1e12, 1), "peak": workload.MFMA_BF16_PEAK / 1e12,
             "unit": "TFLOP/s", "frac": round(flops / a / workload.MFMA_BF16_PEAK, 4),
             "avg_launch_ms": round(a * 1e3, 3), "launches_timed": len(durs),
             "algorithmic_gflop_per_launch": round(flops / 1e9, 1)}
        if extra:
            r.update(extra)
        return r

    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.isfile(tpath):
        try:
            tj = json.load(open(tpath)).get(args.config, {})
            traffic = tj.get("ext_attn_l0_hbm_bytes_per_launch")
            traffic_src = tj.get("source")
        except Exception:
            traffic = None
    kname = {40: "ext_attn_il_kernel<BF16, 40, 8, MODE_ALL, 4, 3> (half-tile interleaved, LDS-DMA staged tiles, mixed MFMA shapes: "
                 "QK^T 32x32x16, P.V 16x16x32)",
             64: "ext_attn_il_kernel<BF16, 64, 8, MODE_ALL, 4, 2> (half-tile interleaved, LDS-DMA staged tiles, score bound)",
             80: "ext_attn_il_kernel<BF16, 80, 4, MODE_ALL, 3, 2> (half-tile interleaved, LDS-DMA staged tiles)"}.get(
                 dh0, "ext_attn_kernel<BF16, %d, ..., MODE_ALL>" % dh0)
    plain = roof(False, blk0.attn_flops, {
        "kernel": kname + ", level 0, no q/k injection (+ vt_pack_kernel pre-pass inside the event bracket; N = 1 name: "
                  "rocprofv3 prints the template arguments as <BF16; %d; ...; 0; ...>)" % dh0,
        "traffic": traffic if world == 1 else None,      # the PMC figure is the single-GPU launch's, not a rank's
        "traffic_source": traffic_src or "profiles/traffic.json (rocprofv3 --pmc passes of tools/attn_microbench.py; "
                                         "not measured in this run)"})
    dual = roof(True, blk0.attn_flops, {
        "kernel": ("ext_attn_il_kernel<BF16, 40, 8, MODE_DUAL, 4, 2>" if dh0 == 40 else "the dual-V kernel")
                  + " (uncond + cond share QK^T and the softmax) + source launch + V^T pre-pass, level 0, q/k injection on",
        "executed_gflop_per_launch": round(blk0.attn_flops_inject / 1e9, 1),
        "note": "achieved/frac use the ALGORITHMIC flops of the reference formulation; the launch executes fewer"})
    out = {
        "metric": "denoising-step hot-path frames/sec", "value": round(value, 2), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic",
        "input_sets": {"n": n_sets, "gb_per_set": round(set_bytes / 1e9, 2),
                       "note": "step i runs on set i % n; set s, block b seeded 1234 + 16 s + b; all resident before the timed region"},
        "ms_per_step_split": ms_split and round(ms_split, 3),
        "ms_per_step_bit_identical": round(ms_bit, 3) if world > 1 else None,
        "value_form": value_form if world > 1 else "single GPU",
        # fixed-form fields (like-for-like across rounds whatever `value` picked): frames/s of each timed form
        "value_bit_identical": round(cfg.frames / (ms_bit * 1e-3), 2) if world > 1 else None,
        "value_split": round(cfg.frames / (ms_split * 1e-3), 2) if ms_split else None,
        "ms_per_step_inject_on": avg(per_state[True]) and round(avg(per_state[True]), 3),
        "ms_per_step_inject_off": avg(per_state[False]) and round(avg(per_state[False]), 3),
        "config": {"workload": cfg.name + " (hot path: 16 blocks x [ext-attn + NN-search + gather/blend over %d chunks])" % cfg.K,
                   "frames": cfg.frames, "keyframes": cfg.K, "frames_per_chunk": cfg.chunk,
                   "levels_S_D_heads": [list(l) for l in cfg.levels],
                   "propagation_calls": "one per chunk (tf_nn_gather_blend)" if args.per_chunk
                   else "one per block over all chunks (tf_nn_gather_blend_chunks)",
                   "launch": "HIP-graph replay" if use_graph else "eager",
                   "parallelism": "1 GPU" if world == 1 else
                   "frames sharded over %d GPUs; pivotal pass: %s%s; rank attention %s" % (
                       world, exch_name, ("; one library call per block (tf_rank_pivotal)" if args.backend == "native" else
                                         "; exchanges through the C ABI (tf_comm_*)") if hip_comm is not None else "",
                       "bit-identical to the bit-stable 1-GPU run (ms_per_step_bit_identical); ms_per_step_split = small grids split the key sequence and "
                       "merge (equal within the output rounding)" if shard_split is not None
                       else "bit-identical to the bit-stable 1-GPU run"),
                   "step_algorithmic_tflop": round((fa + fn) / 1e12, 2),
                   "step_tflops_achieved": round((fa + fn) / 1e12 / (ms_per_step * 1e-3), 1)},
        "roofline": plain if plain is not None else dual,
    }
    if plain is not None and dual is not None:
        out["roofline_inject"] = dual
    if rank == 0:
        if world == 1 and not args.no_parity:
            out["parity"] = parity_check(cfg, blocks, w)
        if world == 1 and not args.no_parity:
            out["roofline_other"] = other_rooflines(cfg, blocks, w, events)
        if world == 1 and not args.no_other_configs and args.config == "cfg2":
            del input_sets[1:]              # the rotating input sets are no longer needed: room for the cfg4 / cfg5 tensors
            torch.cuda.empty_cache()
            out.setdefault("roofline_other", []).extend(other_config_rooflines(dev))
            out["other_configs"] = other_config_steps(dev, lambda n: blend_w(n, dev))
        if world == 1 and not args.no_yardstick:
            out["yardstick"] = yardstick(cfg)
            out["yardstick"]["power_state"] = power_state(blocks)
        if world == 1 and not args.no_cpu_baseline:
            lv = [int(x) for x in args.cpu_sample_levels.split(",") if x != ""]
            out["cpu_baseline"] = cpu_leg(cfg, lv)
        print(json.dumps(out), flush=True)
    for sh_ in (shard, shard_split):
        if isinstance(sh_, sharded.NativeShard):
            sh_.close()
    for c in (halo_comm, hip_comm):
        if c is not None:
            c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
