#!/usr/bin/env python
"""Time one denoising step THROUGH THE DROP-IN HOOK API (tokenflow_utils.register_* / set_tokenflow on
duck-typed diffusers blocks, tests/fake_diffusers.py) at a BASELINE geometry, with the layers that are not
part of the path (q/k/v/out projections, cross-attention, feed-forward) replaced by identities, so that what
is timed is the hook layer itself: norm1, dtype casts, Python, and the HIP ops.  Compare with bench.py, which
calls the ops directly on pre-made tensors.

    python tools/hooks_bench.py [cfg2] [steps] [--proj] [--graph] [--all-chunks]
      --proj        keep attn1's q/k/v/out Linear layers (default: identities)
      --graph       replay every pass from a HIP graph (tokenflow_amd.graphs.GraphCache): one host call per pass
      --all-chunks  ONE propagation pass over all chunks (register_batch_idx(model, range(K))) instead of K passes
      --ranks W [--rank r] [--wire-less]
                    ONE rank of a W-GPU frame-sharded run through the hook API (register_frame_shard with a NativeShard on
                    the library's loopback transport: every exchange a same-size local copy; --wire-less: no copies at
                    all): the rank's Kl keyframes in the pivotal pass, its own chunks in the chunk passes.  Compare with
                    tools/rank_step_microbench.py --native (the same rank through bench.py's direct op calls).  With
                    --graph the rank's passes replay from HIP graphs (loopback exchanges are stream-ordered copies; the
                    pivotal pass ends with hooks.join_frame_shard): the hook-level rank step without Python issue time.
                    --ranks 1: a WORLD-1 FrameShard registered through register_frame_shard (bit-stable attention mode
                    unless --split): what the per-grid kernel choice it gives up costs on one GPU.
      --breakdown   re-run this command under `rocprofv3 --kernel-trace --memory-copy-trace` and split the GPU time of a
                    step into (i) the library's own launches (tf_*), (ii) what the hook layer adds around them (copies and
                    dtype casts: torch copy kernels and device-to-device copies), (iii) the block's own layers (everything
                    else: the residual adds `ff_output + hidden_states`, tokenflow_utils.py:427) -- per step, with the
                    largest kernels of (ii) and (iii) named.  Compare (i) + (ii) on a rank with
                    tools/rank_step_microbench.py --native --no-copies.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tokenflow_utils as tfu  # noqa: E402
from tests import fake_diffusers as fd  # noqa: E402
from tokenflow_amd import workload  # noqa: E402


class _Id(torch.nn.Module):
    def forward(self, x, *a, **k):
        return x


def build(cfg, dev, dtype):
    holder = torch.nn.Module()
    holder.unet = torch.nn.Module()
    blocks = []
    for i, (lvl, injected) in enumerate(workload.BLOCKS):
        S, D, h = cfg.levels[lvl]
        blk = fd.BasicTransformerBlock(D, h, cross_dim=32)
        if not PROJ:
            blk.attn1.to_q, blk.attn1.to_k, blk.attn1.to_v = _Id(), _Id(), _Id()
            blk.attn1.to_out = torch.nn.ModuleList([_Id(), _Id()])
        blk.attn2 = None
        blk.ff = _Id()
        setattr(holder.unet, f"blk{i}", blk)
        blocks.append((blk, lvl, injected))
    holder.to(dev).to(dtype).eval()
    return holder, blocks


RAW_ARGV = list(sys.argv)
PROJ = "--proj" in sys.argv      # keep the real q/k/v/out Linear layers of attn1 (default: identities)
GRAPH = "--graph" in sys.argv
ALL_CHUNKS = "--all-chunks" in sys.argv


def _opt(name, default):
    if name in sys.argv:
        i = sys.argv.index(name)
        v = sys.argv[i + 1]
        del sys.argv[i:i + 2]
        return int(v)
    return default


TF_KERNELS = ("ext_attn", "attn_fused", "attn_merge", "vt_pack", "nn_search", "gather_blend", "gbn_", "layer_norm",
              "add_layer_norm", "pivot_inv_norm", "head_pack", "head_unpack", "inject_copy", "ddim_step", "wire_hold")


def breakdown():
    """Re-run under rocprofv3 and classify the trace (see the module docstring)."""
    import glob
    import re
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    # eager on purpose: a graph capture runs an extra eager warm-up of every pass, which would blur the step count the sums
    # are divided by; kernel durations do not depend on how a launch was issued
    args = [a for a in sys.argv[1:] if a not in ("--breakdown", "--graph")]
    out = tempfile.mkdtemp(prefix="hooks_bd_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--memory-copy-trace", "-d", out, "--", sys.executable,
                        os.path.abspath(__file__)] + args, cwd="/tmp", env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("hooks path")]
    print(line[-1] if line else r.stdout[-400:] + r.stderr[-400:])
    m = re.search(r"\[steps run: (\d+)\]", r.stdout)
    n_steps = int(m.group(1)) if m else 1
    dbs = glob.glob(os.path.join(out, "**", "*_results.db"), recursive=True)
    if not dbs:
        print("no rocprofv3 database produced:", r.stderr[-300:])
        return
    cur = sqlite3.connect(dbs[0]).cursor()
    cls = {"tf": {}, "norm": {}, "hook": {}, "model": {}}
    for name, dur in cur.execute("select name, duration from kernels"):
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        if "distribution_elementwise" in short or "randperm" in short:
            continue                       # this tool's own input generation (torch.randn before the first step)
        if "layer_norm" in short and "at::native" not in short:
            c = "norm"                     # the block's own LayerNorms, run by the library (tf_layer_norm / tf_add_layer_norm)
        elif any(k in short for k in TF_KERNELS) and "at::native" not in short:
            c = "tf"
        elif "at::native" in short and re.search(r"copy|Copy|cast", short):
            c = "hook"
        else:
            c = "model"
        key = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", short)[:100]
        d = cls[c].setdefault(key, [0, 0.0])
        d[0] += 1
        d[1] += dur
    tabs = [r_[0] for r_ in cur.execute("select name from sqlite_master where type in ('table','view')")]
    for tab in ("memory_copies", "memory_copy"):
        if tab in tabs:
            for (dur,) in cur.execute(f"select end - start from {tab}"):
                d = cls["hook"].setdefault("device-to-device copy (hipMemcpyAsync)", [0, 0.0])
                d[0] += 1
                d[1] += dur
            break
    tot = {c: sum(v[1] for v in cls[c].values()) / n_steps / 1e6 for c in cls}
    print(f"GPU time per step over {n_steps} traced steps (kernel durations summed: launches that overlap on two streams count "
          f"twice, idle gaps not at all): (i) the path's launches {tot['tf']:.3f} ms | (ii) hook-added copies / casts "
          f"{tot['hook']:.3f} ms | (iii) the block's own layers {tot['model'] + tot['norm']:.3f} ms, of which its LayerNorms "
          f"through tf_layer_norm {tot['norm']:.3f} | (i)+(ii) = {tot['tf'] + tot['hook']:.3f} ms")
    for c, title, top in (("tf", "(i) path", 8), ("hook", "(ii) hook-added", 4), ("norm", "(iii) LayerNorm", 2), ("model", "(iii) block's own", 4)):
        for key, (cnt, dur) in sorted(cls[c].items(), key=lambda kv: -kv[1][1])[:top]:
            print(f"   {title}: {dur / n_steps / 1e6:7.3f} ms/step  {cnt / n_steps:6.1f} launches/step  {key}")
    shutil.rmtree(out, ignore_errors=True)


def main():
    if "--breakdown" in sys.argv:
        return breakdown()
    ranks, rank = _opt("--ranks", 1), _opt("--rank", 1)
    wireless = "--wire-less" in sys.argv
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    cfg = workload.CONFIGS[argv[0] if len(argv) > 0 else "cfg2"]
    steps = int(argv[1]) if len(argv) > 1 else 6
    dev, dtype = torch.device("cuda"), torch.bfloat16
    holder, blocks = build(cfg, dev, dtype)
    from tokenflow_amd import hooks
    for blk, lvl, injected in blocks:       # what register_extended_attention_pnp does, per block
        blk.attn1.forward = hooks._make_sa_forward(blk.attn1, pnp=True)
        hooks._set_schedule(blk.attn1, [5] if injected else [])
        blk.attn1.t = 5
    tfu.set_tokenflow(holder)
    K, n = cfg.K, cfg.chunk
    shard, Kq, chunks = None, K, list(range(K))
    if ranks > 1:
        from tokenflow_amd import sharded
        from tokenflow_amd.comm import HipComm
        shard = sharded.NativeShard(K, HipComm.loopback(rank, ranks, copies=not wireless),
                                    HipComm.loopback(rank, ranks, copies=not wireless), attn_split="--one-pass" not in sys.argv)
        hooks.register_frame_shard(holder, shard)
        Kq, chunks = shard.Kl, list(range(shard.kf0, shard.kf0 + shard.Kl))
    elif "--ranks" in RAW_ARGV:      # --ranks 1: a world-1 shard through register_frame_shard
        from tokenflow_amd import sharded
        shard = sharded.FrameShard(K, attn_split="--split" in sys.argv)
        hooks.register_frame_shard(holder, shard)
    g = torch.Generator(device=dev).manual_seed(0)
    xs_piv = [torch.randn(3 * Kq, cfg.levels[l][0], cfg.levels[l][1], generator=g, device=dev, dtype=dtype)
              for _, l, _ in blocks]
    xs_chk = [torch.randn(3 * n, cfg.levels[l][0], cfg.levels[l][1], generator=g, device=dev, dtype=dtype)
              for _, l, _ in blocks]

    if ALL_CHUNKS:     # one pass carries every chunk: frames chunk-major inside each branch
        xs_chk = [x.view(3, 1, n, *x.shape[1:]).expand(3, len(chunks), n, *x.shape[1:]).reshape(3 * len(chunks) * n, *x.shape[1:]).contiguous()
                  for x in xs_chk]
    cache = None
    if GRAPH:
        from tokenflow_amd.graphs import GraphCache
        cache = GraphCache()

    def pivotal_pass(*xs):
        tfu.register_pivotal(holder, True)
        out = [blk(x) for (blk, _, _), x in zip(blocks, xs)]
        if shard is not None and cache is not None:
            hooks.join_frame_shard(holder)      # a captured pass joins the halo stream before it ends
        return out

    def chunk_pass(c, *xs):
        tfu.register_pivotal(holder, False)
        tfu.register_batch_idx(holder, range(chunks[0], chunks[-1] + 1) if ALL_CHUNKS else c)
        return [blk(x) for (blk, _, _), x in zip(blocks, xs)]

    def step(inject_on):
        for blk, _, injected in blocks:
            blk.attn1.t = 5 if inject_on else 7
        # the reference runs its UNet passes under autocast (run_tokenflow_pnp.py:220)
        with torch.no_grad(), torch.autocast("cuda", dtype=dtype):
            if cache is not None:
                # after the capture the passes are fed through the graphs' own static inputs: no per-replay input
                # copies (0.55 GB per pivotal pass, 0.35 GB per chunk pass at cfg2 -- what made round 2's replay
                # numbers 1.8 ms SLOWER than eager; in a UNet the block inputs come from the preceding layers)
                def feed(key, default):
                    try:
                        return cache.inputs(key)
                    except KeyError:
                        return default
                cache.run(("pivotal", inject_on), pivotal_pass, *feed(("pivotal", inject_on), xs_piv))
                for c in ([chunks[0]] if ALL_CHUNKS else chunks):
                    cache.run(("chunk", c), lambda *xs, c=c: chunk_pass(c, *xs), *feed(("chunk", c), xs_chk))
            else:
                pivotal_pass(*xs_piv)
                for c in ([chunks[0]] if ALL_CHUNKS else chunks):
                    chunk_pass(c, *xs_chk)

    for i in range(2):
        step(i % 2 == 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i % 2 == 0)
    t_cpu = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    mode = ("graph replay" if GRAPH else "eager") + (", one pass over all chunks" if ALL_CHUNKS else "") + (", real projections" if PROJ else "")
    if shard is not None and ranks > 1:
        mode += f", rank {rank} of {ranks} (native executor, loopback transport{' without copies' if wireless else ''})"
    elif shard is not None:
        mode += f", world-1 FrameShard registered ({'split' if shard.attn_split else 'bit-stable'} attention mode)"
    print(f"[steps run: {steps + 2}]")
    print(f"hooks path [{mode}], {cfg.name}: {t / steps * 1e3:.2f} ms/step ({cfg.frames * steps / t:.0f} frames/s); "
          f"host-side issue time {t_cpu / steps * 1e3:.2f} ms/step")


if __name__ == "__main__":
    main()
