"""Generate tests/golden/*.pt by EXECUTING THE VERBATIM REFERENCE (test infrastructure).

Run in the build container, where /root/reference is mounted:

    python oracle/make_golden.py

The reference has no tests or golden vectors of its own, so the
pin is the reference code itself: its hook closures are imported unmodified
(oracle/ref_loader.py), driven through duck-typed diffusers stand-ins
(tests/fake_diffusers.py) on seeded CPU fp32 inputs, and inputs + outputs are
saved as small fixtures.  `/root/reference` does not exist on the GPU box, so
these files are what travels.

Fixtures
  attn_core.pt   sa_forward closures (tokenflow_utils.py:114-199, 224-281) with
                 to_q/to_k/to_v returning given tensors and to_out = identity.
  propagate.pt   batch_cosine_sim + argmax (util.py:61-69; tokenflow_utils.py:335-343)
                 and the gather/blend/residual of TokenFlowBlock.forward (361-397),
                 obtained by running the real TokenFlowBlock with identity sub-modules.
  blocks.pt      all hooks installed on a 16-block fake UNet
                 (register_extended_attention_pnp, register_conv_injection,
                 set_tokenflow, register_pivotal, register_batch_idx, register_time):
                 per-block outputs of the pivotal pass and of chunks 0..2, and the
                 patched resnet forward.  Module weights come from a seed; a
                 checksum guards against RNG drift.
  inversion.pt   Preprocess.ddim_inversion / ddim_sample (preprocess.py:198-261), cut out of the reference's
                 syntax tree and executed unchanged on a stand-in model: the latents files written
                 (names + contents), the inverted and the reconstructed latents.  fp32, and float16 as the
                 reference runs them (preprocess.py:195).
  driver.pt      the hook-facing methods of the reference DRIVERS -- TokenFlow.init_method / denoise_step /
                 batched_denoise_step, run_tokenflow_pnp.py:195-239 and run_tokenflow_sdedit.py:154-193, cut out of
                 the syntax tree and executed unchanged (oracle/driver_cut.py) -- over the verbatim hooks on the
                 runnable stand-in UNet (tests/driver_seam.py), 3 timesteps (q/k + feature injection, feature
                 injection only, none): the hook-call trace, the `pivotal_idx` draws, the UNet's noise prediction
                 and every transformer block's output of every UNet call, the latents after each step.
  adazero.pt     TokenFlowBlock.forward on an AdaLayerNormZero block (use_ada_layer_norm_zero:
                 gate_msa on the cached / selected attention outputs, 362-366; scale/shift/gate
                 on the feed-forward, 417-424): pivotal pass and chunks 0..K-1.
"""
import os
import zlib
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from tests import fake_diffusers as fd  # noqa: E402
from oracle.golden_util import digest  # noqa: E402
from oracle import golden_cases as gc  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


class _Fixed(nn.Module):
    """to_q / to_k / to_v stand-in that returns a clone of a fixed tensor."""

    def __init__(self, t):
        super().__init__()
        self.t = t

    def forward(self, x):
        return self.t.clone()


class CoreAttention(fd.Attention):
    def __init__(self, q, k, v, heads):
        D = q.shape[-1]
        super().__init__(D, heads)
        self.to_q, self.to_k, self.to_v = _Fixed(q), _Fixed(k), _Fixed(v)
        self.to_out = nn.Identity()


class _OneBlock(nn.Module):
    def __init__(self, block):
        super().__init__()
        self.unet = nn.Module()
        self.unet.blk = block
        # register_extended_attention* also index up_blocks[1..3] (208-214): give
        # them the same block so the re-wrap lands on it too.
        holder = nn.Module()
        holder.transformer_blocks = nn.ModuleList([block])
        stage = nn.Module()
        stage.attentions = nn.ModuleList([holder, holder, holder])
        self.unet.up_blocks = nn.ModuleList([nn.Module(), stage, stage, stage])


def gen_attn_core(tfu):
    out = {}
    for name, (K, S, h, d, sched, t) in gc.ATTN_CASES.items():
        q, k, v = gc.attn_inputs(name)
        D = h * d
        blk = fd.BasicTransformerBlock(D, h)
        blk.attn1 = CoreAttention(q, k, v, h)
        tfu.register_extended_attention_pnp(_OneBlock(blk), sched)
        blk.attn1.t = t
        with torch.no_grad():
            o_pnp = blk.attn1.forward(torch.zeros(3 * K, S, D))
        # sdedit variant (224-281): never injects
        blk2 = fd.BasicTransformerBlock(D, h)
        blk2.attn1 = CoreAttention(q, k, v, h)
        tfu.register_extended_attention(_OneBlock(blk2))
        with torch.no_grad():
            o_sde = blk2.attn1.forward(torch.zeros(3 * K, S, D))
        out[name] = dict(input_checksum=gc.checksum(q, k, v), out_pnp=digest(o_pnp, 5),
                         out_sdedit=digest(o_sde, 7))
    return out


class _ToFloat(nn.Module):
    def forward(self, x):
        return x.float()


class _Zero(nn.Module):
    def forward(self, x):
        return torch.zeros_like(x)


class _IdBlock(fd.BasicTransformerBlock):
    """TokenFlowBlock host whose norm1 is an exact upcast (like an autocast
    LayerNorm: fp32 out whatever the stream dtype) and whose tail (attn2, ff) is
    switched off, so forward() returns attn_output + hidden_states only."""

    def __init__(self, D):
        nn.Module.__init__(self)
        self.only_cross_attention = False
        self.use_ada_layer_norm = False
        self.use_ada_layer_norm_zero = False
        self.norm1 = _ToFloat()
        self.attn1 = None
        self.attn2 = None
        self.norm3 = _Zero()
        self.ff = nn.Identity()


def gen_propagate(tfu, util):
    out = {}
    for name, (K, n, S, D, dt) in gc.PROP_CASES.items():
        piv, kf_out, hidden = gc.prop_inputs(name)
        case = dict(input_checksum=gc.checksum(piv, kf_out, *hidden), chunks={})
        blk = _IdBlock(D)
        blk.__class__ = tfu.make_tokenflow_attention_block(blk.__class__)
        blk.pivot_hidden_states = piv
        blk.kf_attn_output = kf_out
        blk.pivotal_pass = False
        for bi in range(K):
            blk.batch_idx = bi
            with torch.no_grad():
                res = blk.forward(hidden[bi].clone())      # = attn_output + hidden (+0 from ff(zero))
                ids = [bi] if bi == 0 else [bi, bi - 1]
                tgt = hidden[bi].float().view(3, n, S, D)[0]
                sim = util.batch_cosine_sim(tgt.reshape(-1, D), piv[0][ids].reshape(-1, D))
                idx = [c.argmax(-1) for c in sim.chunk(len(ids), dim=1)]
            case["chunks"][bi] = dict(out=digest(res, 17), out_dtype=str(res.dtype),
                                      idx=[i.to(torch.int16) for i in idx])
        out[name] = case
    return out


def gen_blocks(tfu):
    cfg = gc.BLOCKS_CFG
    torch.manual_seed(cfg["seed"])
    pipe = fd.FakePipeline(dims=cfg["dims"], heads=cfg["heads"], cross_dim=cfg["cross_dim"]).eval()
    tfu.register_extended_attention_pnp(pipe, cfg["schedule"])
    tfu.register_conv_injection(pipe, cfg["conv_schedule"])
    tfu.set_tokenflow(pipe.unet)
    out = dict(weights_checksum=gc.checksum(*pipe.parameters()), runs={})
    blocks = [b for _, b in pipe.unet.transformer_blocks_in_order()]
    for t in cfg["timesteps"]:     # qk+conv inject / conv only / none
        tfu.register_time(pipe, t)
        inp = gc.blocks_inputs(t)
        run = dict(input_checksum=gc.checksum(*inp["pivotal"], *sum(inp["chunks"], [])), pivotal=[], chunks=[])
        with torch.no_grad():
            tfu.register_pivotal(pipe, True)
            for blk, x in zip(blocks, inp["pivotal"]):
                run["pivotal"].append(digest(blk(x, encoder_hidden_states=inp["enc"]), 61))
            tfu.register_pivotal(pipe, False)
            for c in range(cfg["n_chunks"]):
                tfu.register_batch_idx(pipe, c)
                run["chunks"].append([digest(blk(x, encoder_hidden_states=inp["enc_n"]), 61)
                                      for blk, x in zip(blocks, inp["chunks"][c])])
                # the state a propagation pass leaves behind (tokenflow_utils.py:361-363): the selected keyframe outputs
                run.setdefault("chunk_attn_state", []).append([digest(blk.attn_output, 61) for blk in blocks])
            run["resnet"] = digest(pipe.unet.up_blocks[1].resnets[1](inp["res_x"], inp["res_temb"]), 3)
        out["runs"][t] = run
    return out


def gen_adazero(tfu):
    """TokenFlowBlock on an AdaLayerNormZero block: pivotal pass, then chunks 0..K-1 (gate_msa applied to the
    selected keyframe outputs before the gather, tokenflow_utils.py:362-366)."""
    blk = gc.adazero_block()
    holder = _OneBlock(blk)
    tfu.register_extended_attention_pnp(holder, [])
    tfu.set_tokenflow(holder.unet)
    blk.attn1.t = blk.attn2.t = 7
    inp = gc.adazero_inputs()
    out = dict(weights_checksum=gc.checksum(*blk.parameters()),
               input_checksum=gc.checksum(inp["pivotal"], *inp["chunks"]), chunks=[])
    with torch.no_grad():
        tfu.register_pivotal(holder, True)
        out["pivotal"] = digest(blk(inp["pivotal"], encoder_hidden_states=inp["enc"], timestep=inp["timestep"]), 7)
        tfu.register_pivotal(holder, False)
        for c in range(gc.ADAZERO_CFG["K"]):
            tfu.register_batch_idx(holder, c)
            out["chunks"].append(digest(blk(inp["chunks"][c], encoder_hidden_states=inp["enc_n"],
                                            timestep=inp["timestep"]), 7))
    return out


def load_reference_inversion():
    """`Preprocess.ddim_inversion` and `Preprocess.ddim_sample` of the VERBATIM reference as plain functions.
    preprocess.py imports diffusers / transformers at module level and cannot be imported here, so the two method
    definitions are cut out of its syntax tree and compiled unchanged (decorators included) in a namespace that
    holds what their bodies use: torch, os and a pass-through tqdm."""
    import ast
    src = open(os.path.join(ref_loader.REF_ROOT, "preprocess.py")).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Preprocess")
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("ddim_inversion", "ddim_sample")]
    assert len(fns) == 2
    ns = {"torch": torch, "os": os, "tqdm": lambda it: it}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "preprocess.py", "exec"), ns)
    return ns["ddim_inversion"], ns["ddim_sample"]


def gen_inversion():
    """The verbatim inversion / reconstruction loops (preprocess.py:198-261) on the stand-in model: the files they
    write and the tensors they return.  fp32 latents, and (keys with the suffix `_f16`) the float16 latents the
    reference actually inverts (preprocess.py:195 `.to(torch.float16)`), which pins the 16-bit rounding points of the
    update to the reference itself."""
    import tempfile
    ref_inv, ref_sample = load_reference_inversion()
    model = gc.InversionModel()
    out = {}
    for dtype, sfx in ((torch.float32, ""), (torch.float16, "_f16")):
        latents, cond = gc.inversion_inputs(dtype)
        out["input_checksum" + sfx] = gc.checksum(latents, cond)
        out["files" + sfx] = {}
        with tempfile.TemporaryDirectory() as d:
            os.makedirs(os.path.join(d, "latents"))
            save_ts = model.scheduler.timesteps[::2]
            inv = ref_inv(model, cond, latents.clone(), d, gc.INVERSION_CFG["batch_size"], save_latents=True,
                          timesteps_to_save=save_ts)
            assert inv.dtype == dtype
            for f in sorted(os.listdir(os.path.join(d, "latents"))):
                out["files" + sfx][f] = digest(torch.load(os.path.join(d, "latents", f)), 3)
            out["inverted" + sfx] = digest(inv, 1)
            out["reconstructed" + sfx] = digest(ref_sample(model, inv.clone(), cond, gc.INVERSION_CFG["batch_size"]), 1)
    return out


def gen_driver(tfu):
    """The verbatim driver methods over the verbatim hooks, CPU fp32 (the decorator's CUDA autocast is inert on
    the CPU).  One record per driver kind."""
    import tempfile
    import warnings
    from oracle import driver_cut
    from tests import driver_seam as ds
    out = {}
    for kind in ("pnp", "sdedit"):
        log = []
        methods = driver_cut.load_reference_driver(kind, ds.traced(tfu, log))
        with tempfile.TemporaryDirectory() as d, warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rec = ds.run_driver(kind, methods, d)
        rec["trace"] = log
        out[kind] = rec
    return out


def main():
    tfu, util = ref_loader.load()
    os.makedirs(GOLDEN, exist_ok=True)
    torch.save(gen_attn_core(tfu), os.path.join(GOLDEN, "attn_core.pt"))
    torch.save(gen_propagate(tfu, util), os.path.join(GOLDEN, "propagate.pt"))
    torch.save(gen_blocks(tfu), os.path.join(GOLDEN, "blocks.pt"))
    torch.save(gen_adazero(tfu), os.path.join(GOLDEN, "adazero.pt"))
    torch.save(gen_inversion(), os.path.join(GOLDEN, "inversion.pt"))
    torch.save(gen_driver(tfu), os.path.join(GOLDEN, "driver.pt"))
    for f in sorted(os.listdir(GOLDEN)):
        print(f, os.path.getsize(os.path.join(GOLDEN, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
