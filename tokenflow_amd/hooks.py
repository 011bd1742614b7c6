"""The reference's hook API for the per-step hot path, on MI355X kernels.

Same function names, signatures, installation style and per-module state as
omerbt/TokenFlow `tokenflow_utils.py` (paths below are into that repository), so the
reference's `run_tokenflow_pnp.py` / `run_tokenflow_sdedit.py` import this through the
drop-in module `tokenflow_utils.py` at the repo root, unmodified.

What differs from the reference is only *how* the tensors are computed:

* `sa_forward.forward` (114-199 / 224-281): the bank replication, head permutes, per-head
  bmm/softmax/bmm loops and re-layout are one fused HIP launch (`ops.ext_attn`); injection
  (124-130) is pointer aliasing instead of four in-place slice copies.
* `TokenFlowBlock.forward` propagation branch (329-397): cosine similarity + argmax is
  `ops.nn_search` (the similarity matrix is never materialised), gather + blend + residual
  is `ops.gather_blend`.
* `conv_forward.forward` (86-91): the two slice copies are `ops.inject_copy_`, applied by a forward hook on the
  resnet's `conv2` instead of a transcription of diffusers' `ResnetBlock2D.forward` (51-98).
* `t in injection_schedule` on a device tensor (86,124) costs a host sync per call in the
  reference; the schedule is converted to a Python set once at registration.
* `register_pivotal` / `register_batch_idx` (7-17) walk `named_modules()` of the whole
  pipeline on every call in the reference; the matching blocks are cached per model.
* `load_source_latents_t` (43-47) re-reads the file from disk on every call in the
  reference; files are cached by (path, mtime).

There is no CPU / eager fallback: the ops raise on CPU tensors or when the HIP library is
missing.
"""
import collections
import logging
import os
import sys
from typing import Type

import torch

from . import ops

__all__ = [
    "register_pivotal", "register_batch_idx", "register_time", "load_source_latents_t",
    "register_conv_injection", "register_extended_attention_pnp", "register_extended_attention",
    "make_tokenflow_attention_block", "set_tokenflow", "isinstance_str", "batch_cosine_sim",
    "register_frame_shard", "join_frame_shard",
]


def isinstance_str(x: object, cls_name: str) -> bool:
    """Class-*name* match over the MRO (util.py:46-58): lets the hooks patch diffusers
    modules without importing diffusers."""
    return any(c.__name__ == cls_name for c in type(x).__mro__)


def batch_cosine_sim(x, y):
    """Public export of util.py:61-69, kept for API compatibility.  The hooks do NOT call
    it: the product path is ops.nn_search, which never materialises this matrix."""
    if type(x) is list:
        x = torch.cat(x, dim=0)
    if type(y) is list:
        y = torch.cat(y, dim=0)
    return (x / x.norm(dim=-1, keepdim=True)) @ (y / y.norm(dim=-1, keepdim=True)).T


# --------------------------------------------------------------------------- state setters
def _tokenflow_blocks(model):
    cache = model.__dict__.get("_tf_block_cache")
    if cache is None:
        cache = [m for _, m in model.named_modules() if isinstance_str(m, "BasicTransformerBlock")]
        model.__dict__["_tf_block_cache"] = cache
    return cache


def register_pivotal(diffusion_model, is_pivotal):
    """tokenflow_utils.py:7-11."""
    for module in _tokenflow_blocks(diffusion_model):
        setattr(module, "pivotal_pass", is_pivotal)


def register_batch_idx(diffusion_model, batch_idx):
    """tokenflow_utils.py:13-17."""
    for module in _tokenflow_blocks(diffusion_model):
        setattr(module, "batch_idx", batch_idx)


def register_frame_shard(diffusion_model, shard):
    """Multi-GPU extension (no counterpart in the single-process reference): one process per GPU, each running the
    SAME driver on ITS run of chunks.  `shard` = `tokenflow_amd.sharded.FrameShard` / `NativeShard` (rank r owns the
    keyframes and chunks shard.kf0 .. shard.kf0 + shard.Kl - 1), or None to go back to one process.  With a shard set:
      * the pivotal pass carries the rank's LOCAL keyframes only ([3*Kl, S, D] per block); `attn1` attends to all K
        keyframes through `shard.pivotal_attention` (frames<->heads all-to-all or the bank all-gather), and the block
        hands its last keyframe's pivots / inverse norms / attention output to rank r+1 (the pivots' half is issued
        before the attention and travels under it);
      * the chunk passes name GLOBAL chunk indices (`register_batch_idx(model, c)`, c owned by this rank; a run of the
        rank's chunks for the one-pass form) and read keyframes c and c-1 (tokenflow_utils.py:331-333) from the
        rank's halo-extended caches, the first one waiting for the neighbour's message.
    Results equal the single-process hooks' in the bit-stable mode (TOKENFLOW_ATTN_NO_SPLIT=1) bit for bit (`FrameShard`'s
    default one-pass attention; against the default single-process mode: within the attention's parity bound); every rank must
    draw the same `pivotal_idx` (run_tokenflow_pnp.py:224).  INTEGRATION.md section 3 shows the driver side.
    A world-1 shard (one GPU) runs the bit-stable attention mode too unless it was built with `attn_split=True`: slower
    than the plain single-process hooks by the per-grid kernel choice it gives up (`tools/hooks_bench.py --ranks 1`)."""
    for module in _tokenflow_blocks(diffusion_model):
        module.__dict__["_tf_shard"] = shard
        module.__dict__.pop("_tf_halo", None)
        module.attn1.__dict__["_tf_shard"] = shard


def join_frame_shard(diffusion_model):
    """Order the CURRENT stream behind every neighbour halo of the pivotal pass that is still in flight.  The chunk
    passes do that themselves, block by block, the first time they read a block's halo slot -- which lets the halos
    travel under the rest of the pivotal pass; call this at the end of a pivotal pass that is captured into a HIP
    graph (tokenflow_amd.graphs.GraphCache): a capture must join every stream it forked before it ends, and the chunk
    passes' graphs then contain no wait on another graph's events."""
    for module in _tokenflow_blocks(diffusion_model):
        shard = module.__dict__.get("_tf_shard")
        halo = module.__dict__.get("_tf_halo")
        if shard is not None and halo is not None and halo[3]:
            shard.halo_wait(halo[3])
            module.__dict__["_tf_halo"] = (halo[0], halo[1], halo[2], [])


def _active_shard(module):
    shard = module.__dict__.get("_tf_shard")
    return shard if shard is not None and shard.world > 1 else None


_DOWN = {0: [0, 1], 1: [0, 1], 2: [0, 1]}
_UP = {1: [0, 1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}
_INJECTED_UP = {1: [1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}


def register_time(model, t):
    """tokenflow_utils.py:20-40: `t` on up_blocks[1].resnets[1] and on attn1/attn2 of the 16 blocks."""
    unet = model.unet
    setattr(unet.up_blocks[1].resnets[1], "t", t)
    for res, blocks in _UP.items():
        for b in blocks:
            tb = unet.up_blocks[res].attentions[b].transformer_blocks[0]
            setattr(tb.attn1, "t", t)
            setattr(tb.attn2, "t", t)
    for res, blocks in _DOWN.items():
        for b in blocks:
            tb = unet.down_blocks[res].attentions[b].transformer_blocks[0]
            setattr(tb.attn1, "t", t)
            setattr(tb.attn2, "t", t)
    tb = unet.mid_block.attentions[0].transformer_blocks[0]
    setattr(tb.attn1, "t", t)
    setattr(tb.attn2, "t", t)


_latents_cache = collections.OrderedDict()
_LATENTS_CACHE_ENTRIES = 2      # the current timestep's file (+ one): the driver loads the SAME file C+1 times per step


def load_source_latents_t(t, latents_path):
    """tokenflow_utils.py:43-47.  The reference re-reads the file on each of the C+1 calls of a denoising step
    (run_tokenflow_pnp.py:198,221); here the most recent files are kept, keyed by (path, mtime) -- a bounded
    LRU, so nothing but the current timestep's latents stays resident."""
    latents_t_path = os.path.join(latents_path, f"noisy_latents_{t}.pt")
    assert os.path.exists(latents_t_path), f"Missing latents at t {t} path {latents_t_path}"
    key = (latents_t_path, os.path.getmtime(latents_t_path))
    hit = _latents_cache.get(key)
    if hit is None:
        hit = torch.load(latents_t_path)
        _latents_cache[key] = hit
        while len(_latents_cache) > _LATENTS_CACHE_ENTRIES:
            _latents_cache.popitem(last=False)
    else:
        _latents_cache.move_to_end(key)
    return hit


# --------------------------------------------------------------------------- schedules
def _schedule_set(schedule):
    """One-time conversion so that `t in schedule` never syncs the device."""
    if schedule is None:
        return None
    if isinstance(schedule, torch.Tensor):
        return set(schedule.detach().cpu().reshape(-1).tolist())
    return set(float(s) if isinstance(s, torch.Tensor) else s for s in schedule)


def _injecting(module) -> bool:
    """`schedule is not None and (t in schedule or t == 1000)` (tokenflow_utils.py:86,124)."""
    sched = module.__dict__.get("_tf_schedule_set")
    if module.injection_schedule is None or sched is None:
        return False
    t = module.t
    if isinstance(t, torch.Tensor):
        t = t.item()
    return t in sched or t == 1000


def _set_schedule(module, injection_schedule):
    setattr(module, "injection_schedule", injection_schedule)
    module.__dict__["_tf_schedule_set"] = _schedule_set(injection_schedule)


# --------------------------------------------------------------------------- PnP feature injection
def register_conv_injection(model, injection_schedule):
    """tokenflow_utils.py:49-104.  The reference REPLACES `up_blocks[1].resnets[1].forward` by a transcription of
    diffusers' `ResnetBlock2D.forward` (51-98) with two slice copies behind `conv2` (86-91) -- and inherits that
    version's signature and body.  Here the module keeps ITS OWN forward, whatever diffusers release wrote it, and a
    forward hook on its `conv2` overwrites the uncond / cond activations with the source branch's on conv2's output:
    the same point of the data flow (after conv2, in front of the shortcut add), one broadcast-copy launch
    (`ops.inject_copy_`), same schedule semantics (`t in schedule or t == 1000`, 86).  Installing twice replaces the
    previous hook."""
    conv_module = model.unet.up_blocks[1].resnets[1]

    def after_conv2(_conv2, _inputs, hidden_states):
        if _injecting(conv_module):
            if not hidden_states.is_contiguous():
                hidden_states = hidden_states.contiguous()
            ops.inject_copy_(hidden_states)
        return hidden_states

    prev = conv_module.__dict__.pop("_tf_conv_hook", None)
    if prev is not None:
        prev.remove()
    conv_module.__dict__["_tf_conv_hook"] = conv_module.conv2.register_forward_hook(after_conv2)
    _set_schedule(conv_module, injection_schedule)


# --------------------------------------------------------------------------- extended attention
_announced = False


def _announce():
    """One line on stderr the first time a hook is installed: a run that silently picked up the reference's own
    `tokenflow_utils.py` (the script directory precedes PYTHONPATH on sys.path) never prints it."""
    global _announced
    if _announced:
        return
    _announced = True
    from . import _lib
    lib = _lib.load()                    # fails loudly here, at installation time, if the library is missing
    msg = f"[tokenflow_amd] MI355X HIP hook path active ({_lib.LIB_PATH}, ABI v{lib.tf_abi_version()})"
    logging.getLogger("tokenflow_amd").info(msg)
    if os.environ.get("TOKENFLOW_QUIET", "0") in ("", "0"):
        print(msg, file=sys.stderr, flush=True)


def _fused_qkv(attn, x: torch.Tensor):
    """q, k, v of a self-attention as column slabs of ONE projection `x @ [Wq; Wk; Wv]^T` (row f2 of SURVEY.md
    section 8; tokenflow_utils.py:120-122 issues three Linear calls): one GEMM, one output [3K, S, 3D] that the
    attention kernel reads in place through its row stride, no per-tensor dtype casts.  The concatenated weight is
    cached on the module, keyed by the three weights' storage, version counter and the compute dtype, so an
    optimizer step, a `.to()` or an in-place LoRA merge rebuilds it.  (An edit THROUGH `.data` -- `w.data.add_(d)` --
    bumps no version counter: call `invalidate_fused_qkv(model)` after one.)  Returns None when the three
    projections are not plain bias-compatible `nn.Linear` layers over the same input, or when autograd is recording
    and a weight requires grad (then the caller issues them one by one).
    Returns the [..., 3D] projection output; q, k, v are its three D-wide column slabs."""
    lq, lk, lv = attn.to_q, attn.to_k, attn.to_v
    Linear = torch.nn.Linear
    if not (type(lq) is Linear and type(lk) is Linear and type(lv) is Linear and x.is_cuda):
        return None
    wq, wk, wv = lq.weight, lk.weight, lv.weight
    if not (wq.shape == wk.shape == wv.shape and wq.dtype == wk.dtype == wv.dtype and wq.shape[1] == x.shape[-1]):
        return None
    biases = (lq.bias, lk.bias, lv.bias)
    if any(b is None for b in biases) != all(b is None for b in biases):
        return None
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (wq, wk, wv) + biases):
        return None      # the cached concatenation is built without autograd: training keeps the three Linear calls
    if torch.is_autocast_enabled("cuda"):
        cdt = torch.get_autocast_dtype("cuda")
    else:
        cdt = x.dtype if x.dtype == wq.dtype else None
    if cdt is None:
        return None
    key = (cdt, wq.data_ptr(), wk.data_ptr(), wv.data_ptr(), wq._version, wk._version, wv._version,
           None if biases[0] is None else tuple((b.data_ptr(), b._version) for b in biases))
    cached = attn.__dict__.get("_tf_qkv_cache")
    if cached is None or cached[0] != key:
        with torch.no_grad():
            wcat = torch.cat([wq, wk, wv], dim=0).to(cdt).contiguous()
            bcat = None if biases[0] is None else torch.cat(list(biases), dim=0).to(cdt).contiguous()
        cached = (key, wcat, bcat)
        attn.__dict__["_tf_qkv_cache"] = cached
    _, wcat, bcat = cached
    with torch.autocast("cuda", enabled=False):
        qkv = torch.nn.functional.linear(x.to(cdt), wcat, bcat)
    return qkv


FUSE_QKV = os.environ.get("TOKENFLOW_FUSED_QKV", "1") not in ("", "0")


def invalidate_fused_qkv(model: torch.nn.Module) -> None:
    """Drop every cached [Wq;Wk;Wv] of `model` (needed only after weight edits that bypass the version counters)."""
    for m in model.modules():
        m.__dict__.pop("_tf_qkv_cache", None)


def _make_sa_forward(self, pnp: bool):
    to_out = self.to_out
    if type(to_out) is torch.nn.modules.container.ModuleList:
        to_out = self.to_out[0]          # dropout to_out[1] skipped, as 108-112

    def forward(x, encoder_hidden_states=None, attention_mask=None):
        is_cross = encoder_hidden_states is not None
        qkv = _fused_qkv(self, x) if (FUSE_QKV and not is_cross) else None
        if qkv is not None:
            proj_dtype = qkv.dtype
            cdt = ops.compute_dtype(qkv)
            if proj_dtype != cdt:        # fp32 model without autocast: ONE cast of the fused output
                qkv = qkv.to(cdt)
            D = qkv.shape[-1] // 3
            q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]     # column slabs, row stride 3D
        else:
            encoder_hidden_states = encoder_hidden_states if is_cross else x
            q, k, v = self.to_q(x), self.to_k(encoder_hidden_states), self.to_v(encoder_hidden_states)
            proj_dtype = q.dtype
            cdt = ops.compute_dtype(q)
            if proj_dtype != cdt:
                q, k, v = q.to(cdt), k.to(cdt), v.to(cdt)
        inject = pnp and _injecting(self)
        shard = None if is_cross else _active_shard(self)
        if shard is not None:     # q, k, v are this rank's keyframes; the bank is everybody's (register_frame_shard)
            out = shard.pivotal_attention(q, k, v, self.heads, self.scale, inject)
        else:
            out = ops.ext_attn(q, k, v, self.heads, self.scale, inject)
        return to_out(out if out.dtype == proj_dtype else out.to(proj_dtype))

    return forward


def register_extended_attention_pnp(model, injection_schedule):
    """tokenflow_utils.py:106-214: every BasicTransformerBlock.attn1 gets the extended
    attention with an empty schedule (203-206); the 8 decoder blocks
    up_blocks[1].attentions[1,2], up_blocks[2,3].attentions[0..2] get the real one (208-214)."""
    _announce()
    for _, module in model.unet.named_modules():
        if isinstance_str(module, "BasicTransformerBlock"):
            module.attn1.forward = _make_sa_forward(module.attn1, pnp=True)
            _set_schedule(module.attn1, [])
    for res, blocks in _INJECTED_UP.items():
        for block in blocks:
            module = model.unet.up_blocks[res].attentions[block].transformer_blocks[0].attn1
            module.forward = _make_sa_forward(module, pnp=True)
            _set_schedule(module, injection_schedule)


def register_extended_attention(model):
    """tokenflow_utils.py:216-294 (SDEdit driver): extended attention, never injects."""
    _announce()
    for _, module in model.unet.named_modules():
        if isinstance_str(module, "BasicTransformerBlock"):
            module.attn1.forward = _make_sa_forward(module.attn1, pnp=False)
    for res, blocks in _INJECTED_UP.items():
        for block in blocks:
            module = model.unet.up_blocks[res].attentions[block].transformer_blocks[0].attn1
            module.forward = _make_sa_forward(module, pnp=False)


# --------------------------------------------------------------------------- TokenFlow block
_weights_cache = {}


def _blend_weights(n: int, device) -> torch.Tensor:
    """w1 of tokenflow_utils.py:375-383.  s - p1 = j - n//2 and s - p2 = j + n - n//2 do not
    depend on the chunk index, so the n-vector is computed once per (n, device) with the
    reference's own torch ops (on the host, so the value is the CPU reference's)."""
    key = (n, str(device))
    w = _weights_cache.get(key)
    if w is None:
        s = torch.arange(0, n)
        d1 = torch.abs(s - n // 2)
        d2 = torch.abs(s + n - n // 2)
        w = torch.sigmoid(d2 / (d1 + d2)).to(device)
        _weights_cache[key] = w
    return w


def _fused_norm_dtype(mod: torch.nn.Module, x: torch.Tensor, in_dtype=None):
    """The 16-bit dtype in which a plain LayerNorm of the block can be produced directly by
    `ops.layer_norm` (one pass instead of torch's autocast sequence cast-up / fp32 norm / cast-down in front
    of the next Linear), or None: keep the module call (fp32 models, AdaLayerNorm, CPU tensors).
    in_dtype: dtype of the norm's input when it is not x's (the residual-add form normalises `a + x`)."""
    D = x.shape[-1]
    in_dtype = x.dtype if in_dtype is None else in_dtype
    if (type(mod) is not torch.nn.LayerNorm or not x.is_cuda or tuple(mod.normalized_shape) != (D,)
            or D % 8 or D > 2048 or in_dtype not in (torch.float32, torch.bfloat16, torch.float16)):
        return None
    if torch.is_autocast_enabled("cuda"):
        dt = torch.get_autocast_dtype("cuda")
        return dt if dt in (torch.bfloat16, torch.float16) else None
    if in_dtype != torch.float32 and (mod.weight is None or mod.weight.dtype == in_dtype):
        return in_dtype
    return None


# Which LayerNorms of a block go through `ops.layer_norm`:  "all" (default) = norm1 -- the producer of the attention
# input, the NN-search rows and the pivots' inverse norms (row f2 of SURVEY.md section 8) -- and also norm2 / norm3,
# whose only consumers are Linear layers (attn2.to_q, the feed-forward): under autocast torch evaluates layer_norm
# in fp32 and the Linear rounds its input to 16 bit, so the kernel's single rounding of the fp32 result IS the value
# the reference's Linear sees (pinned by tests/test_fullsize_gpu.py::test_fused_layer_norm_is_the_autocast_value);
# "norm1" = only norm1; "none" = always the modules.
FUSE_NORMS = os.environ.get("TOKENFLOW_FUSED_NORMS", "all")
# TOKENFLOW_FUSED_GATHER_NORM=0: keep the propagation's gather and the norm behind it as two launches
FUSE_GATHER_NORM = os.environ.get("TOKENFLOW_FUSED_GATHER_NORM", "1") not in ("", "0")
# TOKENFLOW_NORM1_ALL_BRANCHES=1: a propagation pass normalises all three branches as the reference does (323) although
# only the source branch is read (335-343); default: the source branch alone (A/B: profiles/r05_hooks_bench.txt)
NORM1_ALL_BRANCHES = os.environ.get("TOKENFLOW_NORM1_ALL_BRANCHES", "0") not in ("", "0")


def _block_norm(mod: torch.nn.Module, x: torch.Tensor, want_inv_norm: bool = False, which: str = "norm1", dest=None):
    """(LayerNorm(x), 1/||row|| or None) through the fused kernel when it applies, else the module itself.
    dest(dtype) -> (out, inv_out): destination tensors for the fused kernel (the sharded pivotal pass hands out views
    of the block's halo-extended state, so that the norm writes the pivots and their inverse norms in place)."""
    dt = _fused_norm_dtype(mod, x) if (FUSE_NORMS == "all" or FUSE_NORMS == which) else None
    if dt is None:
        return mod(x), None
    out, inv_out = dest(dt) if dest is not None else (None, None)
    return ops.layer_norm(x, mod.weight, mod.bias, mod.eps, dt, want_inv_norm, out=out, inv_out=inv_out)


def _shard_state(block, shard, S: int, D: int, dtype, device):
    """The block's halo-extended propagation state on a sharded rank, allocated ONCE per block and shape and reused
    step after step (a step's chunk passes are done with it before the next pivotal pass writes it):
      norm [1 + 3*Kl, S, D]   slot 0 = the left neighbour's last keyframe, then norm1's output of the pivotal pass,
                              branch-major -- so the first Kl + 1 slots ARE the halo-extended pivots, written by the
                              norm itself;
      inv  [1 + 3*Kl, S]      the same for the rows' inverse norms;
      kfo  [3, Kl + 1, S, D]  cached attention output (after to_out) with the neighbour's slot in front."""
    key = (shard.Kl, S, D, dtype, device)
    st = block.__dict__.get("_tf_shard_state")
    if st is None or st[0] != key:
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            # the state outlives any one graph (the pivotal pass writes it, every chunk pass of the step reads it, eager
            # passes reuse it): it must not come from a capture's private pool
            raise RuntimeError("register_frame_shard: the per-block propagation state would be allocated inside a HIP-graph "
                               "capture; run one eager pivotal pass first (GraphCache(warmup >= 1) does)")
        Kl = shard.Kl
        st = (key, torch.empty(1 + 3 * Kl, S, D, dtype=dtype, device=device),
              torch.empty(1 + 3 * Kl, S, dtype=torch.float32, device=device),
              torch.empty(3, Kl + 1, S, D, dtype=dtype, device=device))
        block.__dict__["_tf_shard_state"] = st
    return st[1], st[2], st[3]


def _add_norm(mod: torch.nn.Module, a: torch.Tensor, h: torch.Tensor, which: str):
    """(a + h, LayerNorm(a + h)): the residual add of the block and the norm that follows it
    (`hidden_states = attn_output + hidden_states` then norm2 / norm3, tokenflow_utils.py:396-403, 409-414) in one
    pass through `ops.add_layer_norm` when the norm is fusable, else the two torch ops.  Bit-identical either way."""
    if a.shape == h.shape and a.is_cuda and h.is_cuda and (FUSE_NORMS == "all" or FUSE_NORMS == which):
        dt = _fused_norm_dtype(mod, h, torch.promote_types(a.dtype, h.dtype))
        if dt is not None:
            return ops.add_layer_norm(a, h, mod.weight, mod.bias, mod.eps, dt)
    total = a + h
    return total, _block_norm(mod, total, which=which)[0]


def _chunk_run(batch_idx):
    """(first chunk, number of chunks) of a `batch_idx` state.  The reference sets an int (one chunk of
    n = batch_size frames per UNet pass, run_tokenflow_pnp.py:228-231).  Extension for large-memory GPUs: a
    `range` / list of CONSECUTIVE chunk indices means the pass carries all those chunks, frames chunk-major
    inside every branch -- one UNet pass over the whole video instead of C passes."""
    if isinstance(batch_idx, (range, list, tuple)):
        ids = [int(i) for i in batch_idx]
        if not ids or any(b - a != 1 for a, b in zip(ids, ids[1:])):
            raise ValueError(f"batch_idx {batch_idx!r}: need a non-empty run of consecutive chunk indices")
        return ids[0], len(ids)
    return int(batch_idx), 1


def make_tokenflow_attention_block(block_class: Type[torch.nn.Module]) -> Type[torch.nn.Module]:
    """tokenflow_utils.py:296-429."""

    class TokenFlowBlock(block_class):

        @property
        def pivot_hidden_states(self):
            """`norm1` output of the last pivotal pass (tokenflow_utils.py:327), in the dtype the reference caches."""
            st = self.__dict__.get("_tf_pivot_hidden")
            if st is None:
                raise AttributeError("pivot_hidden_states: no pivotal pass has run")
            t, dt = st
            return t if t.dtype == dt else t.to(dt)

        @pivot_hidden_states.setter
        def pivot_hidden_states(self, value):
            self.__dict__["_tf_pivot_hidden"] = (value, value.dtype)

        @property
        def attn_output(self):
            """What the reference leaves in `self.attn_output`: the attention output after a pivotal pass (355-360), the
            SELECTED keyframe outputs `kf_attn_output.view(3, K, S, D)[:, batch_idxs]` after a propagation pass (361-363;
            gated under AdaLayerNormZero, 364-365).  Nothing on the path reads the second form -- the gather indexes the
            cached output in place -- so it is built only if somebody asks."""
            st = self.__dict__.get("_tf_attn_output")
            if st is None:
                raise AttributeError("attn_output: no pass has run")
            if callable(st):
                st = st()
                self.__dict__["_tf_attn_output"] = st
            return st

        @attn_output.setter
        def attn_output(self, value):
            self.__dict__["_tf_attn_output"] = value

        def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None,
                    encoder_attention_mask=None, timestep=None, cross_attention_kwargs=None,
                    class_labels=None) -> torch.Tensor:
            batch_size, sequence_length, dim = hidden_states.shape
            n_frames = batch_size // 3
            hidden_states = hidden_states.view(3, n_frames, sequence_length, dim)

            norm_inv = None
            gate_msa = None
            pending = None          # a residual branch output not yet added to hidden_states (see _add_norm)
            prenorm = None          # (which, tensor): the next norm, already produced by the propagation's gather
            if self.use_ada_layer_norm:
                norm_hidden_states = self.norm1(hidden_states, timestep)
            elif self.use_ada_layer_norm_zero:
                norm_hidden_states, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.norm1(
                    hidden_states, timestep, class_labels, hidden_dtype=hidden_states.dtype)
            else:   # row f2: LayerNorm emits the 16-bit rows the kernels read (+ the pivots' inverse norms)
                dest = None
                shard0 = _active_shard(self) if self.pivotal_pass else None
                if shard0 is not None and n_frames == shard0.Kl and hidden_states.is_cuda:
                    # a neighbour halo of the PREVIOUS pivotal pass may still be in flight if no chunk pass (and no
                    # join_frame_shard) consumed it: its send reads, and its receive writes, the state this pass is
                    # about to overwrite -- order behind it first
                    prev = self.__dict__.get("_tf_halo")
                    if prev is not None and prev[3]:
                        shard0.halo_wait(prev[3])
                        self.__dict__["_tf_halo"] = (prev[0], prev[1], prev[2], [])
                    # sharded pivotal pass: norm1 writes straight into the block's halo-extended state
                    def dest(dt, _s=shard0):
                        nb, ib, _ = _shard_state(self, _s, sequence_length, dim, dt, hidden_states.device)
                        return nb[1:].view(3, n_frames, sequence_length, dim), ib[1:].view(3, n_frames, sequence_length)
                if self.pivotal_pass or NORM1_ALL_BRANCHES:
                    norm_hidden_states, norm_inv = _block_norm(self.norm1, hidden_states, bool(self.pivotal_pass), dest=dest)
                else:
                    # a propagation pass reads norm1's output of the SOURCE branch only (the NN-search targets,
                    # 335-343; the reference normalises all three branches at 323 and drops two): one third of the
                    # launch's bytes.  Row-wise op: the rows that are computed are bit-identical.
                    norm_hidden_states, norm_inv = _block_norm(self.norm1, hidden_states[:1])[0], None
            norm_hidden_states = norm_hidden_states.view(-1, n_frames, sequence_length, dim)

            cross_attention_kwargs = cross_attention_kwargs if cross_attention_kwargs is not None else {}
            if self.pivotal_pass:
                # 326-327 + 352-360: cache the normalised features and the attention output
                # the reference caches what norm1 returns (327): fp32 under autocast, where torch evaluates
                # layer_norm in fp32.  The fused norm emitted 16-bit rows; the attribute keeps the reference's
                # dtype and is widened only if somebody reads it (nothing on the path does: the NN search reads
                # _tf_pivots / _tf_pivot_inv_norm)
                self._tf_pivot_hidden = (norm_hidden_states,
                                         torch.float32 if (norm_inv is not None and torch.is_autocast_enabled("cuda"))
                                         else norm_hidden_states.dtype)
                src = norm_hidden_states[0]
                self._tf_pivots = src.to(ops.compute_dtype(src)).contiguous()       # [K,S,D] 16-bit
                self._tf_pivot_inv_norm = (norm_inv.view(3, n_frames, sequence_length)[0] if norm_inv is not None
                                           else ops.pivot_inv_norm(self._tf_pivots))   # [K,S] fp32
                shard = _active_shard(self)
                inplace = False
                if shard is not None:
                    if n_frames != shard.Kl:
                        raise ValueError(f"pivotal pass with {n_frames} keyframes per branch on a rank that owns "
                                         f"{shard.Kl} (register_frame_shard)")
                    st = self.__dict__.get("_tf_shard_state")
                    # in place: norm1 has written the pivots and inverse norms into slots 1.. of the halo-extended
                    # state (the fused LayerNorm path); else (AdaLayerNorm, fp32 models) the two-exchange form, whose
                    # pivots' half does not depend on the attention and travels under it
                    inplace = (norm_inv is not None and st is not None
                               and norm_hidden_states.data_ptr() == st[1][1:].data_ptr())
                    if not inplace:
                        halo = shard.halo_start(self._tf_pivots, self._tf_pivot_inv_norm.contiguous())
                self.attn_output = self.attn1(
                    norm_hidden_states.view(batch_size, sequence_length, dim),
                    encoder_hidden_states=encoder_hidden_states if self.only_cross_attention else None,
                    **cross_attention_kwargs)
                self.kf_attn_output = self.attn_output
                if shard is not None and inplace:
                    # ONE grouped exchange per block on the cached state: pivots / inverse norms were written by norm1,
                    # the to_out-projected attention output takes one copy into its slots (to_out is the model's own
                    # Linear: its output tensor is torch's); no allocation, no other copy
                    nb, ib, kfo = st[1], st[2], st[3]
                    Kl = shard.Kl
                    kfo[:, 1:].copy_(self.kf_attn_output.reshape(3, Kl, sequence_length, dim))
                    piv_e, inv_e = nb[:Kl + 1], ib[:Kl + 1]
                    reqs = shard.halo_block(piv_e, inv_e, kfo)
                    self.__dict__["_tf_halo"] = (piv_e, inv_e, kfo.view(3 * (Kl + 1), sequence_length, dim), reqs)
                elif shard is not None:   # (pivots, inverse norms, attention output) with the neighbour's slot in
                    # front, and the pending requests of the exchange: the first chunk pass waits for them
                    self.__dict__["_tf_halo"] = shard.halo_finish(
                        halo, self.kf_attn_output.reshape(batch_size, sequence_length, dim).contiguous(), wait=False)
                if self.use_ada_layer_norm_zero:
                    self.attn_output = gate_msa.unsqueeze(1) * self.attn_output
                attn_output = self.attn_output
                hidden_states = hidden_states.reshape(batch_size, sequence_length, dim)
                pending = attn_output              # `hidden_states = attn_output + hidden_states` (396-397): fused into
            else:                                  # the norm that follows it, below
                c0, n_chunks = _chunk_run(self.batch_idx)
                if n_frames % n_chunks:
                    raise ValueError(f"{n_frames} frames per branch do not split into {n_chunks} chunks")
                n = n_frames // n_chunks
                kf = self.kf_attn_output
                piv, inv = self._tf_pivots, self._tf_pivot_inv_norm
                s0 = c0                 # slot of keyframe c0 in kf / piv / inv (one process: the keyframe index itself)
                shard = _active_shard(self)
                if shard is not None:   # this rank's caches: slot 0 = the left neighbour's last keyframe, then its own
                    if not (shard.kf0 <= c0 and c0 + n_chunks <= shard.kf0 + shard.Kl):
                        raise ValueError(f"chunks {c0}..{c0 + n_chunks - 1} are not owned by this rank "
                                         f"({shard.kf0}..{shard.kf0 + shard.Kl - 1})")
                    piv, inv, kf, reqs = self.__dict__["_tf_halo"]
                    if reqs:
                        shard.halo_wait(reqs)
                        self.__dict__["_tf_halo"] = (piv, inv, kf, [])
                    s0 = c0 - shard.kf0 + 1
                K = kf.shape[0] // 3
                if self.use_ada_layer_norm_zero:
                    # 362-366: the reference gates the SELECTED keyframe outputs before the gather:
                    # `attn_output = gate_msa.unsqueeze(1) * kf_attn_output.view(3,K,S,D)[:, batch_idxs]`.  Same
                    # torch expression here (same broadcasting, same promotion) on the keyframes this pass reads;
                    # the gated copy becomes the gather source, re-indexed from 0.
                    lo = s0 - 1 if c0 > 0 else s0
                    sel = kf.view(3, K, sequence_length, dim)[:, lo:s0 + n_chunks]
                    kf = (gate_msa.unsqueeze(1) * sel).reshape(-1, sequence_length, dim)
                    # the attribute in the reference's order [i, i-1] (the gather source keeps ascending keyframes)
                    self.attn_output = lambda g_=kf: g_.view(3, -1, sequence_length, dim).flip(1)
                    kf_base, K = lo, kf.shape[0] // 3
                else:
                    kf_base = 0
                    # 361-363: the reference leaves the selected keyframe outputs, order [i, i-1] (a run of chunks: its
                    # keyframes in descending order); lazily -- the gather below reads the cache in place
                    slots = list(range(s0 + n_chunks - 1, (s0 - 1 if c0 > 0 else s0) - 1, -1))
                    self.attn_output = (lambda kf_=kf, K_=K, sl=slots:
                                        kf_.view(3, K_, sequence_length, dim)[:, sl])
                # 329-348: nearest neighbours of the SOURCE branch among keyframe c (and c-1), per chunk
                tgt = norm_hidden_states[0].reshape(n_frames * sequence_length, dim).to(self._tf_pivots.dtype)
                # 361-397: gather (same indices for the 3 branches), blend, residual -- fused with the search.
                # dtype follows torch promotion in the reference: the blend is fp32 (w1 is fp32, 385-388),
                # chunk 0 keeps the cached dtype (390); then `attn_output + hidden_states` (397).
                two = c0 + n_chunks - 1 > 0                      # some chunk blends two keyframes
                blend_dtype = torch.float32 if two else kf.dtype
                out_dtype = torch.promote_types(blend_dtype, hidden_states.dtype)
                w = _blend_weights(n, kf.device) if two else None
                resid = hidden_states.reshape(batch_size, sequence_length, dim)
                if kf_base:      # gated copy holds keyframes kf_base.. only: search the same window of the pivots
                    piv, inv = piv[kf_base:kf_base + K], inv[kf_base:kf_base + K]
                # the norm that consumes this pass's residual stream next (norm2 in front of the cross-attention, else
                # norm3 in front of the feed-forward) rides in the gather's epilogue when it is a plain LayerNorm: the
                # fp32 stream is written once and not re-read by a norm launch (row f1/f2 of SURVEY.md section 8)
                nxt = None
                if self.attn2 is not None:
                    nxt = None if self.use_ada_layer_norm else ("norm2", self.norm2)
                elif not self.use_ada_layer_norm_zero:
                    nxt = ("norm3", self.norm3)
                fuse = None
                if nxt is not None and (FUSE_NORMS == "all" or FUSE_NORMS == nxt[0]) and FUSE_GATHER_NORM:
                    ndt = _fused_norm_dtype(nxt[1], resid, out_dtype)
                    if ndt is not None and ops.norm_fusable(kf, resid, out_dtype, 2 if two else 1, ndt):
                        fuse = (nxt[1].weight, nxt[1].bias, nxt[1].eps, ndt)
                if n_chunks == 1:
                    ids = [s0 - kf_base] if c0 == 0 else [s0 - kf_base, s0 - 1 - kf_base]
                    res = ops.propagate(tgt, piv, inv, ids, kf, w, n, resid, out_dtype, norm=fuse)
                else:
                    res = ops.propagate_chunks(tgt, piv, inv, kf, w, n, n_chunks, s0 - kf_base, c0 == 0, resid,
                                               out_dtype, norm=fuse)
                if fuse is not None:
                    hidden_states, prenorm = res[0], (nxt[0], res[1])
                else:
                    hidden_states = res

            if self.attn2 is not None:
                if self.use_ada_layer_norm:
                    if pending is not None:
                        hidden_states, pending = pending + hidden_states, None
                    norm_hidden_states = self.norm2(hidden_states, timestep)
                elif pending is not None:
                    hidden_states, norm_hidden_states = _add_norm(self.norm2, pending, hidden_states, "norm2")
                    pending = None
                elif prenorm is not None and prenorm[0] == "norm2":
                    norm_hidden_states = prenorm[1]
                else:
                    norm_hidden_states = _block_norm(self.norm2, hidden_states, which="norm2")[0]
                pending = self.attn2(norm_hidden_states, encoder_hidden_states=encoder_hidden_states,
                                     attention_mask=encoder_attention_mask, **cross_attention_kwargs)   # + hidden_states (411)

            if self.use_ada_layer_norm_zero:    # the modulation consumes the fp32 norm output: keep the module
                if pending is not None:
                    hidden_states, pending = pending + hidden_states, None
                norm_hidden_states = self.norm3(hidden_states)
                norm_hidden_states = norm_hidden_states * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
            elif pending is not None:
                hidden_states, norm_hidden_states = _add_norm(self.norm3, pending, hidden_states, "norm3")
                pending = None
            elif prenorm is not None and prenorm[0] == "norm3":
                norm_hidden_states = prenorm[1]
            else:
                norm_hidden_states = _block_norm(self.norm3, hidden_states, which="norm3")[0]
            ff_output = self.ff(norm_hidden_states)
            if self.use_ada_layer_norm_zero:
                ff_output = gate_mlp.unsqueeze(1) * ff_output
            return ff_output + hidden_states

    return TokenFlowBlock


def set_tokenflow(model: torch.nn.Module):
    """tokenflow_utils.py:432-448: class-swap every BasicTransformerBlock in place."""
    _announce()
    for _, module in model.named_modules():
        if isinstance_str(module, "BasicTransformerBlock"):
            module.__class__ = make_tokenflow_attention_block(module.__class__)
            if not hasattr(module, "use_ada_layer_norm_zero"):     # older diffusers (444-446)
                module.use_ada_layer_norm = False
                module.use_ada_layer_norm_zero = False
    return model
