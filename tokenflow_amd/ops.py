"""Torch-tensor wrappers over the C ABI.  PyTorch supplies device memory and the
current HIP stream; all arithmetic happens in the HIP kernels.  No fallbacks."""
import ctypes
import os
from typing import Optional, Sequence

import torch

from . import _lib

_DT = {torch.bfloat16: _lib.TF_BF16, torch.float16: _lib.TF_F16, torch.float32: _lib.TF_F32}


def _need_gpu(*ts):
    """All tensors on ONE GPU (no CPU fallback, no cross-device launches)."""
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.TokenflowHipError(
                "tokenflow_amd ops run on MI355X only: got a CPU tensor (there is no CPU fallback)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise _lib.TokenflowHipError(f"tokenflow_amd ops: tensors on different devices ({dev} and {t.device})")
    return dev


def _launch(dev, what: str, fn, *args, stream: Optional[int] = None):
    """Call a C-ABI entry point whose last argument is the stream: the launch goes to the CURRENT stream OF THE
    TENSORS' DEVICE (or to `stream`, a raw HIP stream of that device, when the caller orders its own streams), with
    that device made current for the duration of the call when it is not already (a kernel enqueued on another
    device's stream with foreign pointers faults or corrupts memory)."""
    if dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev):
            rc = fn(*args, torch.cuda.current_stream(dev).cuda_stream if stream is None else stream)
    else:
        rc = fn(*args, torch.cuda.current_stream(dev).cuda_stream if stream is None else stream)
    if rc != 0:
        _lib.check(rc, what)


# 16-bit type fp32 tensors are rounded to at the op boundary (an fp32 model run WITHOUT autocast; under autocast the
# projections already arrive in the autocast dtype).  bf16 (default) cannot overflow; TOKENFLOW_FP32_AS=f16 keeps 11
# significand bits instead of 8 -- the reference's own GPU dtype (run_tokenflow_pnp.py:47, 220) -- for models whose
# activations stay inside f16's range.
def _fp32_as_from_env() -> torch.dtype:
    name = os.environ.get("TOKENFLOW_FP32_AS", "bf16").strip().lower()
    table = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f16": torch.float16, "fp16": torch.float16,
             "float16": torch.float16, "half": torch.float16}
    if name not in table:
        raise ValueError(f"TOKENFLOW_FP32_AS={name!r}: must be 'bf16' or 'f16' (aliases: bfloat16, fp16, float16, half)")
    return table[name]


# f16 has no overflow guard: an fp32 activation above 65504 becomes inf in q / k / v / the pivots
FP32_AS = _fp32_as_from_env()


def compute_dtype(t: torch.Tensor) -> torch.dtype:
    """16-bit MFMA input type used for a tensor of dtype t.dtype (fp32 inputs are rounded to `FP32_AS`)."""
    return t.dtype if t.dtype in (torch.bfloat16, torch.float16) else FP32_AS


_ws_cache = {}
_attn_ws_bytes = {}   # (K, S, H, Dh, dtype) -> scratch bytes of tf_ext_attn_fwd (a pure function of the shape)


def _workspace(nbytes: int, device, tag: str = "attn", stream: Optional[int] = None) -> torch.Tensor:
    """Scratch for one launch, cached per (purpose, device, stream).  While a HIP graph is being captured the
    buffer comes from the graph's private pool and must live and die with that graph: never cached."""
    if stream is None:
        if device.index != torch.cuda.current_device():      # capture state is a property of the TENSORS' device's stream
            with torch.cuda.device(device):
                capturing = torch.cuda.is_current_stream_capturing()
        else:
            capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        stream = torch.cuda.current_stream(device).cuda_stream
    key = (tag, device.index, stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def drop_workspaces(stream=None) -> None:
    """Forget the cached scratch of `stream` (a torch.cuda.Stream; None = every stream): the memory goes back to
    torch's allocator once the launches that used it have run."""
    if stream is None:
        _ws_cache.clear()
        return
    for key in [k for k in _ws_cache if k[2] == stream.cuda_stream]:
        del _ws_cache[key]


# TOKENFLOW_FOLD_SCALE=1: at head dim 40 fold the softmax scale into q (rounded to the input dtype): several % faster,
# 3-12x outside the parity bound on peaked logits (profiles/r02_fold_accuracy.txt).  Default: fp32 scaling of the scores.
FOLD_SCALE = os.environ.get("TOKENFLOW_FOLD_SCALE", "0") not in ("", "0")
# TOKENFLOW_ATTN_NO_SPLIT=1: never split a bank problem over workgroups.  By default small grids (a sharded rank, the
# 16x16 level) split the bank into runs of frames and merge; the merge re-associates fp32 sums, so results then
# depend on the grid size in the last bits.  With the flag the arithmetic of a (query, head) is the same everywhere.
NO_SPLIT = os.environ.get("TOKENFLOW_ATTN_NO_SPLIT", "0") not in ("", "0")


def ext_attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float,
             inject: bool, out: Optional[torch.Tensor] = None, q_frame0: int = 0,
             fold_scale: Optional[bool] = None, part: str = "all",
             out_dtype: Optional[torch.dtype] = None, no_split: Optional[bool] = None,
             fused: Optional[bool] = None, hints: int = 0) -> torch.Tensor:
    """Extended attention core (tokenflow_utils.py:124-197).  k,v: [3K,S,D] bf16/f16 (the bank),
    q: [3Kq,S,D] = the queries of keyframes q_frame0..q_frame0+Kq-1 (Kq = K on one GPU); last dim
    contiguous, equal token stride (q, k, v may be column slabs of one fused projection output).
    Returns [3Kq,S,D] in the same dtype, or in fp32 with out_dtype=torch.float32 (the normalised fp32
    accumulator, no 16-bit output rounding).
    part = "bank": only the uncond/cond branches are computed (the source slabs of v and out, and those
    of q, k that the call does not read, are never touched); part = "source": only the source branch.
    no_split: True = one pass per bank problem whatever the grid (TF_ATTN_NO_SPLIT: arithmetic independent of the
    grid size), False = let small grids split the bank over workgroups and merge; None = the module default.
    fused: None = the library decides (small problems run in ONE fused launch, csrc/ext_attn_fused.hip), False = the
    streaming kernels at every size (TF_ATTN_NO_FUSED), True = the fused kernel at any size it is built for;
    hints: further TF_ATTN_* bits (_lib.attn_hint, TF_ATTN_PRECISE_P ...; measurements and tests)."""
    dev = _need_gpu(q, k, v, out)
    lib = _lib.load()
    B, S, D = k.shape
    Bq = q.shape[0]
    if B % 3 or Bq % 3 or D % heads or q.shape[1:] != k.shape[1:] or v.shape != k.shape:
        raise ValueError(f"ext_attn: bad shapes q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)} heads {heads}")
    K, Kq, dh = B // 3, Bq // 3, D // heads
    dt = _DT.get(q.dtype)
    if dt is None or dt == _lib.TF_F32 or k.dtype != q.dtype or v.dtype != q.dtype:
        raise TypeError(f"ext_attn: q/k/v must share dtype bf16 or f16, got {q.dtype},{k.dtype},{v.dtype}")

    def rows(t):
        if t.stride(-1) != 1 or t.stride(0) != S * t.stride(1):
            t = t.contiguous()
        return t
    q, k, v = rows(q), rows(k), rows(v)
    ld = q.stride(1)
    if k.stride(1) != ld or v.stride(1) != ld:
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        ld = D
    if out_dtype is None:
        out_dtype = out.dtype if out is not None else q.dtype
    if out_dtype not in (q.dtype, torch.float32):
        raise TypeError(f"ext_attn: out_dtype {out_dtype} (the input dtype or float32)")
    if out is None:
        out = torch.empty(Bq, S, D, dtype=out_dtype, device=q.device)
    elif out.dtype != out_dtype or not out.is_contiguous() or out.shape != (Bq, S, D):
        raise ValueError("ext_attn: `out` must be a contiguous [3Kq,S,D] tensor of out_dtype")
    flags = (1 if inject else 0) | (_lib.TF_ATTN_FOLD_SCALE if (FOLD_SCALE if fold_scale is None else fold_scale) else 0)
    if out_dtype == torch.float32:
        flags |= _lib.TF_ATTN_OUT_F32
    flags |= {"all": 0, "bank": _lib.TF_ATTN_BANK_ONLY, "source": _lib.TF_ATTN_SOURCE_ONLY}[part]
    if NO_SPLIT if no_split is None else no_split:
        flags |= _lib.TF_ATTN_NO_SPLIT
    flags |= int(hints) | (0 if fused is None else _lib.TF_ATTN_FUSED if fused else _lib.TF_ATTN_NO_FUSED)
    key = (K, S, heads, dh, dt)
    nbytes = _attn_ws_bytes.get(key)
    if nbytes is None:
        nbytes = _attn_ws_bytes[key] = lib.tf_ext_attn_workspace_bytes(K, S, heads, dh, dt)
    ws = _workspace(nbytes, q.device)
    _launch(dev, "tf_ext_attn_fwd", lib.tf_ext_attn_fwd, q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(),
            K, Kq, int(q_frame0), S, heads, dh, ld, float(scale), flags, dt, ws.data_ptr(), ws.numel())
    return out


def _view_base(t: torch.Tensor, b0: int, S: int, what: str):
    """(base pointer such that branch b lives at base + b*branch_stride, branch stride, frame stride, token stride)
    of a 4-D view [branches b0.., frames, S, D]."""
    if t.dim() != 4 or t.shape[2] != S or t.stride(3) != 1:
        raise ValueError(f"ext_attn_views: {what} must be a [branches, frames, S, D] view with a contiguous last dim")
    bs = t.stride(0) if t.shape[0] > 1 else 0
    return t.data_ptr() - b0 * bs * t.element_size(), bs, t.stride(1), t.stride(2)


def ext_attn_views(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, heads: int, scale: float,
                   inject: bool, part: str = "all", branch0=(0, 0, 0, 0), q_frame0: int = 0,
                   fold_scale: Optional[bool] = None, no_split: Optional[bool] = None,
                   stream: Optional[int] = None, fused: Optional[bool] = None, hints: int = 0) -> torch.Tensor:
    """`ext_attn` on strided 4-D views [branches, frames, S, D] (tf_ext_attn_fwd_strided): q, k, v are read where
    a collective left them and `out` is written where the next one sends from -- no re-layout copies.  Each view
    holds the branches `branch0[i] ..` of its tensor (q, k, v, out in that order; e.g. a bank-only call passes the
    uncond/cond slabs with branch0 = 1, and under injection the single source slab of q and k with branch0 = 0).
    Branch and frame strides are free; k and v share one token stride, q has its own, out is dense (= D).
    stream: a raw HIP stream of the tensors' device to launch on instead of torch's current one (the caller orders it
    against the others itself: sharded.py runs the source branch beside the bank exchange)."""
    dev = _need_gpu(q, k, v, out)
    lib = _lib.load()
    S, D = k.shape[2], k.shape[3]
    K, Kq, dh = k.shape[1], q.shape[1], D // heads
    dt = _DT.get(q.dtype)
    if dt is None or dt == _lib.TF_F32 or k.dtype != q.dtype or v.dtype != q.dtype or D % heads:
        raise TypeError("ext_attn_views: q/k/v must share dtype bf16 or f16")
    qp, q_bs, q_fs, ld_q = _view_base(q, branch0[0], S, "q")
    kp, k_bs, k_fs, ld = _view_base(k, branch0[1], S, "k")
    vp, v_bs, v_fs, ld_v = _view_base(v, branch0[2], S, "v")
    op, o_bs, o_fs, ld_o = _view_base(out, branch0[3], S, "out")
    if ld_v != ld or ld_o != D or out.shape[1] != Kq or v.shape[1] != K:
        raise ValueError("ext_attn_views: k and v need one token stride, out a dense one; frames of v = frames of k")
    if out.dtype not in (q.dtype, torch.float32):
        raise TypeError("ext_attn_views: out dtype")
    flags = (1 if inject else 0) | (_lib.TF_ATTN_FOLD_SCALE if (FOLD_SCALE if fold_scale is None else fold_scale) else 0)
    flags |= {"all": 0, "bank": _lib.TF_ATTN_BANK_ONLY, "source": _lib.TF_ATTN_SOURCE_ONLY}[part]
    if NO_SPLIT if no_split is None else no_split:
        flags |= _lib.TF_ATTN_NO_SPLIT
    if out.dtype == torch.float32:
        flags |= _lib.TF_ATTN_OUT_F32
    flags |= int(hints) | (0 if fused is None else _lib.TF_ATTN_FUSED if fused else _lib.TF_ATTN_NO_FUSED)
    key = (K, S, heads, dh, dt)
    nbytes = _attn_ws_bytes.get(key)
    if nbytes is None:
        nbytes = _attn_ws_bytes[key] = lib.tf_ext_attn_workspace_bytes(K, S, heads, dh, dt)
    ws = _workspace(nbytes, q.device, stream=stream)
    strides = (ctypes.c_int64 * 9)(q_bs, q_fs, k_bs, k_fs, v_bs, v_fs, o_bs, o_fs, ld_q)
    _launch(dev, "tf_ext_attn_fwd_strided", lib.tf_ext_attn_fwd_strided, qp, kp, vp, op, K, Kq, int(q_frame0), S, heads,
            dh, ld, ctypes.cast(strides, ctypes.c_void_p), float(scale), flags, dt, ws.data_ptr(), ws.numel(),
            stream=stream)
    return out


def head_pack(slabs: Sequence[torch.Tensor], W: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """slabs: ns <= 6 tensors [Kl, S, D] (frame stride free, rows dense-strided, same dtype) -> the all-to-all send
    buffer [W, Kl, ns, S, D // W]: head group w of every slab, frame-major (tf_head_pack, one launch)."""
    dev = _need_gpu(*slabs)
    lib = _lib.load()
    Kl, S, D = slabs[0].shape
    ns, hd = len(slabs), D // W
    ld = slabs[0].stride(1)
    if D % W or any(t.shape != (Kl, S, D) or t.stride(2) != 1 or t.stride(1) != ld or t.dtype != slabs[0].dtype
                    for t in slabs):
        raise ValueError("head_pack: slabs must be [Kl, S, D] views sharing dtype and token stride, D divisible by W")
    if out is None:
        out = torch.empty(W, Kl, ns, S, hd, dtype=slabs[0].dtype, device=slabs[0].device)
    send = out
    ptrs = (ctypes.c_void_p * ns)(*[t.data_ptr() for t in slabs])
    fs = (ctypes.c_int64 * ns)(*[t.stride(0) for t in slabs])
    _launch(dev, "tf_head_pack", lib.tf_head_pack, ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(fs, ctypes.c_void_p),
            ns, send.data_ptr(), W, Kl, S, hd, ld, slabs[0].element_size())
    return send


def head_unpack(recv: torch.Tensor, dsts: Sequence[torch.Tensor]) -> None:
    """recv [W, Kl, nb, S, hd] (what the second all-to-all delivers) -> dsts[b][f, s, w*hd:(w+1)*hd], nb tensors
    [Kl, S, W*hd] (tf_head_unpack, one launch)."""
    dev = _need_gpu(recv, *dsts)
    lib = _lib.load()
    W, Kl, nb, S, hd = recv.shape
    ld = dsts[0].stride(1)
    if (len(dsts) != nb or not recv.is_contiguous()
            or any(t.shape != (Kl, S, W * hd) or t.stride(2) != 1 or t.stride(1) != ld or t.dtype != recv.dtype
                   for t in dsts)):
        raise ValueError("head_unpack: bad arguments")
    ptrs = (ctypes.c_void_p * nb)(*[t.data_ptr() for t in dsts])
    fs = (ctypes.c_int64 * nb)(*[t.stride(0) for t in dsts])
    _launch(dev, "tf_head_unpack", lib.tf_head_unpack, recv.data_ptr(), ctypes.cast(ptrs, ctypes.c_void_p),
            ctypes.cast(fs, ctypes.c_void_p), nb, W, Kl, S, hd, ld, recv.element_size())


def pivot_inv_norm(piv: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """1/||row|| for pivots [..., D] (bf16/f16, contiguous) -> fp32 [...] (written into `out` when given: a
    contiguous fp32 tensor of that shape, e.g. slots of a halo-extended buffer)."""
    dev = _need_gpu(piv, out)
    lib = _lib.load()
    piv = piv.contiguous()
    D = piv.shape[-1]
    rows = piv.numel() // D
    if out is None:
        out = torch.empty(piv.shape[:-1], dtype=torch.float32, device=piv.device)
    elif out.dtype != torch.float32 or out.shape != piv.shape[:-1] or not out.is_contiguous():
        raise ValueError("pivot_inv_norm: `out` must be a contiguous fp32 tensor of shape piv.shape[:-1]")
    _launch(dev, "tf_pivot_inv_norm", lib.tf_pivot_inv_norm, piv.data_ptr(), out.data_ptr(), rows, D, _DT[piv.dtype])
    return out


def nn_search(tgt: torch.Tensor, piv: torch.Tensor, inv_norm: torch.Tensor, kf_ids: Sequence[int]) -> torch.Tensor:
    """tgt [n*S, D], piv [K, S, D] (same 16-bit dtype), inv_norm fp32 [K, S]; kf_ids = 1 or 2 keyframe
    indices in the reference's order [i, i-1] (tokenflow_utils.py:331-333).  Returns int32 [P, n*S]."""
    dev = _need_gpu(tgt, piv, inv_norm)
    lib = _lib.load()
    tgt, piv = tgt.contiguous(), piv.contiguous()
    K, S, D = piv.shape
    n_tgt = tgt.shape[0]
    P = len(kf_ids)
    if tgt.dtype != piv.dtype or tgt.shape[1] != D or P not in (1, 2) or any(not 0 <= i < K for i in kf_ids):
        raise ValueError("nn_search: bad arguments")
    idx = torch.empty(P, n_tgt, dtype=torch.int32, device=tgt.device)
    ws = _workspace(lib.tf_nn_search_workspace_bytes(n_tgt, S, D, P), tgt.device, "nn")
    _launch(dev, "tf_nn_search", lib.tf_nn_search, tgt.data_ptr(), piv.data_ptr(), inv_norm.data_ptr(), idx.data_ptr(),
            n_tgt, S, D, P, int(kf_ids[0]), int(kf_ids[1]) if P == 2 else 0, _DT[tgt.dtype], ws.data_ptr(), ws.numel())
    return idx


def gather_blend(kf_out: torch.Tensor, idx: torch.Tensor, w: Optional[torch.Tensor], kf_ids: Sequence[int],
                 n: int, residual: Optional[torch.Tensor], out_dtype: torch.dtype) -> torch.Tensor:
    """kf_out [3K,S,D]; idx int32 [P, n*S]; w fp32 [n] (P == 2); residual [3n,S,D] or None.
    Returns [3n,S,D] of out_dtype (tokenflow_utils.py:362-397)."""
    dev = _need_gpu(kf_out, idx, w, residual)
    lib = _lib.load()
    kf_out = kf_out.contiguous()
    BK, S, D = kf_out.shape
    K = BK // 3
    P = len(kf_ids)
    if residual is not None:
        residual = residual.contiguous()
    out = torch.empty(3 * n, S, D, dtype=out_dtype, device=kf_out.device)
    _launch(dev, "tf_gather_blend", lib.tf_gather_blend, kf_out.data_ptr(), idx.data_ptr(),
            w.data_ptr() if w is not None else 0, residual.data_ptr() if residual is not None else 0, out.data_ptr(),
            K, n, S, D, P, int(kf_ids[0]), int(kf_ids[1]) if P == 2 else 0, _DT[kf_out.dtype],
            _DT[residual.dtype] if residual is not None else 0, _DT[out_dtype])
    return out


def layer_norm(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor], eps: float,
               out_dtype: torch.dtype, want_inv_norm: bool = False, out: Optional[torch.Tensor] = None,
               inv_out: Optional[torch.Tensor] = None):
    """LayerNorm over the last dim of x [..., D] (fp32 statistics, one rounding to out_dtype) and, on request,
    1/||row||_2 of the rounded output rows (fp32 [...]).  Returns (out, inv_norm or None).
    out / inv_out: contiguous destination tensors of x's shape (out_dtype) / x.shape[:-1] (fp32) to write into -- e.g.
    views of a block's halo-extended propagation state (the sharded hook pass: no staging copy behind the norm)."""
    dev = _need_gpu(x, weight, bias)
    lib = _lib.load()
    x = x.contiguous()
    D = x.shape[-1]
    rows = x.numel() // D
    if weight is not None and bias is not None and weight.dtype != bias.dtype:
        bias = bias.to(weight.dtype)
    wt = weight if weight is not None else bias
    if wt is not None and wt.dtype not in _DT:
        raise TypeError(f"layer_norm: weight dtype {wt.dtype}")
    weight = weight.contiguous() if weight is not None else None
    bias = bias.contiguous() if bias is not None else None
    if out is None:
        out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    elif out.shape != x.shape or out.dtype != out_dtype or not out.is_contiguous() or out.device != x.device:
        raise ValueError("layer_norm: `out` must be a contiguous tensor of x's shape, out_dtype and device")
    inv = None
    if want_inv_norm:
        if inv_out is None:
            inv = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
        elif (inv_out.shape != x.shape[:-1] or inv_out.dtype != torch.float32 or not inv_out.is_contiguous()
              or inv_out.device != x.device):
            raise ValueError("layer_norm: `inv_out` must be a contiguous fp32 tensor of shape x.shape[:-1]")
        else:
            inv = inv_out
    _launch(dev, "tf_layer_norm", lib.tf_layer_norm, x.data_ptr(), weight.data_ptr() if weight is not None else 0,
            bias.data_ptr() if bias is not None else 0, out.data_ptr(), inv.data_ptr() if inv is not None else 0,
            rows, D, float(eps), _DT[x.dtype], _DT[wt.dtype] if wt is not None else 0, _DT[out_dtype])
    return out, inv


def add_layer_norm(a: torch.Tensor, b: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor],
                   eps: float, out_dtype: torch.dtype):
    """(a + b, LayerNorm(a + b)) in one pass (tf_add_layer_norm): the sum has torch's promoted dtype and rounding, the
    norm is `layer_norm` of that rounded sum -- bit-identical to the two separate ops."""
    dev = _need_gpu(a, b, weight, bias)
    lib = _lib.load()
    if a.shape != b.shape:
        raise ValueError("add_layer_norm: a and b must have one shape")
    a, b = a.contiguous(), b.contiguous()
    D = a.shape[-1]
    rows = a.numel() // D
    if weight is not None and bias is not None and weight.dtype != bias.dtype:
        bias = bias.to(weight.dtype)
    wt = weight if weight is not None else bias
    weight = weight.contiguous() if weight is not None else None
    bias = bias.contiguous() if bias is not None else None
    total = torch.empty(a.shape, dtype=torch.promote_types(a.dtype, b.dtype), device=a.device)
    out = torch.empty(a.shape, dtype=out_dtype, device=a.device)
    _launch(dev, "tf_add_layer_norm", lib.tf_add_layer_norm, a.data_ptr(), b.data_ptr(), total.data_ptr(),
            weight.data_ptr() if weight is not None else 0, bias.data_ptr() if bias is not None else 0, out.data_ptr(),
            rows, D, float(eps), _DT[a.dtype], _DT[b.dtype], _DT[total.dtype], _DT[wt.dtype] if wt is not None else 0,
            _DT[out_dtype])
    return total, out


def norm_fusable(kf_out: torch.Tensor, residual: Optional[torch.Tensor], out_dtype: torch.dtype, P: int,
                 norm_dtype: torch.dtype) -> bool:
    """Can `propagate` / `propagate_chunks` carry the block's next LayerNorm (their `norm=` argument)?  The fused
    form covers the hook path's types: a 16-bit cached attention output, a residual and a norm output of that same
    type, the result in fp32 (two keyframes blended) or that type (one keyframe)."""
    dt = kf_out.dtype
    return (dt in (torch.bfloat16, torch.float16) and residual is not None and residual.dtype == dt
            and norm_dtype == dt and out_dtype == (torch.float32 if P == 2 else dt) and kf_out.shape[-1] <= 1536)


def _norm_args(norm, like: torch.Tensor, shape):
    """(gamma ptr, beta ptr, eps, w dtype code, norm_out tensor, norm dtype code) of a `norm=(weight, bias, eps,
    dtype)` argument."""
    weight, bias, eps, ndt = norm
    if weight is not None and bias is not None and weight.dtype != bias.dtype:
        bias = bias.to(weight.dtype)
    wt = weight if weight is not None else bias
    if wt is not None and wt.dtype not in _DT:
        raise TypeError(f"propagate: norm weight dtype {wt.dtype}")
    weight = weight.contiguous() if weight is not None else None
    bias = bias.contiguous() if bias is not None else None
    nout = torch.empty(shape, dtype=ndt, device=like.device)
    return (weight.data_ptr() if weight is not None else 0, bias.data_ptr() if bias is not None else 0, float(eps),
            _DT[wt.dtype] if wt is not None else 0, nout, _DT[ndt], (weight, bias))


def propagate(tgt: torch.Tensor, piv: torch.Tensor, inv_norm: torch.Tensor, kf_ids: Sequence[int],
              kf_out: torch.Tensor, w: Optional[torch.Tensor], n: int, residual: Optional[torch.Tensor],
              out_dtype: torch.dtype, norm=None):
    """nn_search + gather_blend of one chunk in one call (tokenflow_utils.py:329-397): same arguments, same
    results bit for bit, one launch less (the gather merges the search's per-split candidates itself).
    norm = (weight, bias, eps, dtype) of the block's next LayerNorm (see `norm_fusable`): returns
    (result, LayerNorm(result) in `dtype`) from one gather launch -- tf_nn_gather_blend_norm, both bit-identical to
    the separate calls."""
    dev = _need_gpu(tgt, piv, inv_norm, kf_out, w, residual)
    lib = _lib.load()
    tgt, piv, kf_out = tgt.contiguous(), piv.contiguous(), kf_out.contiguous()
    K, S, D = piv.shape
    n_tgt = tgt.shape[0]
    P = len(kf_ids)
    if (tgt.dtype != piv.dtype or tgt.shape[1] != D or P not in (1, 2) or any(not 0 <= i < K for i in kf_ids)
            or n_tgt != n * S or kf_out.shape != (3 * K, S, D)):
        raise ValueError("propagate: bad arguments")
    if residual is not None:
        residual = residual.contiguous()
    out = torch.empty(3 * n, S, D, dtype=out_dtype, device=kf_out.device)
    ws = _workspace(lib.tf_nn_gather_blend_workspace_bytes(n_tgt, S, D, P), tgt.device, "nn")
    if norm is not None:
        if not norm_fusable(kf_out, residual, out_dtype, P, norm[3]):
            raise TypeError("propagate: these dtypes have no fused-norm form (ops.norm_fusable)")
        g, b, eps, wdt, nout, ndt, _keep = _norm_args(norm, kf_out, (3 * n, S, D))
        _launch(dev, "tf_nn_gather_blend_norm", lib.tf_nn_gather_blend_norm, tgt.data_ptr(), piv.data_ptr(),
                inv_norm.data_ptr(), kf_out.data_ptr(), w.data_ptr() if w is not None else 0, residual.data_ptr(),
                out.data_ptr(), K, n, S, D, P, int(kf_ids[0]), int(kf_ids[1]) if P == 2 else 0, _DT[tgt.dtype],
                _DT[kf_out.dtype], _DT[residual.dtype], _DT[out_dtype], g, b, eps, wdt, nout.data_ptr(), ndt,
                ws.data_ptr(), ws.numel())
        return out, nout
    _launch(dev, "tf_nn_gather_blend", lib.tf_nn_gather_blend, tgt.data_ptr(), piv.data_ptr(), inv_norm.data_ptr(),
            kf_out.data_ptr(), w.data_ptr() if w is not None else 0,
            residual.data_ptr() if residual is not None else 0, out.data_ptr(),
            K, n, S, D, P, int(kf_ids[0]), int(kf_ids[1]) if P == 2 else 0, _DT[tgt.dtype], _DT[kf_out.dtype],
            _DT[residual.dtype] if residual is not None else 0, _DT[out_dtype], ws.data_ptr(), ws.numel())
    return out


def propagate_chunks(tgt: torch.Tensor, piv: torch.Tensor, inv_norm: torch.Tensor, kf_out: torch.Tensor,
                     w: torch.Tensor, n: int, n_chunks: int, slot0: int, first_single: bool,
                     residual: Optional[torch.Tensor], out_dtype: torch.dtype, norm=None):
    """`propagate` for a run of `n_chunks` consecutive chunks of n frames in one call (tf_nn_gather_blend_chunks):
    tgt [n_chunks*n*S, D] chunk-major, residual / result [3*n_chunks*n, S, D]; chunk j matches keyframe slots
    slot0 + j and slot0 + j - 1 of piv / inv_norm / kf_out; first_single: chunk 0 of the run is chunk 0 of the
    video (one keyframe).  Bit-identical to n_chunks calls of `propagate`."""
    dev = _need_gpu(tgt, piv, inv_norm, kf_out, w, residual)
    lib = _lib.load()
    tgt, piv, kf_out = tgt.contiguous(), piv.contiguous(), kf_out.contiguous()
    K, S, D = piv.shape
    C = int(n_chunks)
    if C == 1:
        ids = [slot0] if first_single else [slot0, slot0 - 1]
        return propagate(tgt, piv, inv_norm, ids, kf_out, None if first_single else w, n, residual, out_dtype,
                         norm=norm)
    if (tgt.dtype != piv.dtype or tgt.shape != (C * n * S, D) or kf_out.shape != (3 * K, S, D) or w is None
            or slot0 + C > K or slot0 < (0 if first_single else 1)):
        raise ValueError("propagate_chunks: bad arguments")
    if residual is not None:
        residual = residual.contiguous()
    # what the reference's single-keyframe pass would emit for chunk 0: kf dtype (+ residual), torch promotion
    single_dtype = kf_out.dtype if residual is None else torch.promote_types(kf_out.dtype, residual.dtype)
    out = torch.empty(3 * C * n, S, D, dtype=out_dtype, device=kf_out.device)
    ws = _workspace(lib.tf_nn_gather_blend_chunks_workspace_bytes(n * S, S, D, C), tgt.device, "nn")
    if norm is not None:
        if not norm_fusable(kf_out, residual, out_dtype, 2, norm[3]):
            raise TypeError("propagate_chunks: these dtypes have no fused-norm form (ops.norm_fusable)")
        g, b, eps, wdt, nout, ndt, _keep = _norm_args(norm, kf_out, (3 * C * n, S, D))
        _launch(dev, "tf_nn_gather_blend_chunks_norm", lib.tf_nn_gather_blend_chunks_norm, tgt.data_ptr(),
                piv.data_ptr(), inv_norm.data_ptr(), kf_out.data_ptr(), w.data_ptr(), residual.data_ptr(),
                out.data_ptr(), K, n, C, S, D, int(slot0), 1 if first_single else 0, _DT[tgt.dtype], _DT[kf_out.dtype],
                _DT[residual.dtype], _DT[out_dtype], _DT[single_dtype], g, b, eps, wdt, nout.data_ptr(), ndt,
                ws.data_ptr(), ws.numel())
        return out, nout
    _launch(dev, "tf_nn_gather_blend_chunks", lib.tf_nn_gather_blend_chunks, tgt.data_ptr(), piv.data_ptr(),
            inv_norm.data_ptr(), kf_out.data_ptr(), w.data_ptr(), residual.data_ptr() if residual is not None else 0,
            out.data_ptr(), K, n, C, S, D, int(slot0), 1 if first_single else 0, _DT[tgt.dtype], _DT[kf_out.dtype],
            _DT[residual.dtype] if residual is not None else 0, _DT[out_dtype], _DT[single_dtype],
            ws.data_ptr(), ws.numel())
    return out


def inject_copy_(x: torch.Tensor) -> torch.Tensor:
    """In place: x[n:2n] = x[:n]; x[2n:] = x[:n] with n = len(x)//3 (tokenflow_utils.py:87-91)."""
    dev = _need_gpu(x)
    lib = _lib.load()
    if x.shape[0] % 3 or not x.is_contiguous():
        raise ValueError("inject_copy_: need a contiguous tensor whose batch is a multiple of 3")
    per_branch = x.numel() // 3
    _launch(dev, "tf_inject_copy", lib.tf_inject_copy, x.data_ptr(), per_branch, x.element_size())
    return x


def ddim_step(x: torch.Tensor, eps: torch.Tensor, mu_a: float, sigma_a: float, mu_b: float, sigma_b: float,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = mu_b * ((x - sigma_a*eps) / mu_a) + sigma_b*eps  (preprocess.py:224-225 / 259-260), one launch, the
    reference's per-op rounding.  x, eps: same shape and dtype (f16 / bf16 / f32), contiguous; out may be x."""
    dev = _need_gpu(x, eps, out)
    lib = _lib.load()
    if out is None:
        out = torch.empty_like(x)
    if (eps.shape != x.shape or eps.dtype != x.dtype or out.shape != x.shape or out.dtype != x.dtype
            or x.dtype not in _DT or not (x.is_contiguous() and eps.is_contiguous() and out.is_contiguous())):
        raise ValueError("ddim_step: x, eps, out must be contiguous tensors of one shape and dtype (f16/bf16/f32)")
    _launch(dev, "tf_ddim_step", lib.tf_ddim_step, x.data_ptr(), eps.data_ptr(), out.data_ptr(), x.numel(),
            float(mu_a), float(sigma_a), float(mu_b), float(sigma_b), _DT[x.dtype])
    return out
