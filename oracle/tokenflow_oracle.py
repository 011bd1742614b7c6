"""CPU oracle for TokenFlow's per-step hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (plain torch ops on CPU tensors, fp32 unless the
caller passes fp64) of the algorithm in the reference's hook layer.  Nothing in
the product path (`tokenflow_amd/`, `tokenflow_utils.py`, `util.py`) may import
it: only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg do, and only as the checker / the CPU baseline.

Parity pin: the reference ships no tests or golden vectors, so
the oracle is pinned against the *verbatim reference code itself*, imported
from /root/reference by `oracle/ref_loader.py` and executed on seeded inputs:
`oracle/make_golden.py` writes those inputs' outputs to `tests/golden/*.pt`,
and `tests/test_oracle_golden.py` requires this restatement to reproduce them
(bit-for-bit where the op order is identical, <=2e-6 otherwise).

Each function cites the reference lines it restates (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch


# ---------------------------------------------------------------------------
# (A) extended attention  --  tokenflow_utils.py:114-199 (pnp) / 224-281 (sdedit)
# ---------------------------------------------------------------------------

def ext_attn_core(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int,
                  scale: float, inject: bool = False) -> torch.Tensor:
    """Attention core of `sa_forward.forward` after the q/k/v projections and
    before `to_out`.

    q, k, v: [3*K, S, D], batch laid out [source | uncond | cond]
    (tokenflow_utils.py:117 `n_frames = batch_size // 3`).

    * inject (tokenflow_utils.py:124-130): uncond and cond q,k <- source q,k.
    * source frames attend to their own S keys (173,177); uncond / cond frames
      attend to the concatenation of that branch's keys over all K frames
      (133-138,174-175,178-179).
    * output is frame-major [3K, S, D] with feature index = head*d + c
      (181-197: the cat/view/permute/batch_to_head_dim re-layout).
    Independent of the `single_batch` (K<=12) switch at 165-168, which only
    changes how the same numbers are batched.
    """
    B, S, D = q.shape
    K = B // 3
    d = D // heads
    dv = v.shape[-1] // heads            # the value width may differ from the q/k width (tests stack [V, |V|])
    q = q.reshape(3, K, S, heads, d)
    k = k.reshape(3, K, S, heads, d)
    v = v.reshape(3, K, S, heads, dv)
    if inject:
        q = torch.stack([q[0], q[0], q[0]])
        k = torch.stack([k[0], k[0], k[0]])
    out = torch.empty(3, K, S, heads, dv, dtype=q.dtype)
    # source branch: per-frame attention
    sim = torch.einsum("fqhc,fkhc->fhqk", q[0], k[0]) * scale
    out[0] = torch.einsum("fhqk,fkhc->fqhc", sim.softmax(dim=-1), v[0])
    # uncond / cond: bank of K*S keys shared by every frame of the branch
    for b in (1, 2):
        kb = k[b].reshape(K * S, heads, d)
        vb = v[b].reshape(K * S, heads, dv)
        for f in range(K):  # frame loop bounds the [h,S,K*S] score matrix
            sim = torch.einsum("qhc,khc->hqk", q[b, f], kb) * scale
            out[b, f] = torch.einsum("hqk,khc->qhc", sim.softmax(dim=-1), vb)
    return out.reshape(3 * K, S, heads * dv)


def ext_attn_core_bmm(q, k, v, heads, scale, inject=False, frames=None):
    """Same numbers as `ext_attn_core`, with the reference's own cost
    structure (per-head `bmm` -> `* scale` -> `softmax` -> `bmm`,
    tokenflow_utils.py:172-179, and the per-frame loop of 165-168 when K > 12).
    Used as the timed CPU baseline ("port") in bench.py.  The K-fold physical
    replication of the bank (133-138 `.repeat`) is not reproduced (it only
    costs the reference memory and time).

    frames: optional list of query-frame indices -- only those frames' outputs
    are computed (against the full K-frame bank) and returned as [3*len(frames), S, D];
    bench.py times ONE frame this way and scales by K (a bounded sample of the
    same per-frame work)."""
    B, S, D = q.shape
    K = B // 3
    d = D // heads
    if frames is not None:
        sel = torch.tensor(list(frames))
        qsel = q.view(3, K, S, D)[:, sel].reshape(-1, S, D)
        full = _ext_attn_bmm_frames(qsel, k, v, heads, scale, inject, sel)
        return full
    if inject:
        q = torch.cat([q[:K]] * 3)
        k = torch.cat([k[:K]] * 3)

    def hb(t):  # head_to_batch_dim, then [frames, h, L, d]
        f, L, _ = t.shape
        return t.reshape(f, L, heads, d).permute(0, 2, 1, 3)

    qs, ks, vs = hb(q[:K]), hb(k[:K]), hb(v[:K])
    outs = [torch.empty(K, S, heads, d, dtype=q.dtype) for _ in range(3)]
    banks = []
    for b in (1, 2):
        banks.append((hb(q[b * K:(b + 1) * K]),
                      hb(k[b * K:(b + 1) * K].reshape(1, K * S, D))[0],
                      hb(v[b * K:(b + 1) * K].reshape(1, K * S, D))[0]))
    step = K if K <= 12 else 1
    for f0 in range(0, K, step):
        for j in range(heads):
            sim = torch.bmm(qs[f0:f0 + step, j], ks[f0:f0 + step, j].transpose(-1, -2)) * scale
            outs[0][f0:f0 + step, :, j] = torch.bmm(sim.softmax(dim=-1), vs[f0:f0 + step, j])
            for bi, (qb, kb, vb) in enumerate(banks):
                sim = torch.matmul(qb[f0:f0 + step, j], kb[j].transpose(-1, -2)) * scale
                outs[1 + bi][f0:f0 + step, :, j] = torch.matmul(sim.softmax(dim=-1), vb[j])
    return torch.cat([o.reshape(K, S, D) for o in outs])


def _ext_attn_bmm_frames(qsel, k, v, heads, scale, inject, sel):
    """`ext_attn_core_bmm` for the query frames `sel` only: qsel [3*F, S, D]; k, v the full [3K, S, D]."""
    B, S, D = k.shape
    K, F = B // 3, len(sel)
    d = D // heads
    q3, k3, v3 = qsel.view(3, F, S, D), k.view(3, K, S, D), v.view(3, K, S, D)
    if inject:
        q3 = torch.stack([q3[0]] * 3)
        k3 = torch.stack([k3[0]] * 3)

    def hb(t):  # [frames, L, D] -> [frames, h, L, d]
        f, L, _ = t.shape
        return t.reshape(f, L, heads, d).permute(0, 2, 1, 3)

    out = torch.empty(3, F, S, heads, d, dtype=qsel.dtype)
    qs, ks, vs = hb(q3[0]), hb(k3[0][sel]), hb(v3[0][sel])
    for j in range(heads):
        sim = torch.bmm(qs[:, j], ks[:, j].transpose(-1, -2)) * scale          # 173
        out[0, :, :, j] = torch.bmm(sim.softmax(dim=-1), vs[:, j])              # 177
        for b in (1, 2):
            kb = hb(k3[b].reshape(1, K * S, D))[0, j]
            vb = hb(v3[b].reshape(1, K * S, D))[0, j]
            sim = torch.matmul(hb(q3[b])[:, j], kb.transpose(-1, -2)) * scale   # 174-175
            out[b, :, :, j] = torch.matmul(sim.softmax(dim=-1), vb)             # 178-179
    return out.reshape(3 * F, S, D)


def should_inject(t, injection_schedule) -> bool:
    """tokenflow_utils.py:86,124:
    `schedule is not None and (t in schedule or t == 1000)`."""
    if injection_schedule is None:
        return False
    if t == 1000:
        return True
    if isinstance(injection_schedule, torch.Tensor):
        return bool((injection_schedule == t).any().item()) if injection_schedule.numel() else False
    return t in injection_schedule


def sa_forward(attn, x: torch.Tensor, inject: bool) -> torch.Tensor:
    """Whole `sa_forward.forward` closure (tokenflow_utils.py:114-199 and
    224-281): projections, core, `to_out[0]` (dropout `to_out[1]` skipped,
    108-112)."""
    to_out = attn.to_out[0] if isinstance(attn.to_out, torch.nn.ModuleList) else attn.to_out
    q, k, v = attn.to_q(x), attn.to_k(x), attn.to_v(x)
    return to_out(ext_attn_core(q, k, v, attn.heads, attn.scale, inject))


# ---------------------------------------------------------------------------
# (B) token propagation  --  tokenflow_utils.py:329-348, 361-397; util.py:61-69
# ---------------------------------------------------------------------------

def batch_cosine_sim(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """util.py:61-69."""
    x = x / x.norm(dim=-1, keepdim=True)
    y = y / y.norm(dim=-1, keepdim=True)
    return x @ y.T


def keyframe_ids(batch_idx: int) -> List[int]:
    """tokenflow_utils.py:331-333: `[i]` for chunk 0, else `[i, i-1]`."""
    return [batch_idx] if batch_idx == 0 else [batch_idx, batch_idx - 1]


def nn_search(norm_src: torch.Tensor, pivots_src: torch.Tensor, batch_idx: int
              ) -> Tuple[List[torch.Tensor], torch.Tensor]:
    """tokenflow_utils.py:331-343.

    norm_src:   [n, S, D]  norm1 output of the SOURCE branch of this chunk
    pivots_src: [K, S, D]  `pivot_hidden_states[0]` cached in the pivotal pass
    returns ([idx per keyframe in `keyframe_ids` order], sim [n*S, P*S]).
    `argmax` returns the first maximal index."""
    n, S, D = norm_src.shape
    ids = keyframe_ids(batch_idx)
    sim = batch_cosine_sim(norm_src.reshape(-1, D), pivots_src[ids].reshape(-1, D))
    idx = [c.argmax(dim=-1) for c in sim.chunk(len(ids), dim=1)]
    return idx, sim


def blend_weights(n: int, batch_idx: int) -> torch.Tensor:
    """tokenflow_utils.py:375-383.  Depends on n only (batch_idx cancels):
    d1 = |j - n//2|, d2 = |j + n - n//2|, w1 = sigmoid(d2 / (d1 + d2))."""
    ids = keyframe_ids(batch_idx)
    s = torch.arange(0, n) + ids[0] * n
    p1 = ids[0] * n + n // 2
    p2 = ids[1] * n + n // 2
    d1 = torch.abs(s - p1)
    d2 = torch.abs(s - p2)
    return torch.sigmoid(d2 / (d1 + d2))


def gather_blend(kf_attn_output: torch.Tensor, idx: Sequence[torch.Tensor], batch_idx: int,
                 n: int, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """tokenflow_utils.py:362-397 (propagation branch).

    kf_attn_output: [3*K, S, D] cached `attn1` output of the pivotal pass
    idx:            [n*S] int64 per keyframe (same indices for all 3 branches, 344-348)
    returns [3*n, S, D]; dtype follows torch promotion: fp32 whenever two
    keyframes are blended (w1 is fp32, 385-388), the input dtype for chunk 0 (390).
    `residual` = hidden_states [3n,S,D] added at 396-397."""
    BK, S, D = kf_attn_output.shape
    K = BK // 3
    ids = keyframe_ids(batch_idx)
    sel = kf_attn_output.view(3, K, S, D)[:, ids]
    i1 = torch.stack([idx[0]] * 3, 0).unsqueeze(-1).repeat(1, 1, D)
    if len(ids) == 2:
        i2 = torch.stack([idx[1]] * 3, 0).unsqueeze(-1).repeat(1, 1, D)
        a1 = sel[:, 0].gather(dim=1, index=i1).view(3, n, S, D)
        a2 = sel[:, 1].gather(dim=1, index=i2).view(3, n, S, D)
        w1 = blend_weights(n, batch_idx).view(1, n, 1, 1)
        out = w1 * a1 + (1 - w1) * a2
    else:
        out = sel[:, 0].gather(dim=1, index=i1)
    out = out.reshape(3 * n, S, D)
    if residual is not None:
        out = out + residual
    return out


# ---------------------------------------------------------------------------
# (C) PnP feature injection  --  tokenflow_utils.py:86-91
# ---------------------------------------------------------------------------

def conv_inject_(hidden_states: torch.Tensor) -> torch.Tensor:
    n = int(hidden_states.shape[0] // 3)
    hidden_states[n:2 * n] = hidden_states[:n]
    hidden_states[2 * n:] = hidden_states[:n]
    return hidden_states


# ---------------------------------------------------------------------------
# (D) DDIM latent update  --  preprocess.py:224-225 (inversion), 259-260 (sampling)
# ---------------------------------------------------------------------------

def ddim_step(x: torch.Tensor, eps: torch.Tensor, mu_a: float, sigma_a: float, mu_b: float,
              sigma_b: float) -> torch.Tensor:
    """`pred_x0 = (x - sigma_a*eps) / mu_a;  x' = mu_b*pred_x0 + sigma_b*eps` with the reference's operation
    order and per-op rounding: every torch op on a 16-bit tensor is evaluated in fp32 and rounded to the tensor
    dtype, the coefficients are fp32 scalars (on the reference's CUDA path `scheduler.alphas_cumprod` is a host
    tensor, i.e. a scalar operand kept in fp32 opmath.  A CPU run of the same lines rounds the three MULTIPLIED
    0-dim coefficients to the tensor dtype first and keeps the divisor in fp32: the f16 golden of
    tests/golden/inversion.pt, written by a CPU run of the verbatim reference, is reproduced bit for bit by this
    function when it is handed those three coefficients pre-rounded -- tests/test_hooks_cpu.py -- which pins every
    rounding point of the tensor arithmetic to the reference)."""
    dt = x.dtype
    r = (lambda t: t) if dt == torch.float32 else (lambda t: t.to(dt).float())
    xf, ef = x.float(), eps.float()
    f32 = lambda v: torch.tensor(v, dtype=torch.float32)
    pred_x0 = r(r(xf - r(f32(sigma_a) * ef)) / f32(mu_a))
    return r(r(f32(mu_b) * pred_x0) + r(f32(sigma_b) * ef)).to(dt)


# ---------------------------------------------------------------------------
# whole-block restatement  --  tokenflow_utils.py:300-427
# ---------------------------------------------------------------------------

class BlockState:
    """The per-block mutable state the reference keeps on the module
    (pivot_hidden_states 327, kf_attn_output 360)."""

    def __init__(self):
        self.pivot_hidden_states = None   # [3, K, S, D]
        self.kf_attn_output = None        # [3K, S, D]


def block_forward(block, state: BlockState, hidden_states: torch.Tensor, *, pivotal: bool,
                  batch_idx: int = 0, inject: bool = False,
                  encoder_hidden_states: Optional[torch.Tensor] = None, timestep=None,
                  class_labels=None) -> torch.Tensor:
    """`TokenFlowBlock.forward` for a plain-LayerNorm block (the SD case:
    use_ada_layer_norm* False, only_cross_attention False) and for an
    AdaLayerNormZero block (317-320, 365-366, 417-424)."""
    B, S, D = hidden_states.shape
    n = B // 3
    zero = bool(getattr(block, "use_ada_layer_norm_zero", False))
    if zero:                                                                 # 317-320
        norm, gate_msa, shift_mlp, scale_mlp, gate_mlp = block.norm1(
            hidden_states.view(3, n, S, D), timestep, class_labels, hidden_dtype=hidden_states.dtype)
    else:
        norm = block.norm1(hidden_states.view(3, n, S, D))                   # 314-325
    norm = norm.view(3, n, S, D)
    if pivotal:
        state.pivot_hidden_states = norm                                     # 326-327
        attn_output = sa_forward(block.attn1, norm.view(B, S, D), inject)    # 352-358
        state.kf_attn_output = attn_output                                   # 360
        if zero:
            attn_output = gate_msa.unsqueeze(1) * attn_output                # 365-366
    else:
        idx, _ = nn_search(norm[0], state.pivot_hidden_states[0], batch_idx)  # 329-348
        kf = state.kf_attn_output
        if zero:   # 362-366: the gate multiplies the selected keyframes' outputs; gating all K is the same numbers
            K = kf.shape[0] // 3
            kf = (gate_msa.unsqueeze(1) * kf.view(3, K, S, D)).reshape(3 * K, S, D)
        attn_output = gather_blend(kf, idx, batch_idx, n)                    # 361-393
    h = attn_output + hidden_states.reshape(B, S, D)                         # 396-397
    if block.attn2 is not None:                                              # 399-411
        h = block.attn2(block.norm2(h), encoder_hidden_states=encoder_hidden_states) + h
    nh = block.norm3(h)                                                      # 413-425
    if zero:
        nh = nh * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        return gate_mlp.unsqueeze(1) * block.ff(nh) + h
    return block.ff(nh) + h


# ---------------------------------------------------------------------------
# helpers shared by tests / bench
# ---------------------------------------------------------------------------

def bf16_round(t: torch.Tensor) -> torch.Tensor:
    """Round once to bf16 and return as fp32: the oracle is fed exactly the
    values the bf16 kernels see."""
    return t.to(torch.bfloat16).to(torch.float32)


def nn_mismatch_tie_aware(sim: torch.Tensor, ref_idx: torch.Tensor, got_idx: torch.Tensor,
                          tau: float = 1e-5) -> Tuple[int, int]:
    """(#rows whose index differs, #rows whose index differs by more than a
    near-tie): a row passes if `sim[ref] - sim[got] <= tau` (SURVEY.md §7 hard
    parts: argmax parity is discontinuous)."""
    rows = torch.arange(sim.shape[0])
    diff = (ref_idx != got_idx)
    gap = sim[rows, ref_idx] - sim[rows, got_idx.long()]
    bad = diff & (gap > tau)
    return int(diff.sum()), int(bad.sum())
