"""Parity of the HIP kernels (through the C ABI) against the CPU oracle and the
golden fixtures produced by the verbatim reference.  Needs an MI355X.

Tolerances (stated here, used below):
  * integer / index / copy work: bit-exact (NN indices: tie-aware, see below).
  * gather+blend+residual: bit-exact in every output dtype (same fp32 operation order).
  * extended attention (bf16/f16 MFMA, fp32 softmax/accumulate, P rounded to 16 bit):
        |out - ref| <= ATTN_ATOL + EPS * |ref| + EPS * (softmax(QK^T) . |V|)
    ATTN_ATOL = 2e-4 (fp32 accumulation order, v_exp_f32), EPS = 2^-8 (bf16 relative half-ulp:
    8 significand bits; 2^-11 for f16).  Second term = rounding of the output itself, third = worst case of rounding each
    probability before P.V (what the reference's autocast path does too).
    The scores are scaled in fp32 (the default).  With TF_ATTN_FOLD_SCALE (ops.ext_attn(fold_scale=True) /
    TOKENFLOW_FOLD_SCALE=1; Dh = 40 only) the kernel rounds q*scale*log2(e) to 16 bit once: a relative error
    <= 2^-9 per element of q that perturbs every score by a zero-mean amount of standard deviation
    sigma = 2^-9/sqrt(3) * scale * sqrt(sum_d (q_d k_d)^2);  runs with that flag are held to the bound plus a
    fourth term  4 * sigma_max * (|ref| + softmax.|V|)  -- which is why the flag is not the default.
    At BASELINE shapes (thousands of keys, |out| <~ 0.25) the third term averages out and the
    bound is the north-star's "< 1e-3"; tests/test_fullsize_gpu.py asserts that number directly.
  * NN indices: equal, or the oracle's fp32 similarity of the two candidates differs by
    <= NN_TAU = 1e-5 (argmax is discontinuous at near-ties).
"""
import pytest
import torch

from oracle import golden_cases as gc
from oracle import tokenflow_oracle as orc
from oracle.golden_util import check

pytestmark = pytest.mark.gpu

ATTN_ATOL = 2e-4
NN_TAU = 1e-5


def _ops():
    from tokenflow_amd import ops
    return ops


def attn_ref(q, k, v, h, scale, inject, need_sigma=True):
    """(oracle output, softmax.|V|, sigma) -- softmax.|V| drives the P-rounding term of the bound, sigma the
    q*scale rounding term of the folded-scale kernels (Dh = 40): per query, the largest standard deviation over
    its keys of the score perturbation caused by rounding q*c to 8 significant bits,
    sigma = 2^-9/sqrt(3) * scale * max_k sqrt(sum_d (q_d k_d)^2)."""
    B, S, D = q.shape
    K, d = B // 3, D // h
    if not need_sigma:   # one softmax for both: values stacked [V | |V|] per head
        v4 = v.view(B, S, h, d)
        both = orc.ext_attn_core(q, k, torch.cat([v4, v4.abs()], dim=-1).reshape(B, S, 2 * D), h, scale, inject)
        both = both.view(B, S, h, 2 * d)
        return both[..., :d].reshape(B, S, D), both[..., d:].reshape(B, S, D), None
    qs = (q.view(3, K, S, h, d) ** 2)
    ks = (k.view(3, K, S, h, d) ** 2)
    if inject:
        qs, ks = torch.stack([qs[0]] * 3), torch.stack([ks[0]] * 3)
    sig = torch.empty(3, K, S, h)
    sig[0] = torch.einsum("fqhc,fkhc->fhqk", qs[0], ks[0]).amax(-1).permute(0, 2, 1)
    for b in (1, 2):
        sig[b] = torch.einsum("fqhc,khc->fhqk", qs[b], ks[b].reshape(K * S, h, d)).amax(-1).permute(0, 2, 1)
    sig = (2.0 ** -9 / 3 ** 0.5) * scale * sig.sqrt()
    sig = sig[..., None].expand(3, K, S, h, d).reshape(B, S, D)
    return (orc.ext_attn_core(q, k, v, h, scale, inject), orc.ext_attn_core(q, k, v.abs(), h, scale, inject), sig)


def attn_bound(ref, ref_abs, dtype=torch.bfloat16, sigma=None):
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    b = ATTN_ATOL + eps * ref.abs() + eps * ref_abs
    if sigma is not None:   # folded softmax scale: 4 sigma of the score perturbation, on the same weights
        b = b + 4.0 * sigma * (1.0 if dtype == torch.bfloat16 else 2.0 ** -3) * (ref_abs + ref.abs())
    return b


def assert_attn_close(got, refs, what="", dtype=torch.bfloat16, folded=None):
    ref, ref_abs, sigma = refs
    folded = bool(folded)      # only explicit fold_scale=True runs carry the q*scale rounding term
    got = got.float().cpu()
    err = (got - ref).abs()
    worst = float((err - attn_bound(ref, ref_abs, dtype, sigma if folded else None)).max())
    assert worst <= 0, f"{what}: max abs err {float(err.max()):.3e}, exceeds bound by {worst:.3e}"
    return float(err.max())




# --------------------------------------------------------------------------- attention
@pytest.mark.parametrize("name", list(gc.ATTN_CASES))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("fused", [None, False])
def test_ext_attn_vs_oracle_and_golden(name, dtype, fused, golden_attn):
    """fused=None: the library's own choice (the small golden cases run in the fused kernel, csrc/ext_attn_fused.hip);
    fused=False: the streaming kernels on the same inputs (TF_ATTN_NO_FUSED)."""
    ops = _ops()
    K, S, h, d, sched, t = gc.ATTN_CASES[name]
    q, k, v = gc.attn_inputs(name)                       # fp32 holding bf16-representable values
    inject = orc.should_inject(t, sched)
    if dtype == torch.float16:                           # f16 cannot hold every bf16 value: re-round, re-run oracle
        q, k, v = (x.to(torch.float16).float() for x in (q, k, v))
    refs = attn_ref(q, k, v, h, d ** -0.5, inject)
    dq, dk, dv = (x.to(dtype).cuda() for x in (q, k, v))
    out = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, fused=fused)
    torch.cuda.synchronize()
    assert out.dtype == dtype and out.shape == refs[0].shape
    assert_attn_close(out, refs, f"{name}/{dtype}", dtype)
    # inputs must be untouched (the reference mutates q/k in place; we alias instead)
    assert torch.equal(dq.cpu().float(), q) and torch.equal(dk.cpu().float(), k)
    if dtype == torch.bfloat16:                          # pin to the verbatim reference's numbers
        g = golden_attn[name]
        st = g["out_pnp"]["stride"]
        f, r = out.float().cpu().flatten()[::st], g["out_pnp"]["sample"]
        assert float(((f - r).abs() - attn_bound(r, refs[1].flatten()[::st])).max()) <= 0
        out_sde = ops.ext_attn(dq, dk, dv, h, d ** -0.5, False, fused=fused)
        st = g["out_sdedit"]["stride"]
        f, r = out_sde.float().cpu().flatten()[::st], g["out_sdedit"]["sample"]
        r_sde = attn_ref(q, k, v, h, d ** -0.5, False, need_sigma=False)
        assert float(((f - r).abs() - attn_bound(r, r_sde[1].flatten()[::st])).max()) <= 0
    if d == 40:   # opt-in folded scale (TF_ATTN_FOLD_SCALE): the bound with the q*scale rounding term
        out_f = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, fold_scale=True)
        assert_attn_close(out_f, refs, f"{name}/fold_scale", dtype, folded=True)


@pytest.mark.parametrize("K,S,h,d", [(2, 256, 2, 40), (3, 136, 2, 64), (2, 200, 1, 80), (1, 16, 2, 160),
                                     (4, 64, 8, 40), (13, 64, 2, 64),
                                     # S >= 512 at Dh = 64 takes the ping-pong kernel: one tile, odd / even tile
                                     # counts, ragged last tile, several keyframes
                                     (1, 512, 1, 64), (2, 576, 2, 64), (2, 520, 1, 64), (3, 640, 2, 64),
                                     # S >= 256 at Dh = 40 / 80: 8-wave geometry and the dual (shared-softmax) form
                                     (2, 328, 2, 40), (2, 264, 1, 80),
                                     # token counts that are not multiples of 8 (odd latent grids: 9x5, 18x10+1 ...)
                                     (2, 45, 2, 160), (3, 181, 2, 80), (2, 723, 1, 40), (2, 515, 1, 64), (1, 1, 1, 40)])
@pytest.mark.parametrize("inject", [False, True])
@pytest.mark.parametrize("no_split", [False, True])
@pytest.mark.parametrize("fused", [None, False])
def test_ext_attn_shapes(K, S, h, d, inject, no_split, fused, monkeypatch):
    """Ragged S (not a multiple of 64 / 128), single keyframe, many heads, K > 12.  These grids are small: by default
    they run in the fused small-problem kernel; fused=False keeps the streaming kernels, where the bank problems run in
    the split form (runs of bank frames + merge) unless no_split forces the one-pass form on the same inputs."""
    if fused is None and no_split and S > 256:
        pytest.skip("larger frames in one-pass mode run the streaming kernels whatever `fused` says: covered by fused=False")
    ops = _ops()
    monkeypatch.setattr(ops, "NO_SPLIT", no_split)
    g = torch.Generator().manual_seed(K * 1000 + S + d)
    D = h * d
    q, k, v = (orc.bf16_round(torch.randn(3 * K, S, D, generator=g)) for _ in range(3))
    refs = attn_ref(q, k, v, h, d ** -0.5, inject)
    out = ops.ext_attn(q.bfloat16().cuda(), k.bfloat16().cuda(), v.bfloat16().cuda(), h, d ** -0.5, inject, fused=fused)
    assert_attn_close(out, refs, f"K{K} S{S} h{h} d{d} inj{inject} no_split{no_split} fused{fused}")
    if d == 40 and fused is False:
        out = ops.ext_attn(q.bfloat16().cuda(), k.bfloat16().cuda(), v.bfloat16().cuda(), h, d ** -0.5, inject,
                           fold_scale=True)
        assert_attn_close(out, refs, f"fold K{K} S{S} h{h} d{d} inj{inject} no_split{no_split}", folded=True)


@pytest.mark.parametrize("K,S,h,d", [(8, 256, 1, 40), (8, 1024, 1, 80), (8, 576, 2, 64), (5, 328, 1, 40), (7, 200, 2, 80),
                                     (8, 64, 1, 40)])
@pytest.mark.parametrize("inject", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_ext_attn_split_form(K, S, h, d, inject, dtype, monkeypatch):
    """Shapes of a head-sharded rank (one head group, all keyframes) in the STREAMING kernels (fused=False; such
    shapes run in the fused kernel by default, tests/test_fused_attn_gpu.py): the bank is split into runs of frames
    over extra workgroups and merged (attn_merge_kernel).  Against the oracle with the usual bound, and against the
    one-pass form of the same call within the output rounding."""
    ops = _ops()
    g = torch.Generator().manual_seed(K * 77 + S + d)
    D = h * d
    rnd = orc.bf16_round if dtype == torch.bfloat16 else (lambda x: x.half().float())
    q, k, v = (rnd(torch.randn(3 * K, S, D, generator=g)) for _ in range(3))
    refs = attn_ref(q, k, v, h, d ** -0.5, inject)
    dq, dk, dv = (t.to(dtype).cuda() for t in (q, k, v))
    monkeypatch.setattr(ops, "NO_SPLIT", False)
    out = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, fused=False)
    assert_attn_close(out, refs, f"split K{K} S{S} h{h} d{d} inj{inject}", dtype=dtype)
    monkeypatch.setattr(ops, "NO_SPLIT", True)
    one = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, fused=False)
    assert_attn_close(one, refs, f"one-pass K{K} S{S} h{h} d{d} inj{inject}", dtype=dtype)
    diff = (out.float() - one.float()).abs().cpu()     # both within the bound of the oracle; typically 0 or 1 ulp apart
    ref, ref_abs, sigma = refs
    assert bool((diff <= 2 * attn_bound(ref, ref_abs, dtype)).all())
    assert float(diff.mean()) < 2e-4
    assert torch.equal(out.view(3, -1)[0], one.view(3, -1)[0])      # the source branch is never split


def test_ext_attn_strided_qkv():
    """q/k/v as column slices of one fused [3K,S,3D] projection (token stride 3D)."""
    ops = _ops()
    K, S, h, d = 2, 128, 2, 40
    D = h * d
    g = torch.Generator().manual_seed(5)
    qkv = orc.bf16_round(torch.randn(3 * K, S, 3 * D, generator=g))
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    refs = attn_ref(q.contiguous(), k.contiguous(), v.contiguous(), h, d ** -0.5, True)
    dqkv = qkv.bfloat16().cuda()
    out = ops.ext_attn(dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:], h, d ** -0.5, True)
    assert_attn_close(out, refs, "strided")


@pytest.mark.parametrize("d", [64, 40])
def test_ext_attn_softmax_spike(d):
    """Forces large running-max jumps in late tiles (online-softmax rescale / shift path) and a
    near-one-hot softmax: one key per query made strongly aligned (scores up to ~3*|q|^2)."""
    ops = _ops()
    K, S, h = 2, 192, 1
    g = torch.Generator().manual_seed(11)
    q, k, v = (torch.randn(3 * K, S, d, generator=g) for _ in range(3))
    for b in range(3 * K):
        for s in range(0, S, 7):
            k[b, (s * 5 + 150) % S] = q[b, s] * 3.0     # spike lands in the last 64-key tile for many queries
    q, k, v = (orc.bf16_round(x) for x in (q, k, v))
    refs = attn_ref(q, k, v, h, d ** -0.5, False)
    for fused in (None, False):     # the fused small-problem kernel (default at this size) and the streaming kernels
        out = ops.ext_attn(q.bfloat16().cuda(), k.bfloat16().cuda(), v.bfloat16().cuda(), h, d ** -0.5, False, fused=fused)
        assert_attn_close(out, refs, f"spike d={d} fused={fused}")
    if d == 40:
        out_f = ops.ext_attn(q.bfloat16().cuda(), k.bfloat16().cuda(), v.bfloat16().cuda(), h, d ** -0.5, False,
                             fold_scale=True)
        assert_attn_close(out_f, refs, f"spike d={d} fold_scale", folded=True)


@pytest.mark.parametrize("gain", [3.0, 12.0])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("K,S,h,d", [(2, 576, 2, 64), (3, 1024, 1, 64), (2, 576, 2, 40), (2, 576, 1, 80)])
def test_ext_attn_d64_score_bound_shift_paths(K, S, h, d, dtype, gain):
    """The interleaved streaming kernels on peaked logits (S a multiple of 64, >= 512; d = 40 and 80 for the same kernel
    template with the denominator in a spare P.V row and, at 80, the lagged reference point instead of the bound).
    Head dim 64 in the interleaved streaming kernel (S a multiple of 64, >= 512; round 6): the score bound skips the
    per-tile maximum while |q| max|k| c - shift stays under its threshold; planted keys aligned with their queries make
    the scores climb (gain 3: inside the bf16 headroom, no rescale after the first tile; gain 12, and f16 at either
    gain: the bound fails, every half tile looks at its maximum and the deferred shift moves in late tiles -- O AND the
    matrix-pipe denominator are rescaled).  Plain and q/k-injected (the dual-V images of the interleaved kernel: 3 M-tiles at
    d = 40, 4 at d = 64, 5 at d = 80 -- the last one since the end of round 6), one-pass and split."""
    ops = _ops()
    g = torch.Generator().manual_seed(29 + S + d)
    q, k, v = (torch.randn(3 * K, S, h * d, generator=g) for _ in range(3))
    for b in range(3 * K):
        for s_ in range(0, S, 5):
            k[b, (s_ * 3 + S - 60) % S] = q[b, s_] * gain      # many spikes land in the last 64-key tile of a frame
    rnd = orc.bf16_round if dtype == torch.bfloat16 else (lambda x: x.half().float())
    q, k, v = (rnd(x) for x in (q, k, v))
    dq, dk, dv = (t.to(dtype).cuda() for t in (q, k, v))
    for inject in (False, True):
        refs = attn_ref(q, k, v, h, d ** -0.5, inject)
        for no_split in (True, False):
            out = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, fused=False, no_split=no_split)
            assert torch.isfinite(out.float()).all()
            assert_attn_close(out, refs, f"d64 shift paths K{K} S{S} {dtype} gain={gain} inject={inject} no_split={no_split}",
                              dtype=dtype)


@pytest.mark.parametrize("gain", [0.0, 3.0, 12.0])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("K,S,h", [(2, 576, 2), (3, 1024, 1), (2, 256, 2), (4, 320, 3)])
def test_ext_attn_mixed_mfma_shapes(K, S, h, dtype, gain):
    """The mixed-MFMA-shape form of the d = 40 interleaved kernel (QK^T 32x32x16, P re-laid out by v_permlane16_swap, P.V
    16x16x32 over three 16-row M-tiles; csrc/ext_attn.hip, IlScheduleMix).  By default only launches of >= 1024 workgroups
    take it (the full-size tests and bench.py's parity leg); TF_ATTN_HINT_MIX forces it here on small grids: N(0,1) and
    peaked logits (the deferred shift moves in late tiles: O rescaled through the row swap), plain and q/k-injected (the
    source launch in the mixed form beside the dual-V kernel), one-pass and split (the partial results' layout), bf16 and
    f16, against the oracle -- and against the non-mixed kernel within the same bound."""
    ops = _ops()
    from tokenflow_amd import _lib
    d = 40
    g = torch.Generator().manual_seed(41 + S + K)
    q, k, v = (torch.randn(3 * K, S, h * d, generator=g) for _ in range(3))
    if gain:
        for b in range(3 * K):
            for s_ in range(0, S, 5):
                k[b, (s_ * 3 + S - 60) % S] = q[b, s_] * gain
    rnd = orc.bf16_round if dtype == torch.bfloat16 else (lambda x: x.half().float())
    q, k, v = (rnd(x) for x in (q, k, v))
    dq, dk, dv = (t.to(dtype).cuda() for t in (q, k, v))
    for inject in (False, True):
        refs = attn_ref(q, k, v, h, d ** -0.5, inject)
        for no_split in (True, False):
            out = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, fused=False, no_split=no_split, hints=_lib.TF_ATTN_HINT_MIX)
            assert torch.isfinite(out.float()).all()
            assert_attn_close(out, refs, f"mixed shapes K{K} S{S} h{h} {dtype} gain={gain} inject={inject} no_split={no_split}",
                              dtype=dtype)
    # fp32 output (the normalised accumulator): both forms within north_star's 1e-3 of the oracle where the streaming kernels are
    # what the library takes (frames of <= 256 tokens run in the fused kernel with P as hi + lo: the 16-bit P of a streaming
    # kernel averages too few keys there, 1.2e-3 at S = 256 for either form)
    if gain == 0.0 and S >= 576:
        refs = attn_ref(q, k, v, h, d ** -0.5, False)
        for hints in (0, _lib.TF_ATTN_HINT_MIX):
            o32 = ops.ext_attn(dq, dk, dv, h, d ** -0.5, False, fused=False, out_dtype=torch.float32, hints=hints)
            assert float((o32.cpu() - refs[0]).abs().max()) < 1e-3


@pytest.mark.parametrize("gain", [3.0, 12.0])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d", [40, 80])
def test_ext_attn_folded_shift_paths(d, dtype, gain):
    """The Dh = 40 kernels (fp32 scaling = default, and the opt-in folded scale; Dh = 80 is the plain online softmax,
    run for contrast) skip the per-tile maximum while |q| max|k| c - shift stays under a threshold (2^60 headroom
    in bf16, 2^14 in f16).  gain = 3: the bound holds in bf16, the scores climb
    ~25 binades above the first tile's shift without any rescale; gain = 12: the bound fails, so the
    kernel looks at every tile's maximum and moves the shift in late tiles.  Both in both dtypes, on a
    2-frame bank with a ragged last tile."""
    ops = _ops()
    K, S, h = 2, 200, 2
    g = torch.Generator().manual_seed(17)
    q, k, v = (torch.randn(3 * K, S, h * d, generator=g) for _ in range(3))
    for b in range(3 * K):
        for s in range(0, S, 5):
            k[b, (s * 3 + 140) % S] = q[b, s] * gain
    rnd = orc.bf16_round if dtype == torch.bfloat16 else (lambda x: x.half().float())
    q, k, v = (rnd(x) for x in (q, k, v))
    for inject in (False, True):
        refs = attn_ref(q, k, v, h, d ** -0.5, inject)
        for fold in ((False, True) if d == 40 else (False,)):
            out = ops.ext_attn(q.to(dtype).cuda(), k.to(dtype).cuda(), v.to(dtype).cuda(), h, d ** -0.5, inject,
                               fold_scale=fold, fused=False)     # the score bound is a streaming-kernel feature
            assert torch.isfinite(out.float()).all()
            assert_attn_close(out, refs, f"shift paths d={d} {dtype} gain={gain} inject={inject} fold={fold}",
                              dtype=dtype, folded=fold)


@pytest.mark.parametrize("K,S,h,d", [(3, 320, 2, 40), (2, 136, 2, 40), (2, 520, 2, 64), (2, 264, 1, 80), (2, 72, 1, 160)])
@pytest.mark.parametrize("inject", [False, True])
@pytest.mark.parametrize("fused", [None, False])
def test_ext_attn_bank_and_source_parts(K, S, h, d, inject, fused):
    """TF_ATTN_BANK_ONLY + TF_ATTN_SOURCE_ONLY together reproduce the full call bit for bit, write nothing
    outside their branches and never read the slabs they do not need (poisoned with NaN here)."""
    ops = _ops()
    D = h * d
    g = torch.Generator(device="cuda").manual_seed(23)
    q, k, v = (torch.randn(3 * K, S, D, generator=g, device="cuda").bfloat16() for _ in range(3))
    full = ops.ext_attn(q, k, v, h, d ** -0.5, inject, fused=fused)
    nan = float("nan")

    def poisoned(t, branches):
        t = t.clone()
        t.view(3, -1)[branches] = nan
        return t
    qk_unread = [1, 2] if inject else [0]
    out = torch.full_like(full, 7.0)
    ops.ext_attn(poisoned(q, qk_unread), poisoned(k, qk_unread), poisoned(v, [0]), h, d ** -0.5, inject, out=out,
                 part="bank", fused=fused)
    assert torch.equal(out.view(3, -1)[1:], full.view(3, -1)[1:])
    assert bool((out.view(3, -1)[0] == 7.0).all())
    out = torch.full_like(full, 7.0)
    ops.ext_attn(poisoned(q, [1, 2]), poisoned(k, [1, 2]), poisoned(v, [1, 2]), h, d ** -0.5, inject, out=out,
                 part="source", fused=fused)
    assert torch.equal(out.view(3, -1)[0], full.view(3, -1)[0])
    assert bool((out.view(3, -1)[1:] == 7.0).all())


def test_ext_attn_argument_errors():
    ops = _ops()
    from tokenflow_amd._lib import TokenflowHipError
    x = torch.zeros(3, 64, 96, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(TokenflowHipError, match="head dim"):
        ops.ext_attn(x, x, x, 2, 1.0, False)             # d = 48 unsupported
    with pytest.raises(TypeError):
        ops.ext_attn(x.float(), x.float(), x.float(), 2, 1.0, False)
    with pytest.raises(TokenflowHipError, match="CPU tensor"):
        ops.ext_attn(x.cpu(), x.cpu(), x.cpu(), 2, 1.0, False)


# --------------------------------------------------------------------------- NN search
def _nn_check(tgt, piv_all, bi, what):
    """tgt [n,S,D] fp32 (bf16 values), piv_all [K,S,D] fp32 (bf16 values)."""
    ops = _ops()
    ref_idx, sim = orc.nn_search(tgt, piv_all, bi)
    ids = orc.keyframe_ids(bi)
    dp = piv_all.bfloat16().cuda()
    inv = ops.pivot_inv_norm(dp)
    ref_inv = 1.0 / piv_all.norm(dim=-1)
    assert torch.allclose(inv.cpu(), ref_inv, rtol=1e-5), what
    got = ops.nn_search(tgt.reshape(-1, tgt.shape[-1]).bfloat16().cuda(), dp, inv, ids).cpu()
    assert got.dtype == torch.int32 and got.shape == (len(ids), tgt.shape[0] * tgt.shape[1])
    S = piv_all.shape[1]
    n_diff = n_bad = 0
    for p, (r, s) in enumerate(zip(ref_idx, sim.chunk(len(ids), dim=1))):
        assert int(got[p].min()) >= 0 and int(got[p].max()) < S
        a, b = orc.nn_mismatch_tie_aware(s, r, got[p], NN_TAU)
        n_diff += a
        n_bad += b
    assert n_bad == 0, f"{what}: {n_bad} rows differ beyond a near-tie ({n_diff} differ at all)"
    return n_diff


@pytest.mark.parametrize("name", list(gc.PROP_CASES))
def test_nn_search_golden_cases(name, golden_prop):
    K, n, S, D, dt = gc.PROP_CASES[name]
    piv, kf_out, hidden = gc.prop_inputs(name)
    for bi in range(K):
        tgt = hidden[bi].float().view(3, n, S, D)[0]
        n_diff = _nn_check(tgt, piv[0], bi, f"{name}/chunk{bi}")
        if "videolike" in name:
            assert n_diff == 0           # far from ties: must equal the reference's indices exactly
            ops = _ops()
            dp = piv[0].bfloat16().cuda()
            got = ops.nn_search(tgt.reshape(-1, D).bfloat16().cuda(), dp, ops.pivot_inv_norm(dp),
                                orc.keyframe_ids(bi)).cpu()
            for p, gi in enumerate(golden_prop[name]["chunks"][bi]["idx"]):
                assert torch.equal(got[p], gi.int())


@pytest.mark.parametrize("K,n,S,D", [(2, 3, 200, 320), (3, 2, 64, 1280), (2, 5, 16, 1280), (2, 1, 520, 640),
                                     (2, 9, 128, 72), (2, 2, 200, 1280), (2, 1, 72, 1096), (3, 4, 256, 1280)])
def test_nn_search_shapes(K, n, S, D):
    """S not a multiple of the 128-pivot tile, target count not a multiple of the panel,
    D not a multiple of the 64-wide chunk, deep D."""
    g = torch.Generator().manual_seed(S + D)
    ln = torch.nn.LayerNorm(D, elementwise_affine=False)
    piv = orc.bf16_round(ln(torch.randn(K, S, D, generator=g)))
    tgt = orc.bf16_round(ln(torch.randn(n, S, D, generator=g)))
    for bi in range(K):
        _nn_check(tgt, piv, bi, f"S{S} D{D} chunk{bi}")


def test_nn_search_f16():
    """f16 inputs (the reference's own autocast dtype) take the f16 MFMA path."""
    ops = _ops()
    K, n, S, D = 2, 3, 160, 320
    g = torch.Generator().manual_seed(21)
    ln = torch.nn.LayerNorm(D, elementwise_affine=False)
    piv = ln(torch.randn(K, S, D, generator=g)).half().float()
    tgt = (piv[1][torch.randperm(S, generator=g)][None].repeat(n, 1, 1) + 0.1 * torch.randn(n, S, D, generator=g)).half().float()
    ref_idx, sim = orc.nn_search(tgt, piv, 1)
    dp = piv.half().cuda()
    got = ops.nn_search(tgt.reshape(-1, D).half().cuda(), dp, ops.pivot_inv_norm(dp), [1, 0]).cpu()
    for p_, (r, s_) in enumerate(zip(ref_idx, sim.chunk(2, dim=1))):
        _, bad = orc.nn_mismatch_tie_aware(s_, r, got[p_], NN_TAU)
        assert bad == 0


def test_nn_search_argument_errors():
    ops = _ops()
    from tokenflow_amd._lib import TokenflowHipError
    piv = torch.zeros(2, 16, 36, dtype=torch.bfloat16, device="cuda")       # D = 36 is not a multiple of 8
    with pytest.raises(TokenflowHipError, match="tf_nn_search"):
        ops.nn_search(piv[0], piv, torch.ones(2, 16, device="cuda"), [0])
    with pytest.raises(ValueError):
        ops.nn_search(piv[0], piv, torch.ones(2, 16, device="cuda"), [2])   # keyframe index out of range


def test_nn_search_exact_ties_first_index():
    """Duplicate pivot rows give bit-identical scores: the FIRST index must win (torch.argmax)."""
    ops = _ops()
    S, D = 300, 128
    g = torch.Generator().manual_seed(3)
    base = orc.bf16_round(torch.randn(S, D, generator=g))
    piv = base.clone()
    piv[200:300] = base[0:100]            # rows 200.. duplicate rows 0..
    piv[150] = base[10]
    tgt = base[[5, 10, 99, 120, 160]] + 0.0
    dp = piv[None].bfloat16().cuda()
    got = ops.nn_search(tgt.bfloat16().cuda(), dp, ops.pivot_inv_norm(dp), [0]).cpu()[0]
    assert got.tolist() == [5, 10, 99, 120, 160]


# --------------------------------------------------------------------------- gather / blend
@pytest.mark.parametrize("name", list(gc.PROP_CASES))
def test_gather_blend_bit_exact_vs_golden(name, golden_prop):
    ops = _ops()
    K, n, S, D, dt = gc.PROP_CASES[name]
    piv, kf_out, hidden = gc.prop_inputs(name)
    w = orc.blend_weights(n, 1).cuda()
    for bi in range(K):
        ids = orc.keyframe_ids(bi)
        tgt = hidden[bi].float().view(3, n, S, D)[0]
        idx, _ = orc.nn_search(tgt, piv[0], bi)
        ref = orc.gather_blend(kf_out, idx, bi, n, residual=hidden[bi])
        didx = torch.stack(idx).int().cuda()
        out = ops.gather_blend(kf_out.cuda(), didx, w if len(ids) == 2 else None, ids, n, hidden[bi].cuda(), ref.dtype)
        assert out.dtype == ref.dtype
        assert torch.equal(out.cpu(), ref), f"{name}/chunk{bi}: not bit-exact"
        check(out, golden_prop[name]["chunks"][bi]["out"], 0.0, f"{name}/chunk{bi}/golden")


@pytest.mark.parametrize("in_dt,res_dt,out_dt", [
    (torch.bfloat16, torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32, torch.float32),
    (torch.float16, torch.float16, torch.float32), (torch.float32, None, torch.float32),
    (torch.bfloat16, None, torch.bfloat16), (torch.float16, torch.float16, torch.float16)])
@pytest.mark.parametrize("P", [1, 2])
def test_gather_blend_dtypes(in_dt, res_dt, out_dt, P):
    ops = _ops()
    K, n, S, D = 3, 4, 40, 160
    g = torch.Generator().manual_seed(7)
    kf_out = torch.randn(3 * K, S, D, generator=g).to(in_dt)
    res = torch.randn(3 * n, S, D, generator=g).to(res_dt) if res_dt is not None else None
    bi = 2 if P == 2 else 0
    idx = [torch.randint(0, S, (n * S,), generator=g) for _ in range(P)]
    ref = orc.gather_blend(kf_out.float(), idx, bi, n, residual=res.float() if res is not None else None)
    ref = ref.to(out_dt)       # single rounding of the fp32 result, as the kernel does
    out = ops.gather_blend(kf_out.cuda(), torch.stack(idx).int().cuda(),
                           orc.blend_weights(n, 1).cuda() if P == 2 else None, orc.keyframe_ids(bi), n,
                           res.cuda() if res is not None else None, out_dt)
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("K,n,S,D", [(3, 5, 4096, 320), (3, 2, 1024, 640), (2, 3, 256, 1280), (2, 2, 64, 1280),
                                     (2, 3, 200, 72)])
@pytest.mark.parametrize("P", [1, 2])
def test_propagate_equals_search_then_gather(K, n, S, D, P):
    """tf_nn_gather_blend (the gather merges the search's per-split candidates itself) must equal
    tf_nn_search + tf_gather_blend bit for bit -- with the pivot range split (large S) and not."""
    ops = _ops()
    g = torch.Generator(device="cuda").manual_seed(31)
    piv = torch.nn.functional.layer_norm(torch.randn(K, S, D, generator=g, device="cuda"), (D,)).bfloat16()
    tgt = torch.nn.functional.layer_norm(torch.randn(n * S, D, generator=g, device="cuda"), (D,)).bfloat16()
    kf_out = torch.randn(3 * K, S, D, generator=g, device="cuda").bfloat16()
    res = torch.randn(3 * n, S, D, generator=g, device="cuda").bfloat16()
    inv = ops.pivot_inv_norm(piv)
    ids = [K - 1, K - 2][:P]
    w = orc.blend_weights(n, 1).cuda() if P == 2 else None
    out_dtype = torch.float32 if P == 2 else torch.bfloat16
    idx = ops.nn_search(tgt, piv, inv, ids)
    two = ops.gather_blend(kf_out, idx, w, ids, n, res, out_dtype)
    one = ops.propagate(tgt, piv, inv, ids, kf_out, w, n, res, out_dtype)
    assert one.dtype == two.dtype and torch.equal(one, two)


def test_inject_copy_exact():
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    for dt in (torch.float16, torch.float32, torch.bfloat16):
        x = torch.randn(6, 32, 4, 4, generator=g).to(dt)
        ref = orc.conv_inject_(x.clone())
        out = ops.inject_copy_(x.cuda())
        assert torch.equal(out.cpu(), ref)


# --------------------------------------------------------------------------- layer norm (row f2)


@pytest.mark.parametrize("rows,D", [(4096, 320), (515, 640), (37, 1280), (5, 72), (3, 2048)])
@pytest.mark.parametrize("in_dt,w_dt,out_dt", [
    (torch.bfloat16, torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32, torch.bfloat16),
    (torch.float32, torch.float32, torch.bfloat16), (torch.float16, torch.float16, torch.float16),
    (torch.float32, None, torch.float32), (torch.bfloat16, torch.float32, torch.float32)])
def test_layer_norm_vs_torch_fp32(rows, D, in_dt, w_dt, out_dt):
    """tf_layer_norm against torch's fp32 layer_norm of the same (already rounded) input: fp32 output within
    2e-6 (relative to max(1,|ref|)), 16-bit output within one rounding of the fp32 result; the inverse norms
    are those of the ROUNDED output rows (what tf_pivot_inv_norm computes from the stored pivots)."""
    ops = _ops()
    g = torch.Generator().manual_seed(rows * 7 + D)
    x = (torch.randn(rows, D, generator=g) * 3 + 0.5).to(in_dt)
    w = (1 + 0.2 * torch.randn(D, generator=g)).to(w_dt) if w_dt is not None else None
    b = (0.1 * torch.randn(D, generator=g)).to(w_dt) if w_dt is not None else None
    ref = torch.nn.functional.layer_norm(x.float(), (D,), None if w is None else w.float(),
                                         None if b is None else b.float(), 1e-5)
    out, inv = ops.layer_norm(x.cuda(), None if w is None else w.cuda(), None if b is None else b.cuda(), 1e-5,
                              out_dt, want_inv_norm=True)
    assert out.dtype == out_dt and out.shape == x.shape and inv.shape == (rows,)
    got = out.float().cpu()
    eps = {torch.float32: 2e-6, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[out_dt]
    bound = eps * ref.abs().clamp(min=1.0) + (4e-6 if out_dt != torch.float32 else 0.0)
    assert bool(((got - ref).abs() <= bound).all()), float(((got - ref).abs() - bound).max())
    inv_ref = 1.0 / got.norm(dim=-1)
    assert torch.allclose(inv.cpu(), inv_ref, rtol=2e-6, atol=0)
    same = ops.pivot_inv_norm(out) if out_dt != torch.float32 else None
    if same is not None:          # the producer's side output and the stand-alone kernel agree
        assert torch.allclose(inv, same, rtol=2e-6, atol=0)


def test_layer_norm_into_caller_buffers():
    """out= / inv_out=: the kernel writes into the caller's (row-contiguous) views -- the in-place sharded hook pass
    hands it slices of the halo-extended block state -- and returns those views; same bits as the allocating call."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6 * 64, 320, generator=g).bfloat16().cuda()
    w, b = (torch.randn(320, generator=g).cuda() for _ in range(2))
    want, want_inv = ops.layer_norm(x, w, b, 1e-5, torch.bfloat16, want_inv_norm=True)
    state = torch.zeros(8 * 64, 320, dtype=torch.bfloat16, device="cuda")
    inv_state = torch.zeros(8 * 64, dtype=torch.float32, device="cuda")
    out, inv = ops.layer_norm(x, w, b, 1e-5, torch.bfloat16, want_inv_norm=True, out=state[64:7 * 64],
                              inv_out=inv_state[64:7 * 64])
    assert out.data_ptr() == state[64:].data_ptr() and inv.data_ptr() == inv_state[64:].data_ptr()
    assert torch.equal(state[64:7 * 64], want) and torch.equal(inv_state[64:7 * 64], want_inv)
    assert not bool(state[:64].any()) and not bool(state[7 * 64:].any())       # nothing outside the view
    with pytest.raises(ValueError):
        ops.layer_norm(x, w, b, 1e-5, torch.bfloat16, out=state[:64])           # wrong shape


def test_layer_norm_argument_errors():
    ops = _ops()
    from tokenflow_amd._lib import TokenflowHipError
    x = torch.randn(4, 2056, device="cuda").bfloat16()
    with pytest.raises(TokenflowHipError):
        ops.layer_norm(x, None, None, 1e-5, torch.bfloat16)          # D > 2048
    with pytest.raises(TokenflowHipError):
        ops.layer_norm(x[:, :12], None, None, 1e-5, torch.bfloat16)  # D % 8


@pytest.mark.parametrize("rows,D", [(4096, 320), (515, 640), (37, 1280), (5, 72)])
@pytest.mark.parametrize("a_dt,b_dt,out_dt", [
    (torch.bfloat16, torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32, torch.bfloat16),
    (torch.float32, torch.bfloat16, torch.bfloat16), (torch.float16, torch.float16, torch.float16),
    (torch.float32, torch.float32, torch.float32)])
def test_add_layer_norm_equals_add_then_norm(rows, D, a_dt, b_dt, out_dt):
    """tf_add_layer_norm == torch's `a + b` (promoted dtype, one rounding) followed by tf_layer_norm, bit for bit."""
    ops = _ops()
    g = torch.Generator().manual_seed(rows + D)
    a = (torch.randn(rows, D, generator=g) * 2).to(a_dt).cuda()
    b = (torch.randn(rows, D, generator=g) + 0.3).to(b_dt).cuda()
    w = (1 + 0.2 * torch.randn(D, generator=g)).cuda()
    bias = (0.1 * torch.randn(D, generator=g)).cuda()
    total, out = ops.add_layer_norm(a, b, w, bias, 1e-5, out_dt)
    want_total = a + b
    assert total.dtype == want_total.dtype and torch.equal(total, want_total)
    assert torch.equal(out, ops.layer_norm(want_total, w, bias, 1e-5, out_dt)[0])
