"""Parity of the fused small-problem attention kernel (csrc/ext_attn_fused.hip) through the C ABI, against the CPU
oracle (tokenflow_utils.py:124-197 restated) and against its own alternative forms.  Needs an MI355X.

Tolerances:
  * default (P in one 16-bit value): the attention bound of tests/test_kernels_gpu.py,
        |out - ref| <= 2e-4 + eps |ref| + eps softmax.|V|,   eps = 2^-8 (bf16) / 2^-11 (f16);
  * precise P (bf16, hi + lo; the default at S <= 256) with fp32 output: the rounding of P and of the output are both
    gone -- what is left is fp32 accumulation order and v_exp_f32:  |out - ref| <= 2e-5 + 2^-16 softmax.|V| ;
    with bf16 output: that plus the output's own rounding (half an ulp, 2^-9 relative);
  * alternative forms of the SAME arithmetic (wave-private or shared-tile staging, 1 or 2 query waves per workgroup;
    bank + source parts instead of one call): bit-identical.
"""
import pytest
import torch

from oracle import tokenflow_oracle as orc
from tests.test_kernels_gpu import assert_attn_close, attn_ref

pytestmark = pytest.mark.gpu


def _ops():
    from tokenflow_amd import _lib, ops
    return ops, _lib


def _inputs(K, S, h, d, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    rnd = orc.bf16_round if dtype == torch.bfloat16 else (lambda x: x.half().float())
    return tuple(rnd(torch.randn(3 * K, S, h * d, generator=g)) for _ in range(3))


SHAPES = [(2, 256, 2, 40), (3, 136, 2, 64), (2, 200, 1, 80), (1, 16, 2, 160), (4, 64, 8, 40), (2, 45, 2, 160),
          (8, 256, 1, 160), (8, 64, 2, 160), (1, 1, 1, 40), (5, 96, 2, 64), (4, 16, 8, 160), (3, 33, 1, 80)]


@pytest.mark.parametrize("K,S,h,d", SHAPES)
@pytest.mark.parametrize("geom", [(1, 4), (1, 8), (2, 4), (4, 2), (4, 1)])
@pytest.mark.parametrize("inject", [False, True])
def test_fused_geometries_vs_oracle(K, S, h, d, geom, inject):
    """Every built geometry (query waves x key groups) on ragged and tiny shapes: 1-token frames, S below one sub-tile,
    S not a multiple of 32, more key groups than sub-tiles (idle groups in the merge), many heads."""
    ops, lib = _ops()
    q, k, v = _inputs(K, S, h, d, torch.bfloat16, K * 131 + S + d)
    refs = attn_ref(q, k, v, h, d ** -0.5, inject, need_sigma=False)
    dq, dk, dv = (t.bfloat16().cuda() for t in (q, k, v))
    for prec in (lib.TF_ATTN_NO_PRECISE_P, lib.TF_ATTN_PRECISE_P):
        out = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, fused=True, hints=lib.attn_hint(*geom) | prec)
        assert_attn_close(out, refs, f"fused K{K} S{S} h{h} d{d} geom{geom} inj{inject} prec{prec != lib.TF_ATTN_NO_PRECISE_P}")
        if geom == (1, 4) and d <= 80:     # two query blocks per wave: the same bits (the arithmetic of a query is unchanged)
            two = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, fused=True, hints=lib.attn_hint(1, 4, qb=2) | prec)
            assert torch.equal(two, out), f"two query blocks per wave differ: K{K} S{S} h{h} d{d}"


@pytest.mark.parametrize("K,S,h,d", [(8, 1024, 1, 80), (4, 1024, 2, 40), (2, 576, 2, 64), (3, 320, 1, 160)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_long_sequences(K, S, h, d, dtype):
    """A sharded rank's level-1 shape (one head group, all keyframes, 8192-key bank) and BASELINE config 1's level 0:
    hundreds of sub-tiles per key group, both injection states, both dtypes; automatic geometry."""
    ops, lib = _ops()
    q, k, v = _inputs(K, S, h, d, dtype, 7 * K + S + d)
    dq, dk, dv = (t.to(dtype).cuda() for t in (q, k, v))
    for inject in (False, True):
        refs = attn_ref(q, k, v, h, d ** -0.5, inject, need_sigma=False)
        out = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, fused=True)
        assert out.dtype == dtype
        assert_attn_close(out, refs, f"fused long K{K} S{S} h{h} d{d} {dtype} inj{inject}", dtype=dtype)
        if d <= 80:
            two = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, fused=True, hints=lib.attn_hint(1, 4, qb=2))
            assert_attn_close(two, refs, f"fused long, two query blocks per wave K{K} S{S} h{h} d{d} {dtype}", dtype=dtype)
            assert torch.equal(two, ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, fused=True, hints=lib.attn_hint(1, 4)))


@pytest.mark.parametrize("K,S,h,d", [(8, 64, 8, 160), (8, 256, 2, 160), (4, 16, 8, 160), (4, 64, 8, 160), (4, 256, 4, 80),
                                     (2, 45, 2, 160)])
@pytest.mark.parametrize("inject", [False, True])
def test_fused_precise_p_meets_1e3_absolute(K, S, h, d, inject):
    """The coarse levels of BASELINE configs 1 and 2 (64 / 16-token frames: the source branch averages a handful of
    N(0,1) values, |out| ~ 0.6): with P carried as hi + lo the fp32 output is within 3e-5 + 2^-15 softmax.|V| of the
    oracle -- far inside north_star's absolute 1e-3 -- and the bf16 output within its own half-ulp of that.  The
    single-value form on the same inputs is measurably worse (the path under test is active)."""
    ops, lib = _ops()
    q, k, v = _inputs(K, S, h, d, torch.bfloat16, 1000 + K + S)
    ref, ref_abs, _ = attn_ref(q, k, v, h, d ** -0.5, inject, need_sigma=False)
    dq, dk, dv = (t.bfloat16().cuda() for t in (q, k, v))
    out32 = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, out_dtype=torch.float32).cpu()     # default: precise at S <= 256
    err32 = (out32 - ref).abs()
    assert float((err32 - (3e-5 + 2.0 ** -15 * ref_abs)).max()) <= 0, f"fp32-out error {float(err32.max()):.3e}"
    assert float(err32.max()) < 1e-3
    out16 = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject).float().cpu()
    err16 = (out16 - ref).abs()
    half_ulp = torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 8)
    assert float((err16 - (half_ulp * 1.01 + 3e-5 + 2.0 ** -15 * ref_abs)).max()) <= 0
    assert torch.equal(out16, out32.bfloat16().float())          # the bf16 output is the rounded fp32 one
    plain32 = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, out_dtype=torch.float32,
                           hints=lib.TF_ATTN_NO_PRECISE_P).cpu()
    assert float((plain32 - ref).abs().max()) > 4 * float(err32.max())


@pytest.mark.parametrize("K,S,h,d", [(8, 64, 8, 160), (4, 16, 8, 160), (8, 256, 2, 160), (4, 256, 4, 80)])
def test_fused_f16_meets_1e3_absolute(K, S, h, d):
    """f16 -- the reference's own autocast dtype (run_tokenflow_pnp.py:220): P in f16 has 11 significand bits, the fp32
    output stays under 1e-3 in absolute terms at the coarse levels without the hi + lo form."""
    ops, lib = _ops()
    q, k, v = _inputs(K, S, h, d, torch.float16, 2000 + K + S)
    for inject in (False, True):
        ref, ref_abs, _ = attn_ref(q, k, v, h, d ** -0.5, inject, need_sigma=False)
        out32 = ops.ext_attn(q.half().cuda(), k.half().cuda(), v.half().cuda(), h, d ** -0.5, inject,
                             out_dtype=torch.float32).cpu()
        err = (out32 - ref).abs()
        assert float((err - (2e-5 + 2.0 ** -11 * ref_abs)).max()) <= 0
        assert float(err.max()) < 1e-3


@pytest.mark.parametrize("K,S,h,d", [(2, 256, 2, 40), (3, 136, 2, 64), (8, 256, 1, 160), (8, 1024, 1, 80), (4, 64, 8, 160)])
def test_fused_arithmetic_independent_of_query_waves_and_parts(K, S, h, d):
    """The arithmetic of a (query, head) depends on the number of key groups only: the wave-private form (one query wave,
    K fragments straight from global memory, no barriers), the shared-tile form with 2 query waves per workgroup, and
    the bank-only + source-only parts of a sharded rank all reproduce the same bits (what keeps a rank's one-pass
    result identical to the single-GPU one)."""
    ops, lib = _ops()
    q, k, v = _inputs(K, S, h, d, torch.bfloat16, 4000 + K + S + d)
    dq, dk, dv = (t.bfloat16().cuda() for t in (q, k, v))
    for inject in (False, True):
        one = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, fused=True, hints=lib.attn_hint(1, 4))
        two = ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, fused=True, hints=lib.attn_hint(2, 4))
        assert torch.equal(one, two)
        parts = torch.full_like(one, 7.0)
        ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, fused=True, hints=lib.attn_hint(1, 4), out=parts, part="bank")
        assert bool((parts.view(3, -1)[0] == 7.0).all())
        ops.ext_attn(dq, dk, dv, h, d ** -0.5, inject, fused=True, hints=lib.attn_hint(2, 4), out=parts, part="source")
        assert torch.equal(parts, one)


@pytest.mark.parametrize("d", [40, 64, 160])
@pytest.mark.parametrize("gain", [3.0, 12.0])
def test_fused_softmax_spikes_across_key_groups(d, gain):
    """One key per query made strongly aligned, placed so that the maxima of the four key groups of a query differ by
    tens of binades and arrive in late sub-tiles: exercises the rescale inside a group and the exp2((m_k - M) c)
    weights of the LDS merge (an idle or far-below group must contribute exactly its share, never NaN)."""
    ops, lib = _ops()
    K, S, h = 2, 200, 2
    g = torch.Generator().manual_seed(17 + d)
    q, k, v = (torch.randn(3 * K, S, h * d, generator=g) for _ in range(3))
    for b in range(3 * K):
        for s in range(0, S, 5):
            k[b, (s * 3 + 140) % S] = q[b, s] * gain
    q, k, v = (orc.bf16_round(x) for x in (q, k, v))
    for inject in (False, True):
        refs = attn_ref(q, k, v, h, d ** -0.5, inject, need_sigma=False)
        for geom in ((1, 4), (1, 8), (4, 2)):
            out = ops.ext_attn(q.bfloat16().cuda(), k.bfloat16().cuda(), v.bfloat16().cuda(), h, d ** -0.5, inject,
                               fused=True, hints=lib.attn_hint(*geom))
            assert torch.isfinite(out.float()).all()
            assert_attn_close(out, refs, f"fused spikes d={d} gain={gain} inject={inject} geom={geom}")


def test_fused_strided_fused_projection_and_views():
    """q / k / v as column slabs of one fused [3K, S, 3D] projection (token stride 3D) and through the strided entry
    point on 4-D views: same bits as the dense call."""
    ops, lib = _ops()
    K, S, h, d = 3, 72, 2, 160
    D = h * d
    g = torch.Generator(device="cuda").manual_seed(5)
    qkv = torch.randn(3 * K, S, 3 * D, generator=g, device="cuda").bfloat16()
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    dense = ops.ext_attn(q.contiguous(), k.contiguous(), v.contiguous(), h, d ** -0.5, True)
    assert torch.equal(ops.ext_attn(q, k, v, h, d ** -0.5, True), dense)
    out = torch.empty(3, K, S, D, dtype=torch.bfloat16, device="cuda")
    ops.ext_attn_views(q.view(3, K, S, D), k.view(3, K, S, D), v.view(3, K, S, D), out, h, d ** -0.5, True)
    assert torch.equal(out.view(3 * K, S, D), dense)
    refs = attn_ref(q.float().cpu().contiguous(), k.float().cpu().contiguous(), v.float().cpu().contiguous(), h,
                    d ** -0.5, True, need_sigma=False)
    assert_attn_close(dense, refs, "fused strided")


def test_fused_selection_rules():
    """Which calls take the fused kernel: S <= 256 whatever the grid and the split mode (a SHAPE rule: a sharded rank
    and the single GPU agree, so one-pass results stay bit-identical); larger frames only on small grids and only
    when splitting is allowed; TF_ATTN_NO_FUSED never.  Observed through the hint bits: a geometry hint changes the
    result's low bits only when the fused kernel runs."""
    ops, lib = _ops()
    g = torch.Generator(device="cuda").manual_seed(9)

    def differs(K, S, h, d, **kw):
        q, k, v = (torch.randn(3 * K, S, h * d, generator=g, device="cuda").bfloat16() for _ in range(3))
        a = ops.ext_attn(q, k, v, h, d ** -0.5, False, hints=lib.attn_hint(4, 1), **kw)     # one key group
        b = ops.ext_attn(q, k, v, h, d ** -0.5, False, hints=lib.attn_hint(1, 4), **kw)     # four: other summation order
        return not torch.equal(a, b)
    assert differs(8, 256, 2, 160)                       # shape rule
    assert differs(8, 256, 2, 160, no_split=True)        # ... also in one-pass mode
    assert not differs(8, 256, 2, 160, fused=False)      # TF_ATTN_NO_FUSED
    assert differs(8, 1024, 1, 80)                       # small grid (a rank's level 1)
    assert not differs(8, 1024, 1, 80, no_split=True)    # grid rule is off in one-pass mode
    assert not differs(8, 1024, 8, 80)                   # the single GPU's level 1 keeps the streaming kernel
