// Token gather + two-keyframe blend + residual add, and the PnP feature-injection copy.
// Replaces tokenflow_utils.py:362-397 and 87-91 of omerbt/TokenFlow.
//
// HBM-bound: each output row (D elements, 640 B .. 2.5 KB) is a contiguous row of one
// keyframe's cached attention output, so a row gather is naturally coalesced.  One
// thread owns one 8-element piece of one token and loops the 3 branches, re-using the
// token's two int32 indices and the frame's weight; 16-byte loads/stores; fp32 math
// with the reference's operation order and NO fma contraction, so fp32 results are
// bit-identical to torch's  w1*a1 + (1-w1)*a2  (+ hidden_states).
#include "ln_row.h"
#include "tf_common.h"

// bit-exactness vs torch needs separately rounded multiplies and adds: no fma contraction here
#pragma clang fp contract(off)

namespace {

template <typename T>
struct Vec8 {  // 8 elements of T as raw 16-byte words
    static constexpr int WORDS = sizeof(T) * 8 / 16;
    u32x4 w[WORDS];
};

template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&f)[8]) {
    if constexpr (sizeof(T) == 4) {
        const u32x4 a = ld16(p), b = ld16(p + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[i] = __uint_as_float(a[i]);
            f[4 + i] = __uint_as_float(b[i]);
        }
    } else {
        typedef T v8 __attribute__((ext_vector_type(8)));
        const v8 v = __builtin_bit_cast(v8, ld16(p));
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
    }
}

template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&f)[8]) {
    if constexpr (sizeof(T) == 4) {
        u32x4 a, b;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = __float_as_uint(f[i]);
            b[i] = __float_as_uint(f[4 + i]);
        }
        st16(p, a);
        st16(p + 4, b);
    } else {
        typedef T v8 __attribute__((ext_vector_type(8)));
        v8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (T)f[i];  // round-to-nearest-even
        st16(p, __builtin_bit_cast(u32x4, v));
    }
}

// Streaming accesses of the kernel (residual in, result out) are touched exactly once: nontemporal, so that they
// do not push the keyframe rows -- every one of them gathered ~2 n times per block -- out of the L2
// (-2 % at cfg2 levels 0 and 1, profiles/r02_gather_order_ab.txt).
template <typename T>
__device__ __forceinline__ void load8_stream(const T* p, float (&f)[8]) {
    if constexpr (sizeof(T) == 4) {
        const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        const u32x4 b = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[i] = __uint_as_float(a[i]);
            f[4 + i] = __uint_as_float(b[i]);
        }
    } else {
        typedef T v8 __attribute__((ext_vector_type(8)));
        const v8 v = __builtin_bit_cast(v8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)));
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
    }
}

template <typename T>
__device__ __forceinline__ void store8_stream(T* p, const float (&f)[8]) {
    if constexpr (sizeof(T) == 4) {
        u32x4 a, b;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = __float_as_uint(f[i]);
            b[i] = __float_as_uint(f[4 + i]);
        }
        __builtin_nontemporal_store(a, reinterpret_cast<u32x4*>(p));
        __builtin_nontemporal_store(b, reinterpret_cast<u32x4*>(p + 4));
    } else {
        typedef T v8 __attribute__((ext_vector_type(8)));
        v8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (T)f[i];  // round-to-nearest-even
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(p));
    }
}

struct NoRes {};

// Multi-chunk call (tf_nn_gather_blend_chunks): `n` frames are C chunks of nc frames; chunk j = frame / nc gathers
// from keyframe slots kf0 + j and kf1 + j.  When the first chunk of the call is chunk 0 of the video it has ONE
// keyframe (tokenflow_utils.py:331-333, 390): its rows are a1 + residual, rounded to the dtype the reference's
// single-keyframe pass would produce (`single_dtype`: torch promotion of the cached output and hidden_states)
// before they are widened to the call's output type -- so the call reproduces C separate calls bit for bit.
struct GbChunks {
    int nc;             // frames per chunk (0: not a multi-chunk call)
    int first_single;
    int single_dtype;   // TF_BF16 / TF_F16 / TF_F32
};

// MERGE: the indices come as the search's per-split partial results (tf_nn_gather_blend: no finalize launch
// in between); every thread of a token merges them itself -- `splits` 8-byte reads, broadcast from cache.
// CH: multi-chunk call (P == 2, MERGE).
template <typename TIn, typename TRes, typename TOut, int P, bool MERGE, bool CH = false>
__global__ __launch_bounds__(256) void gather_blend_kernel(const TIn* __restrict__ kf_out,
                                                           const int32_t* __restrict__ idx,
                                                           const NnPartial* __restrict__ part, int splits,
                                                           const float* __restrict__ w, const TRes* __restrict__ resid,
                                                           TOut* __restrict__ out, int K, int n, int S, int D, int kf0,
                                                           int kf1, GbChunks ch) {
    const int ppr = D >> 3;  // pieces per row
    const int64_t nS = (int64_t)n * S;
    const int64_t total = nS * ppr;
    const int64_t branch_in = (int64_t)K * S * D;
    const int64_t branch_out = nS * D;
    const int64_t frame_in = (int64_t)S * D;
    const TIn* src1 = kf_out + (int64_t)kf0 * frame_in;
    const TIn* src2 = kf_out + (int64_t)kf1 * frame_in;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        const int64_t t = g / ppr;
        const int c = (int)(g - t * ppr) * 8;
        const int i1 = MERGE ? nn_merge_partials(part + t, P * nS, splits) : idx[t];
        float w1 = 0.f, w2 = 0.f;
        int i2 = 0;
        int frame = (int)(t / S);
        int64_t coff = 0;        // CH: offset of this chunk's keyframe pair from (kf0, kf1)
        bool single = false;     // CH: this token belongs to the one-keyframe chunk
        if constexpr (CH) {
            const int j = frame / ch.nc;
            frame -= j * ch.nc;
            coff = j * frame_in;
            single = ch.first_single && j == 0;
        }
        if constexpr (P == 2) {
            if (!single) i2 = MERGE ? nn_merge_partials(part + nS + t, P * nS, splits) : idx[nS + t];
            w1 = w[frame];
            w2 = __fsub_rn(1.0f, w1);
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float a1[8], o[8];
            load8(src1 + coff + b * branch_in + (int64_t)i1 * D + c, a1);
            if (P == 2 && !single) {
                float a2[8];
                load8(src2 + coff + b * branch_in + (int64_t)i2 * D + c, a2);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = __fadd_rn(__fmul_rn(w1, a1[i]), __fmul_rn(w2, a2[i]));
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = a1[i];
            }
            const int64_t off = b * branch_out + t * D + c;
            if constexpr (!__is_same(TRes, NoRes)) {
                float h[8];
                load8_stream(resid + off, h);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = __fadd_rn(o[i], h[i]);
            }
            if constexpr (CH) {
                if (single && ch.single_dtype != TF_F32) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        o[i] = ch.single_dtype == TF_BF16 ? (float)(__bf16)o[i] : (float)(_Float16)o[i];
                }
            }
            store8_stream(out + off, o);
        }
    }
}

// The same gather + blend + residual with the block's NEXT LayerNorm fused behind it (norm2 / norm3 of
// TokenFlowBlock.forward, tokenflow_utils.py:399-414, whose only consumer is a Linear): the propagation pass leaves
// its residual stream in fp32 (the reference's promotion, 385-397), and a separate norm launch re-reads those 4 bytes
// per element only to emit 2.  Here a row of the result never leaves the registers between the two: LPR lanes own one
// (branch, token) row (the mapping and arithmetic of ln_row.h, so the normalised rows are bit-identical to
// tf_layer_norm on the stored result), write it once and write its norm in the 16-bit type the Linear reads.
// Always the MERGE form (indices from the search's partial results).  T16 = the 16-bit model type (cached attention
// output, residual, norm output); TOut = result type: float (two keyframes) or T16 (the one-keyframe chunk alone).
template <typename T16, typename TOut, int LPR, int NP, int P, bool CH>
__global__ __launch_bounds__(256) void gather_blend_norm_kernel(
    const T16* __restrict__ kf_out, const NnPartial* __restrict__ part, int splits, const float* __restrict__ w,
    const T16* __restrict__ resid, TOut* __restrict__ out, T16* __restrict__ norm_out, const void* __restrict__ gamma,
    const void* __restrict__ beta, int w_dtype, float eps, int K, int n, int S, int D, int kf0, int kf1, GbChunks ch) {
    constexpr int RPW = 64 / LPR;
    __shared__ __attribute__((aligned(16))) float sw[2][LPR * 8 * NP];
    ln_stage_weights<256>(sw[0], sw[1], gamma, beta, w_dtype, D);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int lr = lane % LPR;
    const int pieces = D >> 3;
    const float inv_d = 1.0f / (float)D;
    const int64_t nS = (int64_t)n * S;
    const int64_t rows = 3 * nS;                    // row = b * nS + t  (branch-major, as the output tensor)
    const int64_t branch_in = (int64_t)K * S * D;
    const int64_t frame_in = (int64_t)S * D;
    const T16* src1 = kf_out + (int64_t)kf0 * frame_in;
    const T16* src2 = kf_out + (int64_t)kf1 * frame_in;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    for (int64_t r = row0; r < rows; r += (int64_t)gridDim.x * 4 * RPW) {
        const int b = (int)(r / nS);
        const int64_t t = r - b * nS;
        const int i1 = nn_merge_partials(part + t, P * nS, splits);
        float w1 = 0.f, w2 = 0.f;
        int i2 = 0;
        int frame = (int)(t / S);
        int64_t coff = 0;
        bool single = false;
        if constexpr (CH) {
            const int j = frame / ch.nc;
            frame -= j * ch.nc;
            coff = j * frame_in;
            single = ch.first_single && j == 0;
        }
        if constexpr (P == 2) {
            if (!single) i2 = nn_merge_partials(part + nS + t, P * nS, splits);
            w1 = w[frame];
            w2 = __fsub_rn(1.0f, w1);
        }
        const T16* r1 = src1 + coff + b * branch_in + (int64_t)i1 * D;
        const T16* r2 = src2 + coff + b * branch_in + (int64_t)i2 * D;
        float v[NP][8];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int p = lr + LPR * j;
            if (p < pieces) {
                float a1[8], h[8];
                load8(r1 + p * 8, a1);
                if (P == 2 && !single) {
                    float a2[8];
                    load8(r2 + p * 8, a2);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[j][i] = __fadd_rn(__fmul_rn(w1, a1[i]), __fmul_rn(w2, a2[i]));
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[j][i] = a1[i];
                }
                load8_stream(resid + r * D + p * 8, h);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[j][i] = __fadd_rn(v[j][i], h[i]);
                if constexpr (CH) {   // the one-keyframe chunk: rounded to the dtype its own pass produces
                    if (single && ch.single_dtype != TF_F32) {
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            v[j][i] = ch.single_dtype == TF_BF16 ? (float)(__bf16)v[j][i] : (float)(_Float16)v[j][i];
                    }
                }
                store8_stream(out + r * D + p * 8, v[j]);
                if constexpr (sizeof(TOut) == 2) {   // the norm sees the STORED (rounded) row
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[j][i] = (float)(TOut)v[j][i];
                }
            }
        }
        (void)ln_row_finish<LPR, NP, T16>(v, lr, pieces, inv_d, eps, sw[0], sw[1], norm_out + r * D);
    }
}

__global__ __launch_bounds__(256) void inject_copy_kernel(u32x4* __restrict__ x, int64_t pieces_per_branch) {
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < pieces_per_branch;
         g += (int64_t)gridDim.x * 256) {
        const u32x4 v = x[g];
        x[pieces_per_branch + g] = v;
        x[2 * pieces_per_branch + g] = v;
    }
}

struct GbArgs {
    const void* kf_out;
    const int32_t* idx;
    const NnPartial* part;   // alternative to idx: partial results of `splits` pivot-range splits
    int splits;
    const float* w;
    const void* resid;
    void* out;
    int K, n, S, D, P, kf0, kf1;
    hipStream_t st;
    GbChunks ch;
    // fused norm (gather_blend_norm_kernel): norm_out != nullptr
    void* norm_out = nullptr;
    const void* gamma = nullptr;
    const void* beta = nullptr;
    int w_dtype = 0;
    float eps = 0.f;
};

// fused-norm form: T16 in / residual / norm out, result float (P = 2) or T16 (P = 1)
template <typename T16, typename TOut, int P, bool CH>
void launch_gbn(const GbArgs& a) {
    const int64_t rows = (int64_t)3 * a.n * a.S;
    auto go = [&](auto kern, int lpr) {
        const int rows_per_wg = 4 * (64 / lpr);
        int64_t blocks = (rows + rows_per_wg - 1) / rows_per_wg;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, a.st, (const T16*)a.kf_out, a.part, a.splits, a.w,
                           (const T16*)a.resid, (TOut*)a.out, (T16*)a.norm_out, a.gamma, a.beta, a.w_dtype, a.eps, a.K,
                           a.n, a.S, a.D, a.kf0, P == 2 ? a.kf1 : a.kf0, a.ch);
    };
    const int pieces = a.D >> 3;   // 3 pieces per lane: D <= 384 / 768 / 1536 at 16 / 32 / 64 lanes per row
    if (pieces <= 48) go(gather_blend_norm_kernel<T16, TOut, 16, 3, P, CH>, 16);
    else if (pieces <= 96) go(gather_blend_norm_kernel<T16, TOut, 32, 3, P, CH>, 32);
    else go(gather_blend_norm_kernel<T16, TOut, 64, 3, P, CH>, 64);
}

// which calls the fused-norm form covers (the hook path's: 16-bit model, residual of the model type)
static bool gbn_supported(int D, int P, int in_dtype, int res_dtype, int out_dtype, int norm_dtype, bool has_res) {
    return has_res && (in_dtype == TF_BF16 || in_dtype == TF_F16) && res_dtype == in_dtype && norm_dtype == in_dtype &&
           out_dtype == (P == 2 ? TF_F32 : in_dtype) && D % 8 == 0 && D <= 1536;
}

template <typename T16>
void dispatch_gbn(const GbArgs& a) {
    if (a.ch.nc > 0) launch_gbn<T16, float, 2, true>(a);
    else if (a.P == 2) launch_gbn<T16, float, 2, false>(a);
    else launch_gbn<T16, T16, 1, false>(a);
}

template <typename TIn, typename TRes, typename TOut>
void launch_gb(const GbArgs& a) {
    const int64_t total = (int64_t)a.n * a.S * (a.D >> 3);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    auto go = [&](auto kern, int kf1) {
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, a.st, (const TIn*)a.kf_out, a.idx, a.part,
                           a.splits, a.w, (const TRes*)a.resid, (TOut*)a.out, a.K, a.n, a.S, a.D, a.kf0, kf1, a.ch);
    };
    if (a.ch.nc > 0) {
        go(gather_blend_kernel<TIn, TRes, TOut, 2, true, true>, a.kf1);
    } else if (a.part) {
        if (a.P == 2) go(gather_blend_kernel<TIn, TRes, TOut, 2, true>, a.kf1);
        else go(gather_blend_kernel<TIn, TRes, TOut, 1, true>, a.kf0);
    } else {
        if (a.P == 2) go(gather_blend_kernel<TIn, TRes, TOut, 2, false>, a.kf1);
        else go(gather_blend_kernel<TIn, TRes, TOut, 1, false>, a.kf0);
    }
}

template <typename TIn, typename TRes>
void dispatch_out(const GbArgs& a, int out_dtype) {
    switch (out_dtype) {
        case TF_BF16: launch_gb<TIn, TRes, __bf16>(a); break;
        case TF_F16: launch_gb<TIn, TRes, _Float16>(a); break;
        default: launch_gb<TIn, TRes, float>(a); break;
    }
}

template <typename TIn>
void dispatch_res(const GbArgs& a, int res_dtype, int out_dtype) {
    if (!a.resid) return dispatch_out<TIn, NoRes>(a, out_dtype);
    switch (res_dtype) {
        case TF_BF16: dispatch_out<TIn, __bf16>(a, out_dtype); break;
        case TF_F16: dispatch_out<TIn, _Float16>(a, out_dtype); break;
        default: dispatch_out<TIn, float>(a, out_dtype); break;
    }
}

}  // namespace

extern "C" int tf_gather_blend(const void* kf_out, const int32_t* idx, const float* w, const void* resid, void* out,
                               int K, int n, int S, int D, int P, int kf0, int kf1, int in_dtype, int res_dtype,
                               int out_dtype, void* stream) {
    TF_ARG(kf_out && idx && out && (P == 1 || w), TF_ERR_NULL, "tf_gather_blend: null pointer");
    auto okdt = [](int d) { return d == TF_BF16 || d == TF_F16 || d == TF_F32; };
    TF_ARG(okdt(in_dtype) && okdt(out_dtype) && (!resid || okdt(res_dtype)), TF_ERR_DTYPE,
           "tf_gather_blend: dtypes in=%d res=%d out=%d", in_dtype, res_dtype, out_dtype);
    TF_ARG(K > 0 && n > 0 && S > 0 && D > 0 && D % 8 == 0 && (P == 1 || P == 2) && kf0 >= 0 && kf0 < K &&
               (P == 1 || (kf1 >= 0 && kf1 < K)),
           TF_ERR_SHAPE, "tf_gather_blend: K=%d n=%d S=%d D=%d P=%d kf=(%d,%d)", K, n, S, D, P, kf0, kf1);
    TF_ARG(tf_aligned16(kf_out) && tf_aligned16(out) && tf_aligned16(resid), TF_ERR_ALIGN,
           "tf_gather_blend: tensors not 16-byte aligned");
    GbArgs a{kf_out, idx, nullptr, 0, w, resid, out, K, n, S, D, P, kf0, kf1, reinterpret_cast<hipStream_t>(stream),
             GbChunks{0, 0, 0}};
    switch (in_dtype) {
        case TF_BF16: dispatch_res<__bf16>(a, res_dtype, out_dtype); break;
        case TF_F16: dispatch_res<_Float16>(a, res_dtype, out_dtype); break;
        default: dispatch_res<float>(a, res_dtype, out_dtype); break;
    }
    TF_LAUNCH_CHECK("tf_gather_blend");
    return 0;
}

extern "C" size_t tf_nn_gather_blend_workspace_bytes(int64_t n_tgt, int S, int D, int P) {
    if (n_tgt <= 0 || S <= 0 || D <= 0 || P <= 0) return 0;
    const size_t b = tf_nn_partials_bytes(n_tgt, S, D, P);
    return b < 256 ? 256 : b;
}

struct NormArgs {   // the block's next LayerNorm, fused behind the gather (all null / 0: no norm)
    void* out = nullptr;
    const void* gamma = nullptr;
    const void* beta = nullptr;
    float eps = 0.f;
    int w_dtype = 0, dtype = 0;
};

static int nn_gather_blend_impl(const char* name, const void* tgt, const void* piv, const float* inv_norm,
                                const void* kf_out, const float* w, const void* resid, void* out, int K, int n, int S,
                                int D, int P, int kf0, int kf1, int search_dtype, int in_dtype, int res_dtype,
                                int out_dtype, void* ws, size_t ws_bytes, void* stream, const NormArgs& nm) {
    TF_ARG(tgt && piv && inv_norm && kf_out && out && ws && (P == 1 || w), TF_ERR_NULL, "%s: null pointer", name);
    TF_ARG(search_dtype == TF_BF16 || search_dtype == TF_F16, TF_ERR_DTYPE, "%s: search dtype %d (bf16/f16 only)", name,
           search_dtype);
    auto okdt = [](int d) { return d == TF_BF16 || d == TF_F16 || d == TF_F32; };
    TF_ARG(okdt(in_dtype) && okdt(out_dtype) && (!resid || okdt(res_dtype)), TF_ERR_DTYPE,
           "%s: dtypes in=%d res=%d out=%d", name, in_dtype, res_dtype, out_dtype);
    TF_ARG(K > 0 && n > 0 && S > 0 && D > 0 && D % 8 == 0 && (P == 1 || P == 2) && kf0 >= 0 && kf0 < K &&
               (P == 1 || (kf1 >= 0 && kf1 < K)),
           TF_ERR_SHAPE, "%s: K=%d n=%d S=%d D=%d P=%d kf=(%d,%d)", name, K, n, S, D, P, kf0, kf1);
    TF_ARG(tf_aligned16(tgt) && tf_aligned16(piv) && tf_aligned16(kf_out) && tf_aligned16(out) &&
               tf_aligned16(resid) && tf_aligned16(ws) && tf_aligned16(inv_norm),
           TF_ERR_ALIGN, "%s: tensors not 16-byte aligned (inv_norm included)", name);
    if (nm.out) {
        TF_ARG((!nm.gamma && !nm.beta) || okdt(nm.w_dtype), TF_ERR_DTYPE, "%s: norm weight dtype %d", name, nm.w_dtype);
        TF_ARG(gbn_supported(D, P, in_dtype, res_dtype, out_dtype, nm.dtype, resid != nullptr), TF_ERR_DTYPE,
               "%s: the fused norm needs a 16-bit cached output, a residual and a norm output of that same type, and "
               "the result in fp32 (two keyframes) or that type (one); D <= 1536", name);
        TF_ARG(tf_aligned16(nm.out) && tf_aligned16(nm.gamma) && tf_aligned16(nm.beta), TF_ERR_ALIGN,
               "%s: norm tensors not 16-byte aligned", name);
    }
    const int64_t n_tgt = (int64_t)n * S;
    TF_ARG(ws_bytes >= tf_nn_gather_blend_workspace_bytes(n_tgt, S, D, P), TF_ERR_WORKSPACE,
           "%s: workspace %zu < %zu bytes", name, ws_bytes, tf_nn_gather_blend_workspace_bytes(n_tgt, S, D, P));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    NnPartial* part = reinterpret_cast<NnPartial*>(ws);
    int splits = 1;
    const int rc = tf_nn_search_partials(tgt, piv, inv_norm, part, n_tgt, S, D, P, kf0, kf1, search_dtype, st, &splits);
    if (rc) return rc;
    GbArgs a{kf_out, nullptr, part, splits, w, resid, out, K, n, S, D, P, kf0, kf1, st, GbChunks{0, 0, 0}};
    if (nm.out) {
        a.norm_out = nm.out, a.gamma = nm.gamma, a.beta = nm.beta, a.w_dtype = nm.w_dtype, a.eps = nm.eps;
        if (in_dtype == TF_BF16) dispatch_gbn<__bf16>(a);
        else dispatch_gbn<_Float16>(a);
    } else {
        switch (in_dtype) {
            case TF_BF16: dispatch_res<__bf16>(a, res_dtype, out_dtype); break;
            case TF_F16: dispatch_res<_Float16>(a, res_dtype, out_dtype); break;
            default: dispatch_res<float>(a, res_dtype, out_dtype); break;
        }
    }
    TF_LAUNCH_CHECK(name);
    return 0;
}

extern "C" int tf_nn_gather_blend(const void* tgt, const void* piv, const float* inv_norm, const void* kf_out,
                                  const float* w, const void* resid, void* out, int K, int n, int S, int D, int P,
                                  int kf0, int kf1, int search_dtype, int in_dtype, int res_dtype, int out_dtype,
                                  void* ws, size_t ws_bytes, void* stream) {
    return nn_gather_blend_impl("tf_nn_gather_blend", tgt, piv, inv_norm, kf_out, w, resid, out, K, n, S, D, P, kf0, kf1,
                                search_dtype, in_dtype, res_dtype, out_dtype, ws, ws_bytes, stream, NormArgs{});
}

extern "C" int tf_nn_gather_blend_norm(const void* tgt, const void* piv, const float* inv_norm, const void* kf_out,
                                       const float* w, const void* resid, void* out, int K, int n, int S, int D, int P,
                                       int kf0, int kf1, int search_dtype, int in_dtype, int res_dtype, int out_dtype,
                                       const void* gamma, const void* beta, float eps, int w_dtype, void* norm_out,
                                       int norm_dtype, void* ws, size_t ws_bytes, void* stream) {
    TF_ARG(norm_out, TF_ERR_NULL, "tf_nn_gather_blend_norm: null norm_out");
    NormArgs nm;
    nm.out = norm_out, nm.gamma = gamma, nm.beta = beta, nm.eps = eps, nm.w_dtype = w_dtype, nm.dtype = norm_dtype;
    return nn_gather_blend_impl("tf_nn_gather_blend_norm", tgt, piv, inv_norm, kf_out, w, resid, out, K, n, S, D, P, kf0,
                                kf1, search_dtype, in_dtype, res_dtype, out_dtype, ws, ws_bytes, stream, nm);
}

extern "C" size_t tf_nn_gather_blend_chunks_workspace_bytes(int64_t n_tgt_chunk, int S, int D, int C) {
    if (n_tgt_chunk <= 0 || S <= 0 || D <= 0 || C <= 0) return 0;
    const size_t b = tf_nn_partials_bytes(n_tgt_chunk, S, D, 2, C);
    return b < 256 ? 256 : b;
}

static int nn_gather_blend_chunks_impl(const char* name, const void* tgt, const void* piv, const float* inv_norm,
                                       const void* kf_out, const float* w, const void* resid, void* out, int K, int n,
                                       int C, int S, int D, int slot0, int first_single, int search_dtype, int in_dtype,
                                       int res_dtype, int out_dtype, int single_dtype, void* ws, size_t ws_bytes,
                                       void* stream, const NormArgs& nm) {
    TF_ARG(tgt && piv && inv_norm && kf_out && out && ws && w, TF_ERR_NULL, "%s: null pointer", name);
    TF_ARG(search_dtype == TF_BF16 || search_dtype == TF_F16, TF_ERR_DTYPE, "%s: search dtype %d (bf16/f16 only)", name,
           search_dtype);
    auto okdt = [](int d) { return d == TF_BF16 || d == TF_F16 || d == TF_F32; };
    TF_ARG(okdt(in_dtype) && okdt(out_dtype) && okdt(single_dtype) && (!resid || okdt(res_dtype)), TF_ERR_DTYPE,
           "%s: dtypes in=%d res=%d out=%d single=%d", name, in_dtype, res_dtype, out_dtype, single_dtype);
    // chunk j reads keyframe slots slot0 + j and slot0 + j - 1 (the one-keyframe chunk only slot0)
    TF_ARG(K > 0 && n > 0 && C > 0 && S > 0 && D > 0 && D % 8 == 0 && slot0 + C <= K &&
               slot0 >= (first_single ? 0 : 1) && (C > 1 || !first_single),
           TF_ERR_SHAPE, "%s: K=%d n=%d C=%d S=%d D=%d slot0=%d first_single=%d", name, K, n, C, S, D, slot0,
           first_single);
    TF_ARG(tf_aligned16(tgt) && tf_aligned16(piv) && tf_aligned16(kf_out) && tf_aligned16(out) &&
               tf_aligned16(resid) && tf_aligned16(ws) && tf_aligned16(inv_norm),
           TF_ERR_ALIGN, "%s: tensors not 16-byte aligned (inv_norm included)", name);
    if (nm.out) {
        TF_ARG((!nm.gamma && !nm.beta) || okdt(nm.w_dtype), TF_ERR_DTYPE, "%s: norm weight dtype %d", name, nm.w_dtype);
        TF_ARG(gbn_supported(D, 2, in_dtype, res_dtype, out_dtype, nm.dtype, resid != nullptr), TF_ERR_DTYPE,
               "%s: the fused norm needs a 16-bit cached output, a residual and a norm output of that same type and the "
               "result in fp32; D <= 1536", name);
        TF_ARG(tf_aligned16(nm.out) && tf_aligned16(nm.gamma) && tf_aligned16(nm.beta), TF_ERR_ALIGN,
               "%s: norm tensors not 16-byte aligned", name);
    }
    const int64_t n_tgt = (int64_t)n * S;   // per chunk
    TF_ARG(ws_bytes >= tf_nn_gather_blend_chunks_workspace_bytes(n_tgt, S, D, C), TF_ERR_WORKSPACE,
           "%s: workspace %zu < %zu bytes", name, ws_bytes, tf_nn_gather_blend_chunks_workspace_bytes(n_tgt, S, D, C));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    NnPartial* part = reinterpret_cast<NnPartial*>(ws);
    int splits = 1;
    const int rc = tf_nn_search_partials(tgt, piv, inv_norm, part, n_tgt, S, D, 2, slot0, slot0 - 1, search_dtype, st,
                                         &splits, C, first_single ? 1 : 0);
    if (rc) return rc;
    GbArgs a{kf_out, nullptr, part, splits, w, resid, out, K, n * C, S, D, 2, slot0, slot0 - 1, st,
             GbChunks{n, first_single ? 1 : 0, single_dtype}};
    if (nm.out) {
        a.norm_out = nm.out, a.gamma = nm.gamma, a.beta = nm.beta, a.w_dtype = nm.w_dtype, a.eps = nm.eps;
        if (in_dtype == TF_BF16) dispatch_gbn<__bf16>(a);
        else dispatch_gbn<_Float16>(a);
    } else {
        switch (in_dtype) {
            case TF_BF16: dispatch_res<__bf16>(a, res_dtype, out_dtype); break;
            case TF_F16: dispatch_res<_Float16>(a, res_dtype, out_dtype); break;
            default: dispatch_res<float>(a, res_dtype, out_dtype); break;
        }
    }
    TF_LAUNCH_CHECK(name);
    return 0;
}

extern "C" int tf_nn_gather_blend_chunks(const void* tgt, const void* piv, const float* inv_norm, const void* kf_out,
                                         const float* w, const void* resid, void* out, int K, int n, int C, int S,
                                         int D, int slot0, int first_single, int search_dtype, int in_dtype,
                                         int res_dtype, int out_dtype, int single_dtype, void* ws, size_t ws_bytes,
                                         void* stream) {
    return nn_gather_blend_chunks_impl("tf_nn_gather_blend_chunks", tgt, piv, inv_norm, kf_out, w, resid, out, K, n, C, S,
                                       D, slot0, first_single, search_dtype, in_dtype, res_dtype, out_dtype, single_dtype,
                                       ws, ws_bytes, stream, NormArgs{});
}

extern "C" int tf_nn_gather_blend_chunks_norm(const void* tgt, const void* piv, const float* inv_norm,
                                              const void* kf_out, const float* w, const void* resid, void* out, int K,
                                              int n, int C, int S, int D, int slot0, int first_single, int search_dtype,
                                              int in_dtype, int res_dtype, int out_dtype, int single_dtype,
                                              const void* gamma, const void* beta, float eps, int w_dtype,
                                              void* norm_out, int norm_dtype, void* ws, size_t ws_bytes, void* stream) {
    TF_ARG(norm_out, TF_ERR_NULL, "tf_nn_gather_blend_chunks_norm: null norm_out");
    NormArgs nm;
    nm.out = norm_out, nm.gamma = gamma, nm.beta = beta, nm.eps = eps, nm.w_dtype = w_dtype, nm.dtype = norm_dtype;
    return nn_gather_blend_chunks_impl("tf_nn_gather_blend_chunks_norm", tgt, piv, inv_norm, kf_out, w, resid, out, K, n,
                                       C, S, D, slot0, first_single, search_dtype, in_dtype, res_dtype, out_dtype,
                                       single_dtype, ws, ws_bytes, stream, nm);
}

extern "C" int tf_inject_copy(void* x, int64_t elems_per_branch, int elem_bytes, void* stream) {
    TF_ARG(x, TF_ERR_NULL, "tf_inject_copy: null pointer");
    TF_ARG(elems_per_branch > 0 && elem_bytes > 0 && (elems_per_branch * elem_bytes) % 16 == 0, TF_ERR_SHAPE,
           "tf_inject_copy: elems_per_branch=%lld elem_bytes=%d (bytes per branch %% 16 == 0)",
           (long long)elems_per_branch, elem_bytes);
    TF_ARG(tf_aligned16(x), TF_ERR_ALIGN, "tf_inject_copy: x not 16-byte aligned");
    const int64_t pieces = elems_per_branch * elem_bytes / 16;
    int64_t blocks = (pieces + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(inject_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<u32x4*>(x), pieces);
    TF_LAUNCH_CHECK("tf_inject_copy");
    return 0;
}
