// Token gather + two-keyframe blend + residual add, and the PnP feature-injection copy.
// Replaces tokenflow_utils.py:362-397 and 87-91 of omerbt/TokenFlow.
//
// HBM-bound: each output row (D elements, 640 B .. 2.5 KB) is a contiguous row of one
// keyframe's cached attention output, so a row gather is naturally coalesced.  One
// thread owns one 8-element piece of one token and loops the 3 branches, re-using the
// token's two int32 indices and the frame's weight; 16-byte loads/stores; fp32 math
// with the reference's operation order and NO fma contraction, so fp32 results are
// bit-identical to torch's  w1*a1 + (1-w1)*a2  (+ hidden_states).
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
};

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

extern "C" int tf_nn_gather_blend(const void* tgt, const void* piv, const float* inv_norm, const void* kf_out,
                                  const float* w, const void* resid, void* out, int K, int n, int S, int D, int P,
                                  int kf0, int kf1, int search_dtype, int in_dtype, int res_dtype, int out_dtype,
                                  void* ws, size_t ws_bytes, void* stream) {
    TF_ARG(tgt && piv && inv_norm && kf_out && out && ws && (P == 1 || w), TF_ERR_NULL,
           "tf_nn_gather_blend: null pointer");
    TF_ARG(search_dtype == TF_BF16 || search_dtype == TF_F16, TF_ERR_DTYPE,
           "tf_nn_gather_blend: search dtype %d (bf16/f16 only)", search_dtype);
    auto okdt = [](int d) { return d == TF_BF16 || d == TF_F16 || d == TF_F32; };
    TF_ARG(okdt(in_dtype) && okdt(out_dtype) && (!resid || okdt(res_dtype)), TF_ERR_DTYPE,
           "tf_nn_gather_blend: dtypes in=%d res=%d out=%d", in_dtype, res_dtype, out_dtype);
    TF_ARG(K > 0 && n > 0 && S > 0 && D > 0 && D % 8 == 0 && (P == 1 || P == 2) && kf0 >= 0 && kf0 < K &&
               (P == 1 || (kf1 >= 0 && kf1 < K)),
           TF_ERR_SHAPE, "tf_nn_gather_blend: K=%d n=%d S=%d D=%d P=%d kf=(%d,%d)", K, n, S, D, P, kf0, kf1);
    TF_ARG(tf_aligned16(tgt) && tf_aligned16(piv) && tf_aligned16(kf_out) && tf_aligned16(out) &&
               tf_aligned16(resid) && tf_aligned16(ws),
           TF_ERR_ALIGN, "tf_nn_gather_blend: tensors not 16-byte aligned");
    const int64_t n_tgt = (int64_t)n * S;
    TF_ARG(ws_bytes >= tf_nn_gather_blend_workspace_bytes(n_tgt, S, D, P), TF_ERR_WORKSPACE,
           "tf_nn_gather_blend: workspace %zu < %zu bytes", ws_bytes,
           tf_nn_gather_blend_workspace_bytes(n_tgt, S, D, P));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    NnPartial* part = reinterpret_cast<NnPartial*>(ws);
    int splits = 1;
    const int rc = tf_nn_search_partials(tgt, piv, inv_norm, part, n_tgt, S, D, P, kf0, kf1, search_dtype, st, &splits);
    if (rc) return rc;
    GbArgs a{kf_out, nullptr, part, splits, w, resid, out, K, n, S, D, P, kf0, kf1, st, GbChunks{0, 0, 0}};
    switch (in_dtype) {
        case TF_BF16: dispatch_res<__bf16>(a, res_dtype, out_dtype); break;
        case TF_F16: dispatch_res<_Float16>(a, res_dtype, out_dtype); break;
        default: dispatch_res<float>(a, res_dtype, out_dtype); break;
    }
    TF_LAUNCH_CHECK("tf_nn_gather_blend");
    return 0;
}

extern "C" size_t tf_nn_gather_blend_chunks_workspace_bytes(int64_t n_tgt_chunk, int S, int D, int C) {
    if (n_tgt_chunk <= 0 || S <= 0 || D <= 0 || C <= 0) return 0;
    const size_t b = tf_nn_partials_bytes(n_tgt_chunk, S, D, 2, C);
    return b < 256 ? 256 : b;
}

extern "C" int tf_nn_gather_blend_chunks(const void* tgt, const void* piv, const float* inv_norm, const void* kf_out,
                                         const float* w, const void* resid, void* out, int K, int n, int C, int S,
                                         int D, int slot0, int first_single, int search_dtype, int in_dtype,
                                         int res_dtype, int out_dtype, int single_dtype, void* ws, size_t ws_bytes,
                                         void* stream) {
    TF_ARG(tgt && piv && inv_norm && kf_out && out && ws && w, TF_ERR_NULL, "tf_nn_gather_blend_chunks: null pointer");
    TF_ARG(search_dtype == TF_BF16 || search_dtype == TF_F16, TF_ERR_DTYPE,
           "tf_nn_gather_blend_chunks: search dtype %d (bf16/f16 only)", search_dtype);
    auto okdt = [](int d) { return d == TF_BF16 || d == TF_F16 || d == TF_F32; };
    TF_ARG(okdt(in_dtype) && okdt(out_dtype) && okdt(single_dtype) && (!resid || okdt(res_dtype)), TF_ERR_DTYPE,
           "tf_nn_gather_blend_chunks: dtypes in=%d res=%d out=%d single=%d", in_dtype, res_dtype, out_dtype,
           single_dtype);
    // chunk j reads keyframe slots slot0 + j and slot0 + j - 1 (the one-keyframe chunk only slot0)
    TF_ARG(K > 0 && n > 0 && C > 0 && S > 0 && D > 0 && D % 8 == 0 && slot0 + C <= K &&
               slot0 >= (first_single ? 0 : 1) && (C > 1 || !first_single),
           TF_ERR_SHAPE, "tf_nn_gather_blend_chunks: K=%d n=%d C=%d S=%d D=%d slot0=%d first_single=%d", K, n, C, S, D,
           slot0, first_single);
    TF_ARG(tf_aligned16(tgt) && tf_aligned16(piv) && tf_aligned16(kf_out) && tf_aligned16(out) &&
               tf_aligned16(resid) && tf_aligned16(ws),
           TF_ERR_ALIGN, "tf_nn_gather_blend_chunks: tensors not 16-byte aligned");
    const int64_t n_tgt = (int64_t)n * S;   // per chunk
    TF_ARG(ws_bytes >= tf_nn_gather_blend_chunks_workspace_bytes(n_tgt, S, D, C), TF_ERR_WORKSPACE,
           "tf_nn_gather_blend_chunks: workspace %zu < %zu bytes", ws_bytes,
           tf_nn_gather_blend_chunks_workspace_bytes(n_tgt, S, D, C));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    NnPartial* part = reinterpret_cast<NnPartial*>(ws);
    int splits = 1;
    const int rc = tf_nn_search_partials(tgt, piv, inv_norm, part, n_tgt, S, D, 2, slot0, slot0 - 1, search_dtype, st,
                                         &splits, C, first_single ? 1 : 0);
    if (rc) return rc;
    GbArgs a{kf_out, nullptr, part, splits, w, resid, out, K, n * C, S, D, 2, slot0, slot0 - 1, st,
             GbChunks{n, first_single ? 1 : 0, single_dtype}};
    switch (in_dtype) {
        case TF_BF16: dispatch_res<__bf16>(a, res_dtype, out_dtype); break;
        case TF_F16: dispatch_res<_Float16>(a, res_dtype, out_dtype); break;
        default: dispatch_res<float>(a, res_dtype, out_dtype); break;
    }
    TF_LAUNCH_CHECK("tf_nn_gather_blend_chunks");
    return 0;
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
