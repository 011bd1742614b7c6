// Row LayerNorm producer of the hot path: replaces the `norm1` call of
// TokenFlowBlock.forward (tokenflow_utils.py:313-323) -- and norm2/norm3 of the same forward -- when the
// block runs in 16 bit.  Under autocast torch evaluates layer_norm in fp32: a cast kernel up, the norm,
// and a cast down in front of the next Linear, 20 bytes of HBM traffic per element where 4 are needed.
// Here: one pass, 16-bit (or fp32) in, fp32 statistics, one rounding to the output type, and -- for the
// pivots of the NN search -- 1/||y||_2 of the ROUNDED output row as a side product (what
// tf_pivot_inv_norm computes from the stored pivots, util.py:67), so the pivots are never re-read.
// 16/32/64 lanes per row by D, the row lives in registers (D <= 2048); HBM-bound.
#include <type_traits>

#include "ln_row.h"
#include "tf_common.h"

namespace {

template <typename T>
__device__ __forceinline__ void ln_load8(const T* p, float (&f)[8]) {
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

__device__ __forceinline__ void ln_load_w(const void* w, int w_dtype, int col, float dflt, float (&f)[8]) {
    if (w == nullptr) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = dflt;
    } else if (w_dtype == TF_F32) {
        ln_load8(reinterpret_cast<const float*>(w) + col, f);
    } else if (w_dtype == TF_BF16) {
        ln_load8(reinterpret_cast<const __bf16*>(w) + col, f);
    } else {
        ln_load8(reinterpret_cast<const _Float16*>(w) + col, f);
    }
}

// residual-add form (tf_add_layer_norm): rounds the 8 sums to `dtype` (what torch's `a + b` stores), writes them and
// leaves the ROUNDED values in f -- the LayerNorm then sees exactly the tensor the reference normalises
__device__ __forceinline__ void ln_store_sum(void* p, int dtype, int64_t off, float (&f)[8]) {
    if (dtype == TF_F32) {
        (void)ln_store8(reinterpret_cast<float*>(p) + off, f);
    } else if (dtype == TF_BF16) {
        (void)ln_store8(reinterpret_cast<__bf16*>(p) + off, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)(__bf16)f[i];
    } else {
        (void)ln_store8(reinterpret_cast<_Float16*>(p) + off, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)(_Float16)f[i];
    }
}

constexpr int LN_MAXP = 4;   // 16-B pieces per lane

// 8 consecutive elements as they come from memory (16 B of a 16-bit type, 32 B of fp32): loads are issued as raw
// words and converted when the row is processed, so that the NEXT row group's loads are in flight while the current
// one is reduced and stored
struct Raw8 {
    u32x4 a, b;
};
struct Raw4 {
    u32x4 a;
};
template <typename T>
__device__ __forceinline__ void raw_load(const T* p, Raw8& r) {
    r.a = ld16(p);
    r.b = ld16(p + 4);
}
template <typename T>
__device__ __forceinline__ void raw_load(const T* p, Raw4& r) {
    r.a = ld16(p);
}
__device__ __forceinline__ void raw_load_dt(const void* base, int dtype, int64_t off, Raw8& r) {
    if (dtype == TF_F32) {
        r.a = ld16(reinterpret_cast<const float*>(base) + off);
        r.b = ld16(reinterpret_cast<const float*>(base) + off + 4);
    } else {
        r.a = ld16(reinterpret_cast<const uint16_t*>(base) + off);
    }
}
template <typename T>
__device__ __forceinline__ void raw_cvt(const Raw8& r, float (&f)[8]) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[i] = __uint_as_float(r.a[i]);
            f[4 + i] = __uint_as_float(r.b[i]);
        }
    } else {
        typedef T v8 __attribute__((ext_vector_type(8)));
        const v8 v = __builtin_bit_cast(v8, r.a);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
    }
}
template <typename T>
__device__ __forceinline__ void raw_cvt(const Raw4& r, float (&f)[8]) {
    typedef T v8 __attribute__((ext_vector_type(8)));
    const v8 v = __builtin_bit_cast(v8, r.a);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
__device__ __forceinline__ void raw_cvt_dt(const Raw8& r, int dtype, float dflt, bool have, float (&f)[8]) {
    if (!have) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = dflt;
    } else if (dtype == TF_F32) {
        raw_cvt<float>(r, f);
    } else if (dtype == TF_BF16) {
        raw_cvt<__bf16>(r, f);
    } else {
        raw_cvt<_Float16>(r, f);
    }
}

// LPR lanes per row, 64/LPR rows per wave: a row of D = 320 is 40 pieces -- with a whole wave per row a third of
// the lanes idle and every row pays three full-wave reductions; 16 lanes per row keep all lanes busy on 4 rows.
// D <= LPR * 8 * LN_MAXP.
// A wave walks its row groups with a one-deep software pipeline: the raw loads of the next group (and, once, the
// norm's weights -- they depend on the piece only) are issued before the current group is reduced, normalised and
// stored, so the input latency is paid once per wave, not once per row group.
// ADD: the input row is `x + res` (res of runtime dtype res_dtype), rounded to sum_dtype and written to sum_out
// first -- `hidden_states = attn_output + hidden_states` followed by the next norm of the block
// (tokenflow_utils.py:396-403, 409-414) in one pass.
// NP = 16-B pieces per lane the kernel is unrolled for (3 covers D = 320 / 640 / 1280 at 16 / 32 / 64 lanes per row).
template <typename TIn, typename TOut, int LPR, bool ADD = false, int NP = LN_MAXP>
__global__ __launch_bounds__(256) void layer_norm_kernel(const TIn* __restrict__ x, const void* __restrict__ gamma,
                                                         const void* __restrict__ beta, TOut* __restrict__ out,
                                                         float* __restrict__ inv_norm, int64_t rows, int D, float eps,
                                                         int w_dtype, const void* __restrict__ res = nullptr,
                                                         int res_dtype = 0, void* __restrict__ sum_out = nullptr,
                                                         int sum_dtype = 0) {
    constexpr int RPW = 64 / LPR;   // rows per wave
    __shared__ __attribute__((aligned(16))) float sw[2][LPR * 8 * NP];   // gamma / beta as fp32 (ln_row.h)
    ln_stage_weights<256>(sw[0], sw[1], gamma, beta, w_dtype, D);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int lr = lane % LPR;      // lane within its row
    const int pieces = D >> 3;
    const float inv_d = 1.0f / (float)D;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const int64_t stride = (int64_t)gridDim.x * 4 * RPW;

    typedef typename std::conditional<sizeof(TIn) == 4, Raw8, Raw4>::type RawIn;
    RawIn nx[NP];               // next row group: x as loaded
    Raw8 nr[ADD ? NP : 1];      // ... and res (runtime dtype)
    auto prefetch = [&](int64_t r) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int p = lr + LPR * j;
            if (p < pieces) {
                raw_load(x + r * D + p * 8, nx[j]);
                if constexpr (ADD) raw_load_dt(res, res_dtype, r * D + p * 8, nr[j]);
            }
        }
    };
    if (row0 < rows) prefetch(row0);
    for (int64_t r = row0; r < rows; r += stride) {
        float v[NP][8];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int p = lr + LPR * j;
            if (p < pieces) {
                raw_cvt<TIn>(nx[j], v[j]);
                if constexpr (ADD) {
                    float rr[8];
                    raw_cvt_dt(nr[j], res_dtype, 0.f, true, rr);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[j][i] = __fadd_rn(v[j][i], rr[i]);
                }
            }
        }
        if (r + stride < rows) prefetch(r + stride);   // in flight while this group is processed
        if constexpr (ADD) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int p = lr + LPR * j;
                if (p < pieces) ln_store_sum(sum_out, sum_dtype, r * D + p * 8, v[j]);
            }
        }
        float ss = ln_row_finish<LPR, NP, TOut>(v, lr, pieces, inv_d, eps, sw[0], sw[1], out + r * D);
        if (inv_norm != nullptr) {
            ss = ln_row_sum<LPR>(ss);
            if (lr == 0) inv_norm[r] = 1.0f / __builtin_sqrtf(ss);
        }
    }
}

struct LnAdd {   // residual-add form: nullptr res = plain LayerNorm
    const void* res;
    int res_dtype;
    void* sum_out;
    int sum_dtype;
};

template <typename TIn, typename TOut, int LPR>
void launch_ln_lpr(const void* x, const void* gamma, const void* beta, void* out, float* inv_norm, int64_t rows,
                   int D, float eps, int w_dtype, hipStream_t st, const LnAdd& ad) {
    constexpr int rows_per_wg = 4 * (64 / LPR);
    int64_t blocks = (rows + rows_per_wg - 1) / rows_per_wg;
#ifndef TF_TUNE_LN_WGS_PER_CU
#define TF_TUNE_LN_WGS_PER_CU 8
#endif
    if (blocks > 256 * TF_TUNE_LN_WGS_PER_CU) blocks = 256 * TF_TUNE_LN_WGS_PER_CU;   // resident once; the waves loop
    const bool np3 = (D >> 3) <= 3 * LPR;   // 3 pieces per lane suffice (every SD width): fewer registers
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const TIn*>(x), gamma, beta,
                           reinterpret_cast<TOut*>(out), inv_norm, rows, D, eps, w_dtype, ad.res, ad.res_dtype,
                           ad.sum_out, ad.sum_dtype);
    };
    if (ad.res) {
        if (np3) go(layer_norm_kernel<TIn, TOut, LPR, true, 3>);
        else go(layer_norm_kernel<TIn, TOut, LPR, true, LN_MAXP>);
    } else {
        if (np3) go(layer_norm_kernel<TIn, TOut, LPR, false, 3>);
        else go(layer_norm_kernel<TIn, TOut, LPR, false, LN_MAXP>);
    }
}

template <typename TIn, typename TOut>
void launch_ln(const void* x, const void* gamma, const void* beta, void* out, float* inv_norm, int64_t rows, int D,
               float eps, int w_dtype, hipStream_t st, const LnAdd& ad) {
    if (D <= 16 * 8 * LN_MAXP) launch_ln_lpr<TIn, TOut, 16>(x, gamma, beta, out, inv_norm, rows, D, eps, w_dtype, st, ad);
    else if (D <= 32 * 8 * LN_MAXP) launch_ln_lpr<TIn, TOut, 32>(x, gamma, beta, out, inv_norm, rows, D, eps, w_dtype, st, ad);
    else launch_ln_lpr<TIn, TOut, 64>(x, gamma, beta, out, inv_norm, rows, D, eps, w_dtype, st, ad);
}

template <typename TIn>
void dispatch_ln_out(int out_dtype, const void* x, const void* gamma, const void* beta, void* out, float* inv_norm,
                     int64_t rows, int D, float eps, int w_dtype, hipStream_t st, const LnAdd& ad = LnAdd{}) {
    switch (out_dtype) {
        case TF_BF16: launch_ln<TIn, __bf16>(x, gamma, beta, out, inv_norm, rows, D, eps, w_dtype, st, ad); break;
        case TF_F16: launch_ln<TIn, _Float16>(x, gamma, beta, out, inv_norm, rows, D, eps, w_dtype, st, ad); break;
        default: launch_ln<TIn, float>(x, gamma, beta, out, inv_norm, rows, D, eps, w_dtype, st, ad); break;
    }
}

}  // namespace

extern "C" int tf_layer_norm(const void* x, const void* gamma, const void* beta, void* out, float* inv_norm,
                             int64_t rows, int D, float eps, int in_dtype, int w_dtype, int out_dtype,
                             void* stream) {
    TF_ARG(x && out, TF_ERR_NULL, "tf_layer_norm: null pointer");
    auto okdt = [](int d) { return d == TF_BF16 || d == TF_F16 || d == TF_F32; };
    TF_ARG(okdt(in_dtype) && okdt(out_dtype) && ((!gamma && !beta) || okdt(w_dtype)), TF_ERR_DTYPE,
           "tf_layer_norm: dtypes in=%d w=%d out=%d", in_dtype, w_dtype, out_dtype);
    TF_ARG(rows > 0 && D > 0 && D % 8 == 0 && D <= 64 * 8 * LN_MAXP, TF_ERR_SHAPE,
           "tf_layer_norm: rows=%lld D=%d (D %% 8 == 0, D <= %d)", (long long)rows, D, 64 * 8 * LN_MAXP);
    TF_ARG(tf_aligned16(x) && tf_aligned16(out) && tf_aligned16(gamma) && tf_aligned16(beta), TF_ERR_ALIGN,
           "tf_layer_norm: tensors not 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (in_dtype) {
        case TF_BF16: dispatch_ln_out<__bf16>(out_dtype, x, gamma, beta, out, inv_norm, rows, D, eps, w_dtype, st); break;
        case TF_F16: dispatch_ln_out<_Float16>(out_dtype, x, gamma, beta, out, inv_norm, rows, D, eps, w_dtype, st); break;
        default: dispatch_ln_out<float>(out_dtype, x, gamma, beta, out, inv_norm, rows, D, eps, w_dtype, st); break;
    }
    TF_LAUNCH_CHECK("tf_layer_norm");
    return 0;
}

extern "C" int tf_add_layer_norm(const void* a, const void* b, void* sum_out, const void* gamma, const void* beta,
                                 void* out, int64_t rows, int D, float eps, int a_dtype, int b_dtype, int sum_dtype,
                                 int w_dtype, int out_dtype, void* stream) {
    TF_ARG(a && b && sum_out && out, TF_ERR_NULL, "tf_add_layer_norm: null pointer");
    auto okdt = [](int d) { return d == TF_BF16 || d == TF_F16 || d == TF_F32; };
    TF_ARG(okdt(a_dtype) && okdt(b_dtype) && okdt(sum_dtype) && okdt(out_dtype) && ((!gamma && !beta) || okdt(w_dtype)),
           TF_ERR_DTYPE, "tf_add_layer_norm: dtypes a=%d b=%d sum=%d w=%d out=%d", a_dtype, b_dtype, sum_dtype, w_dtype,
           out_dtype);
    TF_ARG(rows > 0 && D > 0 && D % 8 == 0 && D <= 64 * 8 * LN_MAXP, TF_ERR_SHAPE,
           "tf_add_layer_norm: rows=%lld D=%d (D %% 8 == 0, D <= %d)", (long long)rows, D, 64 * 8 * LN_MAXP);
    TF_ARG(tf_aligned16(a) && tf_aligned16(b) && tf_aligned16(sum_out) && tf_aligned16(out) && tf_aligned16(gamma) &&
               tf_aligned16(beta),
           TF_ERR_ALIGN, "tf_add_layer_norm: tensors not 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const LnAdd ad{b, b_dtype, sum_out, sum_dtype};
    switch (a_dtype) {
        case TF_BF16: dispatch_ln_out<__bf16>(out_dtype, a, gamma, beta, out, nullptr, rows, D, eps, w_dtype, st, ad); break;
        case TF_F16: dispatch_ln_out<_Float16>(out_dtype, a, gamma, beta, out, nullptr, rows, D, eps, w_dtype, st, ad); break;
        default: dispatch_ln_out<float>(out_dtype, a, gamma, beta, out, nullptr, rows, D, eps, w_dtype, st, ad); break;
    }
    TF_LAUNCH_CHECK("tf_add_layer_norm");
    return 0;
}
