// Row LayerNorm arithmetic shared by layer_norm.hip (tf_layer_norm / tf_add_layer_norm) and gather_blend.hip (the
// propagation's gather + blend + residual with the block's next norm fused behind it): ONE definition of the
// statistics, their reduction order and the output rounding, so that the fused producer is bit-identical to the
// separate launches whatever flags its translation unit is compiled with (every multiply-add below is explicit).
//
// Mapping: LPR (16 / 32 / 64) consecutive lanes share a row; lane lr owns the 16-byte pieces p = lr + LPR*j,
// j < NP, p < pieces = D/8, as fp32 values v[j][0..7].
#pragma once
#include "tf_common.h"

// sum over the LPR lanes that share a row
template <int LPR>
__device__ __forceinline__ float ln_row_sum(float x) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// stores 8 values rounded to T and returns the sum of squares of the ROUNDED values
template <typename T>
__device__ __forceinline__ float ln_store8(T* p, const float (&f)[8]) {
    float ss = 0.f;
    if constexpr (sizeof(T) == 4) {
        u32x4 a, b;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = __float_as_uint(f[i]);
            b[i] = __float_as_uint(f[4 + i]);
        }
        st16(p, a);
        st16(p + 4, b);
#pragma unroll
        for (int i = 0; i < 8; ++i) ss = fmaf(f[i], f[i], ss);
    } else {
        typedef T v8 __attribute__((ext_vector_type(8)));
        v8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // the fp32 value exists as such before it is rounded (torch: fp32 layer_norm, then the cast).  Without the
            // barrier hipcc fuses `(f16)fmaf(..)` into v_fma_mixlo_f16 in one translation unit and emits v_pk_fma_f32 +
            // v_cvt_pk_f16_f32 in the other, and the two differ in rare elements (seen on MI355X, f16 only)
            float t = f[i];
            asm("" : "+v"(t));
            v[i] = (T)t;
            ss = fmaf((float)v[i], (float)v[i], ss);
        }
        st16(p, __builtin_bit_cast(u32x4, v));
    }
    return ss;
}

// gamma / beta of a norm as fp32 in LDS (sw[0] = gamma or 1, sw[1] = beta or 0), converted once per workgroup:
// no dtype switch and no global load in the row loop.  Call from all threads of the workgroup, then __syncthreads().
template <int NT>
__device__ __forceinline__ void ln_stage_weights(float* sw_g, float* sw_b, const void* gamma, const void* beta,
                                                 int w_dtype, int D) {
    for (int c = threadIdx.x; c < D; c += NT) {
        float g = 1.f, b = 0.f;
        if (gamma)
            g = w_dtype == TF_F32    ? reinterpret_cast<const float*>(gamma)[c]
                : w_dtype == TF_BF16 ? (float)reinterpret_cast<const __bf16*>(gamma)[c]
                                     : (float)reinterpret_cast<const _Float16*>(gamma)[c];
        if (beta)
            b = w_dtype == TF_F32    ? reinterpret_cast<const float*>(beta)[c]
                : w_dtype == TF_BF16 ? (float)reinterpret_cast<const __bf16*>(beta)[c]
                                     : (float)reinterpret_cast<const _Float16*>(beta)[c];
        sw_g[c] = g;
        sw_b[c] = b;
    }
}

// mean, biased variance (two passes over the register-resident row, as torch), y = (x - mean) * rstd * gamma + beta
// rounded ONCE to TOut and stored; returns this lane's share of sum(y_rounded^2) (reduce with ln_row_sum for
// 1/||y||_2).
template <int LPR, int NP, typename TOut>
__device__ __forceinline__ float ln_row_finish(const float (&v)[NP][8], int lr, int pieces, float inv_d, float eps,
                                               const float* sw_g, const float* sw_b, TOut* orow) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j)
        if (lr + LPR * j < pieces) {
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[j][i];
        }
    // __fmul_rn: a multiply the compiler may NOT contract with the subtractions below into an fma -- this header is
    // compiled into translation units with different -ffp-contract settings (gather_blend.o: off, layer_norm.o: the
    // default), and the fused-norm forms must stay bit-identical to tf_layer_norm in both
    const float mean = __fmul_rn(ln_row_sum<LPR>(s), inv_d);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j)
        if (lr + LPR * j < pieces) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float c = v[j][i] - mean;
                q = fmaf(c, c, q);
            }
        }
    const float rstd = 1.0f / __builtin_sqrtf(fmaf(ln_row_sum<LPR>(q), inv_d, eps));   // biased variance, as torch
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int p = lr + LPR * j;
        if (p < pieces) {
            float y[8];
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(sw_g + p * 8);
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(sw_g + p * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sw_b + p * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(sw_b + p * 8 + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                y[i] = fmaf(__fmul_rn(v[j][i] - mean, rstd), g0[i], b0[i]);
                y[4 + i] = fmaf(__fmul_rn(v[j][4 + i] - mean, rstd), g1[i], b1[i]);
            }
            ss += ln_store8(orow + p * 8, y);
        }
    }
    return ss;
}
