// Shared device/host helpers for the gfx950 kernels of libtokenflow_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tokenflow_hip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---- element traits: the two 16-bit MFMA input types of gfx950 -------------
struct BF16 {
    typedef __bf16 elem;
    typedef __bf16 vec8 __attribute__((ext_vector_type(8)));
    typedef __bf16 vec4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    // 16x16x32: A lane l = row l & 15, k = 8 (l >> 4) .. +7; B lane l = column l & 15, same k; C/D lane l = column l & 15,
    // rows 4 (l >> 4) + i, i = 0..3
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    // 16 blocks of 4x4x4: lane l = block l / 4, column l % 4 of B and of D.  With A = ones every lane gets the sum of its
    // own 4 B values in all 4 result registers (tools/ubench/mfma4_probe.hip)
    static __device__ __forceinline__ f32x4 mfma4(vec4 a, vec4 b, f32x4 c) {
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    }
};
struct F16 {
    typedef _Float16 elem;
    typedef _Float16 vec8 __attribute__((ext_vector_type(8)));
    typedef _Float16 vec4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma4(vec4 a, vec4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0);
    }
};

// 32x32x16 MFMA fragment conventions used everywhere in this library
// (cdna_hip_programming.md §3):
//   A (32 x 16): lane l holds row (l & 31), k = 8*(l >> 5) .. +7   (8 contiguous elements)
//   B (16 x 32): lane l holds col (l & 31), k = 8*(l >> 5) .. +7
//   C/D (32 x 32): lane l holds col (l & 31), rows (r & 3) + 8*(r >> 2) + 4*(l >> 5), r = 0..15
// Only the C/D map is relied on for addressing; A and B always use the SAME
// lane->k assignment, so a contraction is correct for any k order.
__device__ __forceinline__ int cd_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

template <typename T>
__device__ __forceinline__ float to_f32(T x) { return (float)x; }

// 1 / ||row||_2 of one row of D 16-bit elements, computed by one wave (every lane returns the value): the arithmetic of
// tf_pivot_inv_norm, shared with the kernels that produce the inverse norms on the side (head pack of the rank
// executor) so that they are bit-identical to it.
template <typename T>
__device__ __forceinline__ float tf_row_inv_norm(const typename T::elem* row, int D, int lane) {
    const int pieces = D >> 3;
    float s = 0.f;
    for (int p = lane; p < pieces; p += 64) {
        const typename T::vec8 v = __builtin_bit_cast(typename T::vec8, ld16(row + p * 8));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = (float)v[i];
            s = fmaf(x, x, s);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    return 1.0f / sqrtf(s);
}

// head pack with the pivots' inverse norms on the side (csrc/head_exchange.hip; piv = nullptr: plain tf_head_pack)
int tf_head_pack_norm(const void* const* slabs, const int64_t* frame_strides, int ns, void* send, int W, int Kl, int S,
                      int hd, int64_t ld, int elem_bytes, const void* piv, float* inv_norm, int64_t rows, int D, int dtype,
                      void* stream);

// ---- nearest-neighbour partial results (nn_search.hip -> gather_blend.hip) ----------------------
// One (best score, index) pair per (pivot-range split, keyframe, target); merged in ascending split
// order with a strict '>' so the first index wins ties.
struct NnPartial {
    float v;
    int i;
};
__device__ __forceinline__ int nn_merge_partials(const NnPartial* part, int64_t stride, int splits) {
    NnPartial b = part[0];
    for (int s = 1; s < splits; ++s) {
        const NnPartial c = part[(int64_t)s * stride];
        if (c.v > b.v) b = c;
    }
    return b.i;
}
// A launch over C consecutive chunks of the video (C = 1: the reference's one chunk per call): target panel ->
// chunk j = panel / ppc, whose nS targets are matched against keyframe slots kf0 + j and kf1 + j.
struct NnChunks {
    int64_t nS;        // targets per chunk (n frames * S tokens)
    int ppc;           // target panels per chunk
    int first_single;  // chunk 0 of the launch is chunk 0 of the video: ONE keyframe (tokenflow_utils.py:331-333)
};
// Search only (no finalize launch): partial results [splits][P][C * n_tgt] into `part`; arguments already validated.
// n_tgt = targets per chunk.
int tf_nn_search_partials(const void* tgt, const void* piv, const float* inv_norm, NnPartial* part, int64_t n_tgt,
                          int S, int D, int P, int kf0, int kf1, int dtype, hipStream_t st, int* splits, int C = 1,
                          int first_single = 0);
size_t tf_nn_partials_bytes(int64_t n_tgt, int S, int D, int P, int C = 1);

// ---- host-side error plumbing ------------------------------------------------
void tf_set_error(const char* fmt, ...);

#define TF_ARG(cond, code, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            tf_set_error(__VA_ARGS__);     \
            return (code);                 \
        }                                  \
    } while (0)

#define TF_LAUNCH_CHECK(name)                                                   \
    do {                                                                        \
        hipError_t e_ = hipGetLastError();                                      \
        if (e_ != hipSuccess) {                                                 \
            tf_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return (int)e_;                                                     \
        }                                                                       \
    } while (0)

static inline bool tf_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int tf_elem_bytes(int dtype) { return dtype == TF_F32 ? 4 : 2; }
