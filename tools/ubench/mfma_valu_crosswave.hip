// Do the matrix pipe and the VALU of one SIMD run concurrently when the two kinds of work come from DIFFERENT
// waves?  8 waves per workgroup, one workgroup per CU (100 KB of LDS requested), every wave loops over "tiles":
//   matrix tile = 14 x v_mfma_f32_32x32x16_bf16 (4 independent accumulators, round robin)      448 pipe cycles
//   vector tile = 32 x v_exp_f32 + 16 x v_cvt_pk_bf16_f32 + 8 x v_fma (the softmax mix)        ~370 cycles
// ROLE 0: every wave runs matrix tiles      ROLE 1: every wave runs vector tiles
// ROLE 2: waves 0-3 matrix, waves 4-7 vector   ROLE 3: even waves matrix, odd waves vector
// ROLE 4: every wave alternates matrix tile / vector tile (same total work as 2 and 3 per pair of waves)
// ROLE 5: as 4, but in enforced anti-phase: waves 4-7 start half a period late and there is a workgroup barrier after
//         every half, and the halves depend on each other as in attention (scores <- MFMA, P -> MFMA B operand)
// ROLE 6: as 5 without the anti-phase (all waves in phase, barrier after every half)
// If the pipes overlap across waves, ROLE 2 (or 3) takes ~max(T0, T1) / 2; if not, ~(T0 + T1) / 2.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_crosswave mfma_valu_crosswave.hip && ./mfma_valu_crosswave
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// LDSA: every MFMA takes its A operand from LDS (one ds_read_b128 per MFMA, as in the attention kernel)
template <bool LDSA>
__device__ __forceinline__ void matrix_tile(f32x16 (&acc)[4], bf16x8 a, bf16x8 b, const unsigned char* base) {
#pragma unroll
    for (int m = 0; m < 14; ++m) {
        bf16x8 av = a;
        if (LDSA) av = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + m * 1168));
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b, acc[m & 3], 0, 0, 0);
    }
}
__device__ __forceinline__ void vector_tile(float (&s)[32], bf16x8 (&p)[4]) {
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        float x = s[r];
        if ((r & 3) == 0) x = fmaf(x, 1.0001f, -0.5f);
        p[r >> 3][r & 7] = (__bf16)__builtin_amdgcn_exp2f(x);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(p[j]));
#pragma unroll
    for (int r = 0; r < 32; ++r) asm volatile("" : "+v"(s[r]));
}

template <int ROLE, bool LDSA>
__global__ __launch_bounds__(512) void k(const bf16x8* in, float* out, int iters) {
    extern __shared__ unsigned char lds[];
    const int l = threadIdx.x, wave = l >> 6;
    for (int i = l; i < 2048; i += 512) reinterpret_cast<u32x4*>(lds)[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    const unsigned char* base = lds + (l & 31) * 112 + (l & 32) / 2 + wave * 16;   // 7-slot row stride: conflict-free
    bf16x8 a = in[l & 63], b = in[64 + (l & 63)];
    f32x16 acc[4];
    float s[32];
    bf16x8 p[4] = {a, a, a, a};
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int r = 0; r < 32; ++r) s[r] = -0.001f * (l + r);
    const bool mat = ROLE == 0 || (ROLE == 2 && wave < 4) || (ROLE == 3 && !(wave & 1));
    const bool vec = ROLE == 1 || (ROLE == 2 && wave >= 4) || (ROLE == 3 && (wave & 1));
    for (int it = 0; it < iters; ++it) {
        if (ROLE == 4) {
            matrix_tile<LDSA>(acc, a, b, base);
            vector_tile(s, p);
        } else if (ROLE == 5 || ROLE == 6 || ROLE == 7) {
            if (ROLE != 6 && it == 0 && wave >= 4) __syncthreads();
            matrix_tile<LDSA>(acc, a, ROLE == 7 ? b : p[0], base);
#pragma unroll
            for (int r = 0; r < 16; ++r) if (ROLE != 7) { s[r] = acc[2][r]; s[16 + r] = acc[3][r]; }   // scores = this half's MFMA results
            __syncthreads();
            vector_tile(s, p);
            __syncthreads();
            if (ROLE != 6 && it == iters - 1 && wave < 4) __syncthreads();
        } else if (mat) {
            matrix_tile<LDSA>(acc, a, b, base);
            matrix_tile<LDSA>(acc, a, b, base);
        } else if (vec) {
            vector_tile(s, p);
            vector_tile(s, p);
        }
    }
    float r = 0.f;
    for (int j = 0; j < 4; ++j) r += acc[j][0] + (float)p[j][0];
    if (r == 12345.678f) out[l] = r + lds[l];
}

template <int ROLE, bool LDSA>
float run(const bf16x8* in, float* out, int iters) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<ROLE, LDSA>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<ROLE, LDSA><<<256 * 4, 512, 100 * 1024>>>(in, out, iters);
    hipEventRecord(e0);
    k<ROLE, LDSA><<<256 * 4, 512, 100 * 1024>>>(in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    bf16x8* in;
    float* out;
    hipMalloc(&in, 128 * 16);
    hipMalloc(&out, 4096);
    hipMemset(in, 0x3c, 128 * 16);
    const int iters = 2000;
    const double cyc = 2.4e9 * 1e-3 / (4.0 * iters);   // 4 workgroups per CU in sequence
    for (int ldsa = 0; ldsa < 2; ++ldsa) {
    printf("MFMA A operands from %s\n", ldsa ? "LDS (ds_read_b128 per MFMA)" : "registers");
    const float t0 = ldsa ? run<0, true>(in, out, iters) : run<0, false>(in, out, iters), t1 = run<1, false>(in, out, iters),
                t2 = ldsa ? run<2, true>(in, out, iters) : run<2, false>(in, out, iters),
                t3 = ldsa ? run<3, true>(in, out, iters) : run<3, false>(in, out, iters),
                t4 = ldsa ? run<4, true>(in, out, iters) : run<4, false>(in, out, iters),
                t5 = ldsa ? run<5, true>(in, out, iters) : run<5, false>(in, out, iters),
                t6 = ldsa ? run<6, true>(in, out, iters) : run<6, false>(in, out, iters),
                t7 = ldsa ? run<7, true>(in, out, iters) : run<7, false>(in, out, iters);
    // per SIMD: 2 waves; roles 0/1: both waves run 2 tiles of one kind per iteration; roles 2/3/4: 2 matrix + 2 vector tiles
    printf("cycles per iteration and SIMD (2.4 GHz): matrix only (4 tiles) %.0f | vector only (4 tiles) %.0f\n"
           "  2 matrix + 2 vector tiles: waves 0-3 / 4-7 split %.0f | even / odd split %.0f | alternating in every wave %.0f\n"
           "  no overlap would be %.0f, perfect overlap %.0f\n"
           "  dependent halves + barrier after each: anti-phase %.0f | in phase %.0f | anti-phase, independent halves %.0f\n",
           t0 * cyc, t1 * cyc, t2 * cyc, t3 * cyc, t4 * cyc, (t0 + t1) * cyc / 2, (t0 > t1 ? t0 : t1) * cyc / 2, t5 * cyc,
           t6 * cyc, t7 * cyc);
    }
    return 0;
}
