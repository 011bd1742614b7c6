// Proxy of the level-0 NN-search inner loop (nn_search_rbg_kernel<.., 2>: a wave keeps 64 targets x 320 features as MFMA B
// fragments in registers, 32-pivot tiles stream through LDS as A fragments), stripped of staging, barriers and the argmax
// epilogue, run for hundreds of ms on toggling data (the boxes are power-limited).  Question (round 6, after the mixed
// MFMA shapes of the attention kernel): the bare v_mfma_f32_16x16x32_bf16 sustains 17 % more than v_mfma_f32_32x32x16_bf16
// on random data (profiles/r02_mfma_shapes.txt) -- does the search loop get any of that at equal matrix-pipe clocks?
//   SHAPE 0: per 32-pivot tile and 16-wide k-step: 1 ds_read_b128 (32 pivots x 16 k) -> 2 MFMAs 32x32x16 (two 32-target tiles)
//   SHAPE 1: per 32-pivot tile and 32-wide k-step: 2 ds_read_b128 (16 pivots x 32 k each) -> 8 MFMAs 16x16x32 (four 16-target tiles)
// Same LDS bytes, same matrix-pipe clocks (1280 per tile), same accumulator and B-fragment register counts.
//   hipcc --offload-arch=gfx950 -O3 -o nn_loop_proxy nn_loop_proxy.hip && ./nn_loop_proxy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const u32x4* __restrict__ src, float* out, int iters) {
    __shared__ u32x4 lds[2 * 20 * 64];   // two tile images of 20 fragments x 64 lanes x 16 B (32 pivots x 320 features)
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 20 * 64; i += 256) lds[i] = src[i & 2047];
    __syncthreads();
    bf16x8 fb[2][20];                     // 64 targets x 320 features: 160 VGPRs
    for (int t = 0; t < 20; ++t)
        for (int j = 0; j < 2; ++j) fb[j][t] = __builtin_bit_cast(bf16x8, src[(t * 128 + j * 64 + lane + tid) & 2047]);
    float best = -1e30f;
    for (int it = 0; it < iters; ++it) {
        int off = (it & 1) * 1280 + lane;
        asm volatile("" : "+v"(off));
        const u32x4* f = lds + off;
        if (SHAPE == 0) {
            f32x16 a0, a1;
            for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 20; ++t) {
                const bf16x8 fa = __builtin_bit_cast(bf16x8, f[t * 64]);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[0][t], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[1][t], a1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) best = fmaxf(best, fmaxf(a0[r], a1[r]));
        } else {
            f32x4 a[2][4];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 4; ++j) a[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 10; ++t) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16x8 fa = __builtin_bit_cast(bf16x8, f[(2 * t + i) * 64]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        a[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[j & 1][2 * t + (j >> 1)], a[i][j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) best = fmaxf(best, a[i][j][r]);
        }
    }
    out[blockIdx.x * 256 + tid] = best;
}

template <int SHAPE>
float run(const u32x4* src, float* out, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE>), dim3(512), dim3(256), 0, 0, src, out, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE>), dim3(512), dim3(256), 0, 0, src, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 60000;
    unsigned short* h = (unsigned short*)malloc(2048 * 16);
    unsigned x = 12345u;
    for (int i = 0; i < 2048 * 8; ++i) {
        x = x * 1664525u + 1013904223u;
        h[i] = (unsigned short)(((x >> 16) & 0x80ff) | 0x3f00);
    }
    u32x4* src;
    float* out;
    (void)hipMalloc(&src, 2048 * 16);
    (void)hipMemcpy(src, h, 2048 * 16, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 512 * 256 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        const float t0 = run<0>(src, out, iters), t1 = run<1>(src, out, iters);
        const double fl = (double)iters * 512 * 4 * 2.0 * 32 * 64 * 320;   // per wave and tile: 32 pivots x 64 targets x 320
        printf("2 waves/SIMD, %d tiles per wave: 32x32x16 %.1f ms = %.0f TF/s | 16x16x32 %.1f ms = %.0f TF/s\n", iters, t0,
               fl / t0 / 1e9, t1, fl / t1 / 1e9);
    }
    return 0;
}
