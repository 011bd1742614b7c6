// Microbenchmark: issue cost of the bf16 MFMA shapes on gfx950, 1..2 waves per SIMD, 4 independent accumulators:
// v_mfma_f32_32x32x16_bf16 (gfx950), v_mfma_f32_16x16x32_bf16 (gfx950), and the gfx90a-era forms
// v_mfma_f32_32x32x8_bf16_1k / v_mfma_f32_16x16x16_bf16_1k -- is a half-K legacy MFMA half the cycles?
// (question behind it: QK^T at head dim 40 pads K to 48 = 3 x 16; with a half-cost K = 8 / 16 step the pad would shrink.)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shapes mfma_shapes.hip && ./mfma_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ void k(const bf16x8* a, float* out, int iters) {
    const int l = threadIdx.x;
    bf16x8 a0 = a[l & 63], b0 = a[64 + (l & 63)];
    s16x4 a4 = __builtin_bit_cast(s16x4, a[l & 63].lo), b4 = __builtin_bit_cast(s16x4, a[64 + (l & 63)].lo);
    f32x16 acc[4];
    f32x4 acc4[4];
    for (int j = 0; j < 4; ++j) {
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int r = 0; r < 4; ++r) acc4[j][r] = 0.f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (SHAPE == 0) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[m & 3], 0, 0, 0);
            if (SHAPE == 1) acc4[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc4[m & 3], 0, 0, 0);
            if (SHAPE == 2) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc[m & 3], 0, 0, 0);
            if (SHAPE == 3) acc4[m & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc4[m & 3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) {
        for (int r = 0; r < 16; ++r) s += acc[j][r];
        for (int r = 0; r < 4; ++r) s += acc4[j][r];
    }
    out[blockIdx.x * blockDim.x + l] = s;
}

template <int SHAPE>
float run(const bf16x8* a, float* out, int block, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE>), dim3(256), dim3(block), 0, 0, a, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE>), dim3(256), dim3(block), 0, 0, a, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

// `./mfma_shapes rand`: operands = pseudo-random bf16 in (-2, 2) instead of one repeated value -- the same
// instruction stream on toggling data shows how far the board's power limit pulls the clock (and the rate) down.
int main(int argc, char** argv) {
    bf16x8* a;
    float* out;
    hipMalloc(&a, 128 * sizeof(bf16x8));
    hipMemset(a, 0x3c, 128 * sizeof(bf16x8));
    if (argc > 1) {
        unsigned short h[128 * 8];
        unsigned x = 12345u;
        for (int i = 0; i < 128 * 8; ++i) {
            x = x * 1664525u + 1013904223u;
            h[i] = (unsigned short)(((x >> 16) & 0x80ff) | 0x3f00);   // sign, exponent 126/127, random mantissa
        }
        hipMemcpy(a, h, sizeof(h), hipMemcpyHostToDevice);
        printf("operands: pseudo-random bf16\n");
    }
    hipMalloc(&out, 256 * 1024 * 4);
    const int iters = argc > 1 ? 200000 : 20000;
    const char* names[4] = {"32x32x16", "16x16x32", "32x32x8_1k", "16x16x16_1k"};
    const double macs[4] = {32. * 32 * 16, 16. * 16 * 32, 32. * 32 * 8, 16. * 16 * 16};
    for (int block : {256, 512}) {
        float t[4] = {run<0>(a, out, block, iters), run<1>(a, out, block, iters), run<2>(a, out, block, iters),
                      run<3>(a, out, block, iters)};
        for (int s = 0; s < 4; ++s) {
            const double per_simd = (double)iters * 16 * (block / 256);          // MFMAs per SIMD
            printf("waves/SIMD %d  %-12s %.3f ms  %.1f ns/MFMA/SIMD  %.0f TFLOP/s chip\n", block / 256, names[s], t[s],
                   t[s] * 1e6 / per_simd, 2 * macs[s] * per_simd * 1024 / (t[s] * 1e-3) / 1e12);
        }
    }
    return 0;
}
