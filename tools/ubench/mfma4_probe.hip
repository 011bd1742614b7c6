// Probe (round 6): v_mfma_f32_4x4x4_16b_bf16 as a lane-local row-sum engine for the softmax denominator at head dims
// without a spare P.V row (Dh = 64).  With A = all ones, D[i][j] = sum_k B[k][j]: does every lane get the sum of ITS OWN
// four B values (block = lane / 4, column = lane % 4 for B and for D alike)?  And what does one such MFMA cost -- alone,
// and slipped between v_mfma_f32_32x32x16_bf16 -- against the two v_add_f32 per pair it would replace?
//   hipcc --offload-arch=gfx950 -O3 -o mfma4_probe mfma4_probe.hip && ./mfma4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void semantics(const bf16x4* b, f32x4* out) {
    const int l = threadIdx.x;
    const bf16x4 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, ones), __builtin_bit_cast(s16x4, b[l]), acc, 0, 0, 0);
    out[l] = acc;
}

// MODE 0: 16 big MFMAs per iteration (4 accumulators)            -- the matrix work of one 32 x 64 tile at Dh = 64
// MODE 1: + 8 small MFMAs (one accumulator chain, every other gap) -- the row sum on the matrix pipe
// MODE 2: + 32 v_add_f32 (two chains)                              -- the row sum on the VALU
// MODE 3: 8 small MFMAs alone; MODE 4: 32 v_add alone
template <int MODE>
__global__ void cost(const bf16x8* a, float* out, int iters) {
    const int l = threadIdx.x & 63;
    bf16x8 a0 = a[l], b0 = a[64 + l];
    const bf16x4 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
    const s16x4 o4 = __builtin_bit_cast(s16x4, ones);
    s16x4 p4 = __builtin_bit_cast(s16x4, a[l].lo);
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f32x4 ls = {0.f, 0.f, 0.f, 0.f};
    float s0 = 0.f, s1 = 0.f, x = out[l];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (MODE <= 2) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[m & 3], 0, 0, 0);
            if ((MODE == 1 || MODE == 3) && (m & 1)) ls = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(o4, p4, ls, 0, 0, 0);
            if (MODE == 2 || MODE == 4) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(s0) : "v"(x));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(s1) : "v"(x));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = s0 + s1 + ls[0] + ls[1] + ls[2] + ls[3];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(const bf16x8* a, float* out, int block, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((cost<MODE>), dim3(256), dim3(block), 0, 0, a, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((cost<MODE>), dim3(256), dim3(block), 0, 0, a, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    // ---- semantics
    unsigned short hb[64 * 4];
    float want[64];
    for (int l = 0; l < 64; ++l) {
        want[l] = 0.f;
        for (int k = 0; k < 4; ++k) {
            const float v = (float)((l * 7 + k * 3) % 16) + 0.5f * (k & 1);   // exactly representable in bf16
            unsigned u;
            memcpy(&u, &v, 4);
            hb[l * 4 + k] = (unsigned short)(u >> 16);
            want[l] += v;
        }
    }
    bf16x4* db;
    f32x4* dout;
    hipMalloc(&db, sizeof(hb));
    hipMalloc(&dout, 64 * sizeof(f32x4));
    hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, db, dout);
    float ho[64 * 4];
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) bad += ho[l * 4 + r] != want[l];
    printf("semantics: A = ones -> every lane holds the sum of its own 4 B values in all 4 result registers: %s (%d of 256 "
           "values differ; lane 5: got %.1f %.1f %.1f %.1f want %.1f)\n",
           bad ? "NO" : "yes", bad, ho[20], ho[21], ho[22], ho[23], want[5]);
    // ---- cost
    bf16x8* a;
    float* out;
    hipMalloc(&a, 128 * sizeof(bf16x8));
    unsigned short h[128 * 8];
    unsigned x = 12345u;
    for (int i = 0; i < 128 * 8; ++i) {
        x = x * 1664525u + 1013904223u;
        h[i] = (unsigned short)(((x >> 16) & 0x80ff) | 0x3f00);
    }
    hipMemcpy(a, h, sizeof(h), hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 1024 * 4);
    hipMemset(out, 0, 256 * 1024 * 4);
    const int iters = 100000;
    const char* names[5] = {"16 x mfma 32x32x16", "+ 8 x mfma 4x4x4 (row sum on the matrix pipe)", "+ 32 x v_add_f32 (row sum on the VALU)",
                            "8 x mfma 4x4x4 alone", "32 x v_add_f32 alone"};
    for (int block : {256, 512, 1024}) {
        const float t[5] = {run<0>(a, out, block, iters), run<1>(a, out, block, iters), run<2>(a, out, block, iters),
                            run<3>(a, out, block, iters), run<4>(a, out, block, iters)};
        for (int s = 0; s < 5; ++s)
            printf("waves/SIMD %d  %-48s %8.3f ms  %7.1f ns per tile-equivalent and SIMD\n", block / 256, names[s], t[s],
                   t[s] * 1e6 / ((double)iters * (block / 256)));
    }
    return 0;
}
