// Microbenchmark: can VALU/TRANS work hide in the issue gaps of v_mfma_f32_32x32x16_bf16 on gfx950?
// Variants (template MODE): 0 = MFMA only, 1 = VALU only (fma + exp + cvt mix of the attention softmax),
// 2 = interleaved in program order: 1 MFMA : G VALU groups.  One or two waves per SIMD (block = 256 or 512).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NM, int NV>
__global__ void k(const bf16x8* a, float* out, int iters) {
    const int l = threadIdx.x;
    bf16x8 a0 = a[l & 63], b0 = a[64 + (l & 63)];
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = 0.001f * (l + i);
    const float c = 1.0001f, mc = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (MODE != 1) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[m & 3], 0, 0, 0);
            if (MODE != 0) {
#pragma unroll
                for (int u = (m * NV) / NM; u < ((m + 1) * NV) / NM; ++u) {   // one unit = 2 fma + 2 exp + 1 cvt-ish
                    const int i = (2 * u) & 31;
                    float p0 = __builtin_amdgcn_exp2f(fmaf(v[i], c, -mc));
                    float p1 = __builtin_amdgcn_exp2f(fmaf(v[i + 1], c, -mc));
                    __bf16 q0 = (__bf16)p0, q1 = (__bf16)p1;
                    v[i] = (float)q0 * 0.5f;
                    v[i + 1] = (float)q1 * 0.5f;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int i = 0; i < 32; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + l] = s;
}

template <int MODE, int NM, int NV>
float run(const bf16x8* a, float* out, int block, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NM, NV>), dim3(256), dim3(block), 0, 0, a, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NM, NV>), dim3(256), dim3(block), 0, 0, a, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    bf16x8* a;
    float* out;
    hipMalloc(&a, 128 * sizeof(bf16x8));
    hipMemset(a, 0x3c, 128 * sizeof(bf16x8));
    hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 20000;
    for (int block : {256, 512, 1024}) {
        float m = run<0, 14, 16>(a, out, block, iters);
        float v = run<1, 14, 16>(a, out, block, iters);
        float b = run<2, 14, 16>(a, out, block, iters);
        float v8 = run<1, 14, 8>(a, out, block, iters);
        float b8 = run<2, 14, 8>(a, out, block, iters);
        // cycles per iteration per SIMD at the nominal 2.4 GHz (waves/SIMD = block/256)
        printf("waves/SIMD %d: 14 MFMA %.3f ms | 16 units VALU %.3f ms | both %.3f ms (sum %.3f) | 8 units %.3f, both %.3f\n",
               block / 256, m, v, b, m + v, v8, b8);
    }
    return 0;
}
