// Which ingredient of the attention inner loop stops MFMA / VALU overlap inside one wave?
// Base loop: 14 MFMAs interleaved 1 : ~7 with the softmax VALU mix (fma, exp2, cvt) -- cf. mfma_valu_overlap.hip.
// FLAGS bit 0: MFMA A operands come from LDS (ds_read_b128, 4 steps ahead) instead of registers
//       bit 1: the VALU results (rounded P) are the B operands of the NEXT iteration's MFMAs
//       bit 2: the MFMA results (S) are the inputs of the NEXT iteration's VALU work
//       bit 3: a workgroup barrier per iteration
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int FLAGS>
__global__ __launch_bounds__(256) void k(const bf16x8* a, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[16 * 1024];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 256) reinterpret_cast<u32x4*>(lds)[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    bf16x8 a0 = a[l & 63], b0 = a[64 + (l & 63)];
    bf16x8 pb[4] = {b0, b0, b0, b0};
    f32x16 acc[4], s[2];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int j = 0; j < 2; ++j)
        for (int r = 0; r < 16; ++r) s[j][r] = 0.001f * (l + r);
    const float c = 1.0001f, mc = 0.5f;
    const unsigned char* base = lds + (l & 63) * 16;
    for (int it = 0; it < iters; ++it) {
        bf16x8 fr[14];
        if (FLAGS & 1)
            for (int i = 0; i < 4; ++i) fr[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + i * 1024));
        asm volatile("" : "+v"(s[0]), "+v"(s[1]));   // opaque: the VALU work cannot be hoisted out of the loop
        bf16x8 pn[4];
        f32x16 sn[2] = {s[0], s[1]};
#pragma unroll
        for (int m = 0; m < 14; ++m) {
            if (MODE != 1) {
                if ((FLAGS & 1) && m + 4 < 14)
                    fr[m + 4] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + ((m + 4) & 15) * 1024));
                const bf16x8 av = (FLAGS & 1) ? fr[m] : a0;
                const bf16x8 bv = (FLAGS & 2) ? pb[m & 3] : b0;
                if ((FLAGS & 4) && m >= 8)
                    sn[(m - 8) / 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, sn[(m - 8) / 3], 0, 0, 0);
                else
                    acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[m & 3], 0, 0, 0);
            }
            if (MODE != 0) {
#pragma unroll
                for (int u = (m * 16) / 14; u < ((m + 1) * 16) / 14; ++u) {   // unit = 2 fma + 2 exp + 1 cvt_pk
                    const int kt = u >> 3, r = (u & 7) * 2;
                    const float p0 = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c, -mc));
                    const float p1 = __builtin_amdgcn_exp2f(fmaf(s[kt][r + 1], c, -mc));
                    pn[kt * 2 + (r >> 3)][r & 7] = (__bf16)p0;
                    pn[kt * 2 + (r >> 3)][(r & 7) + 1] = (__bf16)p1;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE != 0) {
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(pn[j]));
            if (FLAGS & 2) for (int j = 0; j < 4; ++j) pb[j] = pn[j];
        }
        if (FLAGS & 4) { s[0] = sn[0]; s[1] = sn[1]; }
        if (FLAGS & 8) __syncthreads();
    }
    float t = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) t += acc[j][r];
    for (int j = 0; j < 2; ++j)
        for (int r = 0; r < 16; ++r) t += s[j][r];
    for (int j = 0; j < 4; ++j) t += (float)pb[j][0];
    out[blockIdx.x * blockDim.x + l] = t;
}

template <int MODE, int FLAGS>
float run(const bf16x8* a, float* out, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, FLAGS>), dim3(blocks), dim3(256), 0, 0, a, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, FLAGS>), dim3(blocks), dim3(256), 0, 0, a, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int FLAGS>
void report(const bf16x8* a, float* out, int blocks, const char* what) {
    const int iters = 20000;
    const float m = run<0, FLAGS>(a, out, blocks, iters), v = run<1, FLAGS>(a, out, blocks, iters),
                b = run<2, FLAGS>(a, out, blocks, iters);
    printf("%-44s blocks/CU %d: MFMA %.2f  VALU %.2f  both %.2f ms  (max %.2f, sum %.2f)\n", what, blocks / 256, m, v, b,
           m > v ? m : v, m + v);
}

int main() {
    bf16x8* a;
    float* out;
    hipMalloc(&a, 128 * sizeof(bf16x8));
    hipMemset(a, 0x3c, 128 * sizeof(bf16x8));
    hipMalloc(&out, 1024 * 1024 * 4);
    for (int blocks : {256, 512}) {
        report<0>(a, out, blocks, "registers only");
        report<1>(a, out, blocks, "+ A fragments from LDS");
        report<2>(a, out, blocks, "+ P (VALU) -> next MFMAs");
        report<4>(a, out, blocks, "+ S (MFMA) -> next VALU");
        report<7>(a, out, blocks, "LDS + P->MFMA + S->VALU");
        report<15>(a, out, blocks, "all + barrier");
    }
    return 0;
}
