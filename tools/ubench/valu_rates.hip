// Microbenchmark: issue cost of the vector instructions of the attention softmax on gfx950, relative to v_fma_f32:
// v_exp_f32, v_exp_f16, v_pk_fma_f32, v_pk_mul_f32, v_cvt_pk_bf16_f32, v_max3_f32, v_permlane32_swap.
// (question behind it: is a half-precision exponential cheaper than v_exp_f32?  the softmax at head dim 40 is
// bound by 32 exponentials per lane and tile.)
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void k(float* out, int iters) {
    float v[8];
    float2 p[8];
    for (int i = 0; i < 8; ++i) {
        v[i] = -0.001f * (threadIdx.x + i);
        p[i] = make_float2(v[i], v[i] * 0.5f);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#define ONE(i)                                                                                          \
    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));                                 \
    if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));                                         \
    if (OP == 2) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));                                         \
    if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));                              \
    if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));                                  \
    if (OP == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v[i]));                             \
    if (OP == 6) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(v[i]));                                \
    if (OP == 7) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[(i + 1) & 7]));           \
    if (OP == 8) asm volatile("v_log_f32 %0, %0" : "+v"(v[i]));                                         \
    if (OP == 9) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
            REP8(ONE)
#undef ONE
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
float run(float* out, int block, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(block), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(block), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    const char* names[10] = {"v_fma_f32", "v_exp_f32", "v_exp_f16", "v_pk_fma_f32", "v_pk_mul_f32",
                             "v_cvt_pk_bf16_f32", "v_max3_f32", "v_permlane32_swap", "v_log_f32", "v_rcp_f32"};
    const int iters = 20000;
    for (int block : {256, 512, 1024}) {   // 1, 2, 4 waves per SIMD (one workgroup per CU)
        float ms[10];
        ms[0] = run<0>(out, block, iters);
        ms[1] = run<1>(out, block, iters);
        ms[2] = run<2>(out, block, iters);
        ms[3] = run<3>(out, block, iters);
        ms[4] = run<4>(out, block, iters);
        ms[5] = run<5>(out, block, iters);
        ms[6] = run<6>(out, block, iters);
        ms[7] = run<7>(out, block, iters);
        ms[8] = run<8>(out, block, iters);
        ms[9] = run<9>(out, block, iters);
        const double insts = (double)iters * 32 * (block / 256);   // wave-instructions per SIMD
        for (int i = 0; i < 10; ++i)
            printf("%d waves/SIMD  %-20s %7.3f ns per wave-instruction and SIMD  (x%.2f of v_fma_f32)\n", block / 256,
                   names[i], ms[i] * 1e6 / insts, ms[i] / ms[0]);
    }
    return 0;
}
