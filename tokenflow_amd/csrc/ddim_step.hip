// DDIM latent update of the inversion / reconstruction loops that produce and consume the latents directory --
// row f4 of SURVEY.md section 8, the step BEFORE the hot path.  Replaces preprocess.py:224-225 (ddim_inversion) and
// 259-260 (ddim_sample) of omerbt/TokenFlow:
//     pred_x0 = (x - sigma_a * eps) / mu_a
//     x'      = mu_b * pred_x0 + sigma_b * eps
// Six elementwise torch ops on the [F,4,H/8,W/8] latents per UNet call in the reference; one pass here.
// Arithmetic follows the reference op by op: every intermediate is rounded to the tensor dtype (torch evaluates a
// 16-bit op in fp32 and rounds its result; the four coefficients are fp32 scalars, as they are on the reference's
// CUDA path where scheduler.alphas_cumprod lives on the host), no fused multiply-add, IEEE division -- so the
// result is bit-identical to the reference's sequence AS TORCH EVALUATES IT ON THE CPU (what the golden fixture
// tests/golden/inversion.pt pins) in fp32, f16 and bf16.  On a GPU torch's true-divide by a host scalar takes a
// fast path, x * (1 / mu) with the reciprocal formed once in fp32, so there pred_x0 can differ from this kernel's
// IEEE quotient by one rounding of the tensor dtype on a small fraction of the elements
// (tests/test_kernels_gpu.py::test_ddim_step_vs_torch_gpu_sequence bounds it).
#include "tf_common.h"

#pragma clang fp contract(off)

namespace {

template <typename T>
__device__ __forceinline__ float rnd(float x) {
    if constexpr (sizeof(T) == 4) return x;
    return (float)(T)x;
}

template <typename T>
__global__ __launch_bounds__(256) void ddim_step_kernel(const T* x, const T* __restrict__ eps,   // out may alias x:
                                                        T* out, int64_t n, float mu_a, float sigma_a,   // no restrict
                                                        float mu_b, float sigma_b) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float xe = (float)x[i], e = (float)eps[i];
        const float t1 = rnd<T>(__fmul_rn(sigma_a, e));       // sigma_prev * eps
        const float t2 = rnd<T>(__fsub_rn(xe, t1));           // x - ...
        const float p0 = rnd<T>(__fdiv_rn(t2, mu_a));         // / mu_prev            -> pred_x0
        const float t3 = rnd<T>(__fmul_rn(mu_b, p0));         // mu * pred_x0
        const float t4 = rnd<T>(__fmul_rn(sigma_b, e));       // sigma * eps
        out[i] = (T)__fadd_rn(t3, t4);
    }
}

}  // namespace

extern "C" int tf_ddim_step(const void* x, const void* eps, void* out, int64_t n, float mu_a, float sigma_a,
                            float mu_b, float sigma_b, int dtype, void* stream) {
    TF_ARG(x && eps && out, TF_ERR_NULL, "tf_ddim_step: null pointer");
    TF_ARG(dtype == TF_BF16 || dtype == TF_F16 || dtype == TF_F32, TF_ERR_DTYPE, "tf_ddim_step: dtype %d", dtype);
    TF_ARG(n > 0 && mu_a != 0.f, TF_ERR_SHAPE, "tf_ddim_step: n=%lld mu_a=%g", (long long)n, (double)mu_a);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    switch (dtype) {
        case TF_BF16:
            hipLaunchKernelGGL(ddim_step_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, st, (const __bf16*)x,
                               (const __bf16*)eps, (__bf16*)out, n, mu_a, sigma_a, mu_b, sigma_b);
            break;
        case TF_F16:
            hipLaunchKernelGGL(ddim_step_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, st, (const _Float16*)x,
                               (const _Float16*)eps, (_Float16*)out, n, mu_a, sigma_a, mu_b, sigma_b);
            break;
        default:
            hipLaunchKernelGGL(ddim_step_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x,
                               (const float*)eps, (float*)out, n, mu_a, sigma_a, mu_b, sigma_b);
            break;
    }
    TF_LAUNCH_CHECK("tf_ddim_step");
    return 0;
}
