// One-wave probe that measures the shader clock WHILE other kernels run: it spins for `ms` milliseconds of the
// constant 100 MHz counter (s_memrealtime) and reports how many shader-clock ticks (s_memtime) went by.
// Launched on a side stream before the workload; tools/clock_probe.py drives it.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libclock_probe.so clock_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void probe_kernel(uint64_t* out, uint64_t ticks100mhz) {
    const uint64_t r0 = wall_clock64();
    const uint64_t c0 = clock64();
    uint64_t r1 = r0;
    while (r1 - r0 < ticks100mhz) {
        __builtin_amdgcn_s_sleep(32);
        r1 = wall_clock64();
    }
    const uint64_t c1 = clock64();
    out[0] = r1 - r0;
    out[1] = c1 - c0;
}

extern "C" int clock_probe_launch(uint64_t* out, double ms, void* stream) {
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), out,
                       (uint64_t)(ms * 1e5));
    return (int)hipGetLastError();
}
