// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read) -- prints the lane/element -> LDS element map the hardware
// implements and checks it against the model csrc/ext_attn_fused.hip is written for:
//   within a 16-lane group, lane i passes the address of block[i >> 2][4 * (i & 3) .. +3] of a [4][16] block of 16-bit
//   elements and receives block[0..3][i].
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_probe.hip -o tools/ubench/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __bf16 bf16x4_vs __attribute__((__vector_size__(8)));
#define LDS_AS __attribute__((address_space(3)))

// LDS image: 64 rows x ROWS elements, element value = its own index (exact in 16 bits as a raw pattern)
__global__ void probe(uint16_t* out, int row_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* s = reinterpret_cast<uint16_t*>(smem);
    for (int i = threadIdx.x; i < 64 * row_stride; i += 64) s[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4, hi = l >> 5;
    // the fused kernel's address for (ks = 0, mt = 0): rows 4*hi + (i >> 2), columns 16*(g & 1) + 4*(i & 3)
    const uint16_t* p = s + (4 * hi + (i >> 2)) * row_stride + 16 * (g & 1) + 4 * (i & 3);
    bf16x4_vs r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4_vs*)(p));
    uint16_t v[4];
    __builtin_memcpy(v, &r, 8);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

int main() {
    int bad_total = 0;
    for (int rs : {160, 96}) {
        uint16_t* d;
        hipMalloc(&d, 256 * 2);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 64 * rs * 2, 0, d, rs);
        std::vector<uint16_t> h(256);
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            const int i = l & 15, g = l >> 4, hi = l >> 5;
            for (int j = 0; j < 4; ++j) {
                // model: element j of lane l = V[key = 4*hi + j][feature = 16*(g & 1) + i]
                const int want = (4 * hi + j) * rs + 16 * (g & 1) + i;
                if (h[l * 4 + j] != want) {
                    if (bad < 8) printf("row_stride %d lane %2d elem %d: got LDS element %5d (row %d col %d), model %5d\n", rs, l, j,
                                        h[l * 4 + j], h[l * 4 + j] / rs, h[l * 4 + j] % rs, want);
                    ++bad;
                }
            }
        }
        printf("ds_read_b64_tr_b16 row stride %d elements: %s (%d of 256 differ from the model)\n", rs, bad ? "MODEL WRONG" : "model confirmed", bad);
        bad_total += bad;
        hipFree(d);
    }
    return bad_total ? 1 : 0;
}
