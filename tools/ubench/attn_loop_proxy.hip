// Proxy of the level-0 attention inner loop (head dim 40) in the structure VERDICT r03 item 2 asks to be tested:
// ONE wave per SIMD (4-wave workgroups, one per CU), every wave owning TWO independent 32-query blocks A and B, the
// softmax VALU of one block placed -- instruction by instruction -- in the issue gaps of the other block's MFMAs:
//     region 1 of tile t:  O_B += V(t-1) P_B(t-1), S_B(t) = K(t) Q_B   (8 + 6 MFMAs)  ||  P_A(t) = exp2(S_A(t) c - m)
//     region 2 of tile t:  O_A += V(t)   P_A(t),   S_A(t+1) = K(t+1) Q_A              ||  P_B(t) = exp2(S_B(t) c - m)
// Per MFMA slot: the MFMA, one ds_read_b128 (the fragment of the MFMA PF slots ahead) and one softmax unit
// {2 multiply-adds, 2 v_exp_f32, 1 v_cvt_pk_bf16_f32} (two slots of a region carry two units: 16 units per 14 MFMAs),
// i.e. 6-7 single-issue fillers per 32-cycle MFMA gap (MI355X_MICROARCH.md: <= 5 hide completely).
// Stripped exactly like tools/ubench/attn_tile_proxy.hip (the 4-waves-per-SIMD, compiler-scheduled yardstick: 865-881
// TF/s): no global loads, LDS staging writes, barriers, running maxima or rescales; two LDS images alternate so the
// MFMA operands toggle.  Same fragment count per query tile (14 ds_read_b128 per 32 x 64 tile), same softmax mix.
//   VAR 0  one __builtin_amdgcn_sched_barrier(0) after EVERY instruction of a slot: the program order below is the issue
//          order (hand placement); scalar v_fma_f32
//   VAR 1  the same with every v_cvt_pk deferred by one unit (no wait state behind the transcendental)
//   VAR 2  sched_barrier only between slots: the compiler orders the 7 instructions of a slot itself
//   VAR 3  VAR 0 with the fragment reads 5 slots ahead instead of 3
//   VAR 9  the yardstick loop of attn_tile_proxy.hip (SHAPE 0) in the same binary, 4 waves per SIMD
// Kill criterion: the best of VAR 0-3 must beat VAR 9 by >= 8 %, else the level-0 kernel stays as it is.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o attn_loop_proxy attn_loop_proxy.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2 r = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, r);
}

// MFMA order of a region: P.V (2 M-tiles x 4 k-steps) and QK^T (2 key halves x 3 k-steps), round-robin over the four
// accumulators so that consecutive MFMAs never share one.  Fragment f of the LDS image: 0-5 K, 6-13 V^T.
struct Slot {
    int pv, acc, kstep, frag;
};
constexpr Slot slot_of(int i) {
    // i = 0..13: o0 s0 o1 s1 | o0 s0 o1 s1 | o0 s0 o1 s1 | o0 o1
    if (i < 12) {
        const int r = i / 4, w = i % 4;
        if (w == 0) return {1, 0, r, 6 + 2 * r};
        if (w == 1) return {0, 0, r, 2 * r};
        if (w == 2) return {1, 1, r, 7 + 2 * r};
        return {0, 1, r, 2 * r + 1};
    }
    return {1, i - 12, 3, 12 + (i - 12)};
}

template <int VAR>
__global__ __launch_bounds__(256, 1) void k1(const u32x4* __restrict__ src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 images x 14 fragments x 64 lanes x 16 B
    u32x4* lds = reinterpret_cast<u32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 16 * 64; i += 256) lds[i] = src[i];
    __syncthreads();
    const float c = 0.11f, m = 8.f;
    constexpr int PF = VAR == 3 ? 5 : 3;
    bf16x8 q[2][3];
    for (int b = 0; b < 2; ++b)
        for (int t = 0; t < 3; ++t) q[b][t] = __builtin_bit_cast(bf16x8, src[(b * 192 + t * 64 + lane + tid) & 2047]);
    f32x16 o[2][2], s[2][2];
    u32x4 p[2][4];
    for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 2; ++a)
            for (int r = 0; r < 16; ++r) o[b][a][r] = 0.f, s[b][a][r] = (float)(lane + r + a) * 0.01f;
    for (int b = 0; b < 2; ++b)
        for (int g = 0; g < 4; ++g) p[b][g] = src[(b * 4 + g) * 64 + lane];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // one softmax unit of block X: scores (half kt, registers r, r + 1) -> one packed pair of P.
    // VAR 1 defers the conversion by one unit: {fma, fma, exp, exp} of unit un, then the v_cvt_pk of unit un - 1, so that
    // no instruction waits on the transcendental issued right before it (VAR 0 carries an s_nop per unit there).
    float pe0 = 0.f, pe1 = 0.f;
    auto unit = [&](auto x_c, int un, unsigned (&w)[16]) {
        constexpr int X = decltype(x_c)::value;
        const int kt = un >> 3, r = (un & 7) * 2;
        float x0 = fmaf(s[X][kt][r], c, -m);
        asm volatile("" : "+v"(x0));   // keeps the two multiply-adds scalar (no SLP packing into v_pk_fma_f32)
        if (VAR != 2) SB();
        float x1 = fmaf(s[X][kt][r + 1], c, -m);
        asm volatile("" : "+v"(x1));
        if (VAR != 2) SB();
        // the empty asm statements pin every result where it is computed (LLVM otherwise sinks the exponentials and the
        // conversions to their first use, behind the region's last MFMA)
        float e0 = __builtin_amdgcn_exp2f(x0);
        asm volatile("" : "+v"(e0));
        if (VAR != 2) SB();
        float e1 = __builtin_amdgcn_exp2f(x1);
        asm volatile("" : "+v"(e1));
        if (VAR != 2) SB();
        if constexpr (VAR == 1) {
            if (un > 0) {
                w[un - 1] = pack2(pe0, pe1);
                asm volatile("" : "+v"(w[un - 1]));
                SB();
            }
            pe0 = e0, pe1 = e1;
            if (un == 15) {
                w[15] = pack2(pe0, pe1);
                asm volatile("" : "+v"(w[15]));
                SB();
            }
        } else {
            w[un] = pack2(e0, e1);
            asm volatile("" : "+v"(w[un]));
            if (VAR != 2) SB();
        }
    };
    // a region: the 14 MFMAs of block Y beside the 16 softmax units of block X = 1 - Y
    auto region = [&](auto x_c, const u32x4* f) {
        constexpr int X = decltype(x_c)::value, Y = 1 - X;
        u32x4 fr[14];
        unsigned w[16];
#pragma unroll
        for (int i = 0; i < PF; ++i) fr[i] = f[slot_of(i).frag * 64];
        SB();
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            constexpr int dummy = 0;
            (void)dummy;
            const Slot sl = slot_of(i);
            if (sl.pv)
                o[Y][sl.acc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[i]),
                                                                     __builtin_bit_cast(bf16x8, p[Y][sl.kstep]), o[Y][sl.acc], 0, 0, 0);
            else
                s[Y][sl.acc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[i]), q[Y][sl.kstep],
                                                                     sl.kstep == 0 ? zero : s[Y][sl.acc], 0, 0, 0);
            if (VAR != 2) SB();
            if (i + PF < 14) {
                fr[i + PF] = f[slot_of(i + PF).frag * 64];
                if (VAR != 2) SB();
            }
#pragma unroll
            for (int un = (i * 16) / 14; un < ((i + 1) * 16) / 14; ++un) unit(x_c, un, w);
            SB();
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) p[X][g] = u32x4{w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]};
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(p[X][g]));
    };
    typedef std::integral_constant<int, 0> A;
    typedef std::integral_constant<int, 1> B;
    for (int it = 0; it < iters; ++it) {
        int off = (it & 1) * 1024 + lane;
        asm volatile("" : "+v"(off));
        const u32x4* f = lds + off;
        region(A{}, f);   // MFMAs of B  ||  softmax of A
        region(B{}, f);   // MFMAs of A  ||  softmax of B
    }
    float sm = 0.f;
    for (int b = 0; b < 2; ++b)
        for (int a = 0; a < 2; ++a)
            for (int r = 0; r < 16; ++r) sm += o[b][a][r] + s[b][a][r];
    out[blockIdx.x * 256 + tid] = sm;
}

// the yardstick: tools/ubench/attn_tile_proxy.hip SHAPE 0 (one query block per wave, 4 waves per SIMD, compiler-scheduled)
__global__ __launch_bounds__(512) void k9(const u32x4* __restrict__ src, float* out, int iters) {
    __shared__ u32x4 lds[2 * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 16 * 64; i += 512) lds[i] = src[i];
    __syncthreads();
    const float c = 0.11f, m = 8.f;
    bf16x8 q[4];
    for (int t = 0; t < 4; ++t) q[t] = __builtin_bit_cast(bf16x8, src[(t * 64 + lane + tid) & 2047]);
    f32x16 o0, o1;
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        int off = (it & 1) * 1024 + lane;
        asm volatile("" : "+v"(off));
        const u32x4* f = lds + off;
        f32x16 s0, s1;
        for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[(2 * t) * 64]), q[t], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[(2 * t + 1) * 64]), q[t], s1, 0, 0, 0);
        }
        u32x4 p[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            unsigned w[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int r = 4 * (g & 1) * 2 + 2 * (h & 1) + 4 * (h >> 1);
                const f32x16& s = g < 2 ? s0 : s1;
                const f32x2 x = f32x2{s[r], s[r + 1]} * c - m;
                w[h] = pack2(__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1]));
            }
            p[g] = u32x4{w[0], w[1], w[2], w[3]};
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[(6 + 2 * g) * 64]),
                                                         __builtin_bit_cast(bf16x8, p[g]), o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[(7 + 2 * g) * 64]),
                                                         __builtin_bit_cast(bf16x8, p[g]), o1, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += o0[r] + o1[r];
    out[blockIdx.x * 512 + tid] = s;
}

template <typename F>
float timed(F launch, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch(100);
    (void)hipEventRecord(e0);
    launch(iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;   // 64-key tiles per QUERY BLOCK
    unsigned short* h = (unsigned short*)malloc(2048 * 16);
    unsigned x = 12345u;
    for (int i = 0; i < 2048 * 8; ++i) {
        x = x * 1664525u + 1013904223u;
        h[i] = (unsigned short)(((x >> 16) & 0x80ff) | 0x3f00);   // bf16 in +-[0.5, 2)
    }
    u32x4* src;
    float* out;
    (void)hipMalloc(&src, 2048 * 16);
    (void)hipMemcpy(src, h, 2048 * 16, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 512 * 512 * 4);
    // one-wave-per-SIMD variants: 256 workgroups x 4 waves x 2 query blocks; 100 KB of LDS keep a second workgroup off the CU
    const size_t lds1 = 100 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    const double fl_tile = 2.0 * 2 * 32 * 64 * 40;   // algorithmic flops of one 32-query x 64-key tile at head dim 40
    for (int rep = 0; rep < 2; ++rep) {
        // same number of (query block, tile) pairs in every variant: the yardstick runs 512 x 8 waves x iters tiles
        const double tiles = (double)iters * 512 * 8;
        const int it1 = iters * 2;   // 256 x 4 waves x 2 blocks x it1 = 512 x 8 x iters
        const float t9 = timed([&](int n) { hipLaunchKernelGGL(k9, dim3(512), dim3(512), 0, 0, src, out, n); }, iters);
        const float t0 = timed([&](int n) { hipLaunchKernelGGL(k1<0>, dim3(256), dim3(256), lds1, 0, src, out, n); }, it1);
        const float t1 = timed([&](int n) { hipLaunchKernelGGL(k1<1>, dim3(256), dim3(256), lds1, 0, src, out, n); }, it1);
        const float t2 = timed([&](int n) { hipLaunchKernelGGL(k1<2>, dim3(256), dim3(256), lds1, 0, src, out, n); }, it1);
        const float t3 = timed([&](int n) { hipLaunchKernelGGL(k1<3>, dim3(256), dim3(256), lds1, 0, src, out, n); }, it1);
        printf("yardstick (4 waves/SIMD, compiler-scheduled) %.1f ms = %.0f TF/s | 1 wave/SIMD, 2 query blocks, placed: "
               "instruction-pinned %.1f ms = %.0f TF/s | + deferred cvt %.1f ms = %.0f TF/s | per-slot pinning only %.1f ms = %.0f TF/s | "
               "reads 5 ahead %.1f ms = %.0f TF/s\n",
               t9, tiles * fl_tile / t9 / 1e9, t0, tiles * fl_tile / t0 / 1e9, t1, tiles * fl_tile / t1 / 1e9, t2,
               tiles * fl_tile / t2 / 1e9, t3, tiles * fl_tile / t3 / 1e9);
    }
    return 0;
}
