// Extended (cross-keyframe) attention forward for gfx950.
// Replaces tokenflow_utils.py:124-197 / 234-279 of omerbt/TokenFlow: the per-head
// bmm -> *scale -> softmax -> bmm loops over a K-times replicated key/value bank.
//
// Flash-style: one workgroup = 128..256 queries of one (branch, frame, head); it streams the
// key/value sequence (S keys for the source branch, the K*S-key bank of the branch for
// uncond / cond) in 64-key tiles with an online softmax; nothing of size S x K*S exists.
// q/k are read in place from the [3,K,S,H*Dh] projection output (head = a Dh-wide column
// slab, token stride ld); PnP injection is pointer aliasing of the source branch's q/k.
//
// MFMA mapping (v_mfma_f32_32x32x16, 64-lane waves, one wave = 32 queries):
//   S^T = K . Q^T   A = K tile rows (keys) from LDS, B = Q fragments held in registers
//                   -> lane owns ONE query (col = lane & 31) and 16 keys per 32-key tile:
//                      softmax statistics are lane-local (+1 exchange with lane ^ 32).
//   O^T = V^T . P   A = V^T rows (d) from LDS, B = P straight from the S^T accumulator
//                   registers: C/D register r of lane half hi is key (r&3)+8(r>>2)+4hi, so
//                   regs 0..7 / 8..15 are the two 16-key k-steps.  The V^T image stores keys
//                   in exactly that order (bits 2 and 3 of the key index swapped inside
//                   every 16-key group), so P needs NO cross-lane movement at all.
//   The alpha rescale of O^T is lane-local as well (col = query).
// V^T comes from a small pre-pass (vt_pack_kernel) that writes the bank transposed,
// key-permuted and zero-padded per frame to a multiple of 128 keys into caller-provided scratch
// (1 read + 1 write of V, ~1.4% of the attention time at the sizes that matter).
// LDS: double-buffered K [64][DKP+8] and V^T [32*MT][64+8] tiles; the +8 element pad makes
// every row stride an odd number of 16-B slots -> conflict-free ds_read_b128.
// Pipeline: tile i+1 is fetched global->registers before the MFMAs of tile i and written
// to the other LDS buffer after them; one barrier per tile.
// Block order: head = blockIdx % H, so with H = 8 every XCD (block b runs on XCD b % 8)
// serves one head and its L2 holds only that head's bank; bank problems are queued
// before the short source problems so the tail of the grid is filled with short work.
// Three kernels share this structure (geometry table and measurements: DESIGN.md section 4.1):
//   ext_attn_kernel<.., MODE_ALL / MODE_SOURCE>  the plain form
//   ext_attn_kernel<.., MODE_DUAL>               q/k injection: uncond + cond share QK^T and the softmax
//   ext_attn_pp_kernel                           two query tiles per wave, softmax of one interleaved
//                                                in program order with the MFMAs of the other
#include <stdlib.h>

#include <type_traits>

#include "attn_fused.h"
#include "tf_common.h"

// Binades by which a query's softmax reference point may trail its running maximum (kernels without the score bound)
#ifndef TF_ATTN_LAG
#define TF_ATTN_LAG 8.0f
#endif

namespace {

template <int DH, int KT>   // KT = keys per staged tile (one barrier interval): 64 or 128
struct AttnCfg {
    static constexpr int KS = (DH + 15) / 16;   // QK^T k-steps over the head dim
    static constexpr int DKP = KS * 16;         // head dim padded for QK^T (zero columns)
    static constexpr int KROW = DKP + 8;        // K row stride in LDS (elements)
    static constexpr int MT = (DH + 31) / 32;   // PV M-tiles over the head dim
    static constexpr int VROWS = MT * 32;       // V^T rows in LDS (rows >= DH stay constant)
    static constexpr int VROW = KT + 8;         // V^T row stride in LDS (elements)
    static constexpr int PPR = DH / 8;          // 16-B pieces per K row
    static constexpr int VPR = KT / 8;          // 16-B pieces per V^T row
    static constexpr int SUB = KT / 64;         // 64-key sub-tiles per staged tile
    static constexpr int K_ELEMS = KT * KROW;
    static constexpr int V_ELEMS = VROWS * VROW;
    static constexpr int npk(int nt) { return (KT * PPR + nt - 1) / nt; }   // K pieces per thread
    static constexpr int npv(int nt) { return (DH * VPR + nt - 1) / nt; }   // V^T pieces per thread
    static constexpr size_t lds_bytes(int nb) { return 2 * (size_t)(K_ELEMS + nb * V_ELEMS) * 2; }
};

enum { MODE_ALL = 0, MODE_SOURCE = 1, MODE_DUAL = 2 };

// Head dims whose streaming kernels use the Cauchy-Schwarz score bound |q.k| <= |q| max|k| (per-block key norms from the
// pre-pass) to skip the per-tile maximum: Dh = 40 since round 2, Dh = 64 since round 6 (A/B switch TF_TUNE_NO_BOUND64).
constexpr bool attn_has_bound(int dh) {
#ifdef TF_TUNE_NO_BOUND64
    return dh == 40;
#elif defined(TF_TUNE_BOUND80)
    return dh == 40 || dh == 64 || dh == 80;
#else
    return dh == 40 || dh == 64;
#endif
}

struct AttnParams {
    const void* q;
    const void* k;
    const void* vt;
    const float* knorm2;  // [3][H][K*Spad/64] max |k|^2 per 64-key block, Dh = 40 kernels only (from vt_pack_kernel)
    void* out;
    int K, Kq, q_frame0, S, H, Spad, nQT, inject, fold;   // fold: TF_ATTN_FOLD_SCALE (Dh = 40 only)
    int part;  // 0 = all three branches, TF_ATTN_BANK_ONLY, TF_ATTN_SOURCE_ONLY
    int out_f32;       // TF_ATTN_OUT_F32: `out` is float (the normalised fp32 accumulator, no 16-bit rounding)
    int nseg;          // > 1: every bank problem is split into nseg runs of bank frames (small grids, see split_plan)
    int bit_stable;    // TF_ATTN_NO_SPLIT: kernel choice and arithmetic are functions of the shape alone
    int mix;           // TF_ATTN_HINT_MIX: the mixed-MFMA-shape form (Dh = 40) whatever the launch size
    float* partials;   // [2 banks][Kq][H][S][nseg][Dh + 8] fp32: unnormalised O, l, log2-domain shift  // K bank frames; queries = frames q_frame0 .. +Kq
    int64_t ld;      // token stride of k and v
    int64_t ld_q;    // token stride of q (its own: a rank's q may be a column slab of the fused projection while the
                     // bank arrives from a collective as dense slabs)
    // branch / frame strides in elements (dense tensors: frame = S*ld, branch = frames*S*ld; out: S*H*Dh, Kq*S*H*Dh).
    // A caller whose q / k / v arrive from a collective reads them in the layout the collective delivers and has
    // the output written in the layout the next collective sends (tf_ext_attn_fwd_strided, sharded.py).
    int64_t q_bs, q_fs, k_bs, k_fs, v_bs, v_fs, o_bs, o_fs;
    float c;  // scale * log2(e)
};

// max over the two lanes (l, l ^ 32) that share a query: v_permlane32_swap instead of an LDS round trip
__device__ __forceinline__ float max_with_lane_xor32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Row stride (elements) of the V^T scratch: K*Spad positions + 64 elements of padding.  K*Spad*2 bytes is a
// large power of two at the BASELINE shapes (64 KiB at cfg2 level 0); the rows d = 0..Dh-1 of one V^T tile
// would then all map to the same memory channel and the tile loads serialise.  The 128-byte skew spreads them.
__host__ __device__ __forceinline__ int64_t vt_row_stride(int K, int Spad) { return (int64_t)K * Spad + 64; }

static inline size_t vt_bytes(int K, int Spad, int H, int Dh) {
    return (size_t)3 * H * Dh * (size_t)vt_row_stride(K, Spad) * 2;
}

// 4 consecutive output features of one query: rounded to the 16-bit I/O type, or, with TF_ATTN_OUT_F32, the
// normalised fp32 accumulator itself (the caller's `out` is then float [3,Kq,S,H*Dh])
template <typename E, typename V4>
__device__ __forceinline__ void store_out4(void* out, int64_t elem_off, f32x4 x, int out_f32) {
    if (out_f32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + elem_off) = x;
    } else {
        V4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (E)x[i];
        *reinterpret_cast<u32x2*>(reinterpret_cast<E*>(out) + elem_off) = __builtin_bit_cast(u32x2, w);
    }
}

__device__ __forceinline__ int swap23(int x) { return (x & ~12) | ((x & 4) << 1) | ((x & 8) >> 1); }

// V [3,K,S,H*DH] (token stride ld) -> Vt [3][H][DH][K*Spad + 64], position = f*Spad + swap23(key in frame),
// zero for keys >= S.  grid = (Spad/64, H, 3*K), 256 threads; one workgroup = 64 keys x DH of one head.
// 16-byte global accesses on both sides (rows of V in, 8 consecutive positions of one V^T row out); the
// transpose itself is 2-byte LDS reads of a [64][DH+2] tile (odd dword stride: conflict-free columns).
// With k != nullptr (Dh = 40 kernels) the same workgroup also writes max |k|^2 over its 64 keys of this head to
// knorm2[(b*H + h) * K*Spad/64 + f*Spad/64 + tt]: the score bound q.k <= |q| max|k| of ext_attn_kernel.
template <typename T>
__global__ __launch_bounds__(256) void vt_pack_kernel(const typename T::elem* __restrict__ v,
                                                      typename T::elem* __restrict__ vt,
                                                      const typename T::elem* __restrict__ k,
                                                      float* __restrict__ knorm2, int inject, int bf0, int K,
                                                      int S, int H, int DH, int Spad, int64_t ld, int64_t v_bs,
                                                      int64_t v_fs, int64_t k_bs, int64_t k_fs) {
    typedef typename T::elem E;
    typedef typename T::vec8 vec8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    E* tile = reinterpret_cast<E*>(smem);  // [64][DH + 2]
    const int row = DH + 2;
    const int ppr = DH >> 3;               // 16-B pieces per V row
    const int tt = blockIdx.x, h = blockIdx.y, bf = blockIdx.z + bf0;  // bf = b*K + f; bf0 = first (branch, frame)
    const int b = bf / K, f = bf - b * K;
    const E* src = v + b * v_bs + f * v_fs + h * DH;
    // keys of branch b without injection; with injection every branch reads the SOURCE keys, whose norms the
    // first packed branch computes
    if (k != nullptr && (!inject || b == bf0 / K) && threadIdx.x < 64) {   // wave 0: one key per lane
        const int kb = inject ? 0 : b;
        const int kk = tt * 64 + (int)threadIdx.x;
        float acc = 0.f;
        if (kk < S) {
            const E* kp = k + kb * k_bs + f * k_fs + (int64_t)kk * ld + h * DH;
            for (int c8 = 0; c8 < DH; c8 += 8) {
                const vec8 x = __builtin_bit_cast(vec8, ld16(kp + c8));
#pragma unroll
                for (int j = 0; j < 8; ++j) acc = fmaf((float)x[j], (float)x[j], acc);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc = fmaxf(acc, __shfl_xor(acc, o));
        if (threadIdx.x == 0) knorm2[((int64_t)(kb * H + h) * K + f) * (Spad / 64) + tt] = acc;
    }
    for (int id = threadIdx.x; id < 64 * ppr; id += 256) {
        const int key = id / ppr, pc = id - key * ppr;
        const int kk = tt * 64 + key;
        const vec8 val = kk < S ? __builtin_bit_cast(vec8, ld16(src + (int64_t)kk * ld + pc * 8))
                                : __builtin_bit_cast(vec8, u32x4{0, 0, 0, 0});
        E* dstp = tile + key * row + pc * 8;   // (DH+2)*2 bytes per row: only 4-byte aligned -> element stores
#pragma unroll
        for (int j = 0; j < 8; ++j) dstp[j] = val[j];
    }
    __syncthreads();
    const int64_t vt_row = vt_row_stride(K, Spad);
    E* dst = vt + ((int64_t)(b * H + h) * DH) * vt_row + (int64_t)f * Spad + tt * 64;
    for (int id = threadIdx.x; id < DH * 8; id += 256) {
        const int d = id >> 3, pg = id & 7;   // 8 consecutive positions pg*8 .. +7 of V^T row d
        vec8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = tile[swap23(pg * 8 + j) * row + d];
        st16(dst + (int64_t)d * vt_row + pg * 8, __builtin_bit_cast(u32x4, o));
    }
}

// QT   = 32-query tiles per wave (1 or 2)
// NW   = waves per workgroup (4 or 8): a workgroup covers 32*QT*NW queries of one (branch, frame, head)
//        and shares every staged K / V^T tile among them
// MODE = MODE_ALL:    every (branch, frame, head, query tile) problem, bank problems first
//        MODE_SOURCE: only the source-branch problems
//        MODE_DUAL:   q/k injection active -- uncond and cond share q, k, the scores and P
//                     (tokenflow_utils.py:124-130), so ONE workgroup computes both: QK^T and the softmax
//                     once, two P.V products against the two V banks (NB = 2).
// MINW = min waves per SIMD for the register allocator
// FQ   = fold the softmax scale into Q (see FOLD below; opt-in, TF_ATTN_FOLD_SCALE); false = the default, fp32
//        scaling of the scores as the reference does (tokenflow_utils.py:173-175 `* self.scale` on the bmm output)
// SB   = single LDS buffer (two barriers per tile) instead of two: half the LDS per workgroup.  For head dim 160, where
//        the double-buffered tiles (89 KB) allow ONE workgroup per CU and a wave waits alone for every 1 KB fragment
template <typename T, int DH, int QT, int NW, int MODE, int MINW, int KT, bool FQ, bool SB = false>
__global__ __launch_bounds__(64 * NW, MINW) void ext_attn_kernel(AttnParams p) {
    typedef AttnCfg<DH, KT> C;
    typedef typename T::elem E;
    typedef typename T::vec8 vec8;
    typedef typename T::vec4 vec4;
    constexpr int NT = 64 * NW;
    constexpr int NB = MODE == MODE_DUAL ? 2 : 1;   // V banks handled by this workgroup
    constexpr int NPK = C::npk(NT), NPV = C::npv(NT);
    constexpr int NBUFS = SB ? 1 : 2;
    // PACK (dual-V at Dh = 40): the two banks' V^T rows share ONE LDS image of 3 M-tiles -- rows 0-39 uncond,
    // 40-79 cond, row 80 = 1.0 (the common denominator row), 81-95 zero -- instead of two images of 2 M-tiles
    // with 24 idle rows each: 12 instead of 16 P.V MFMAs per 64-key tile (18 instead of 22 with QK^T).
    constexpr bool PACK = MODE == MODE_DUAL && DH == 40;
    constexpr int VIMG_ROWS = PACK ? 96 : NB * C::VROWS;       // V^T rows of one LDS buffer
    constexpr int VB_ROWS = PACK ? DH : C::VROWS;              // row offset between the banks inside it
    constexpr int BUF_ELEMS = C::K_ELEMS + VIMG_ROWS * C::VROW;
    // When the head dim is not a multiple of 32 the last PV M-tile has unused rows: row DH of the
    // V^T image is set to 1.0, so that accumulator row collects sum_k P[k] -- the softmax
    // denominator comes out of the MFMA for free, summed over the SAME rounded P as the numerator.
    constexpr bool ONES = (DH % 32) != 0;
    constexpr int ONES_R = ((DH % 32) & 3) + 4 * ((DH % 32) >> 3);  // C/D register of row DH%32 (lane half 0)
    static_assert(!ONES || ((DH % 32) & 4) == 0, "row DH must live in lane half 0");
    // FOLD (head dims with spare QK^T columns, i.e. Dh = 40): the softmax's scale AND shift ride in the MFMA.
    //   * Q fragments hold q * (scale*log2 e), rounded to the MFMA input type once per kernel;
    //   * the first pad column of the K image (column Dh) is 1.0 and the matching pad element of the Q
    //     fragment holds -shift, so the accumulator comes out as  s*c - shift  and P = exp2(acc) directly:
    //     no v_fma per score (32 of ~87 VALU instructions per 32x64 tile; the kernel is VALU-issue bound).
    //   The shift is a per-query running value, representable in the input type, moved only when a tile's
    //   maximum exceeds it by more than FOLD_T (and always on the first tile); softmax is invariant to the
    //   shift, numerator and denominator see the same P, so no accuracy is traded for the deferral.
    //   What IS traded: q*c is rounded to 16 bit once, a relative error <= 2^-9 per element that perturbs each
    //   score by ~2^-9/sqrt(3) * c * sqrt(sum_d (q_d k_d)^2) -- the size class of the P rounding for ordinary
    //   scores, but 3-12x the whole error budget on peaked softmaxes (logit std 4-16, profiles/r02_fold_accuracy.txt):
    //   NOT the default; TF_ATTN_FOLD_SCALE opts in.
    constexpr bool FOLD = FQ && ONES && (C::DKP > DH);
    constexpr int SH_T = DH / 16, SH_HI = (DH % 16) / 8;   // k-step and lane half that hold column Dh
    //   Most tiles never look at their maximum: |acc + shift| = |q'.k| <= |q'| max_k|k| (Cauchy-Schwarz; the
    //   key norm bound comes with the vt_pack_kernel pre-pass), so while  |q'| |k|max - shift <= FOLD_T  no
    //   score of any tile can exceed the threshold and the max3 chain + permlane (16 of ~66 VALU per tile)
    //   is skipped; a query whose bound is loose falls back to the per-tile maximum.  exp2 of FOLD_T must
    //   stay inside the input type's range (the row sum is accumulated in fp32): 2^60 for bf16, 2^14 for f16
    //   (f16 tops out at 65504; on N(0,1) data the f16 bound is usually too loose to skip anything, and a
    //   per-tile bound from the block's own max |k| measured slower than the fallback it avoids).
    //   Measured (MI355X, cfg2 level 0): -6.5 % kernel time for +7..15 us in the pre-pass.
    constexpr float FOLD_T = std::is_same<E, _Float16>::value ? 14.0f : 60.0f;
    // BOUND (Dh = 40, both scalings): the Cauchy-Schwarz score bound described above lets a wave skip the per-tile
    // maximum.  With fp32 scaling the running "maximum" m_run becomes a deferred shift exactly as in the folded
    // form: it is set from the first tile's maximum and moved only when a tile maximum exceeds it by more than
    // FOLD_T binades; P = exp2((s - m_run) c) may then exceed 1 (<= 2^FOLD_T), numerator and denominator see the
    // same P.  Saves the 16 v_max3 + permlane of most tiles and most O rescales.
    //   Measured (round 2, cfg2 level 0, fp32 scaling): 4.25 -> 4.03 ms with the bound; the folded form is 3.58 ms.
    constexpr bool BOUND = attn_has_bound(DH);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    auto sK = [&](int buf) { return reinterpret_cast<E*>(smem) + buf * BUF_ELEMS; };
    auto sV = [&](int buf, int vb) {
        return reinterpret_cast<E*>(smem) + buf * BUF_ELEMS + C::K_ELEMS + vb * VB_ROWS * C::VROW;
    };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hi = lane >> 5;
    const int l31 = lane & 31;
    const int K = p.K, Kq = p.Kq, S = p.S, H = p.H;

    // ---- problem decode
    const int h = blockIdx.x % H;
    int u = blockIdx.x / H;
    int b, f, qt;  // f = query frame, local index in [0, Kq)
    int seg = 0;   // run of bank frames this workgroup covers (split form: the bank problems come nseg times)
    const int nseg = MODE == MODE_SOURCE ? 1 : p.nseg;
    if constexpr (MODE == MODE_ALL) {   // bank problems (uncond, cond) first, then the short source ones
        const int nbank = 2 * Kq * p.nQT * nseg;
        if (u < nbank) {
            seg = u % nseg;
            u /= nseg;
            b = 1 + u / (Kq * p.nQT);
            u -= (b - 1) * Kq * p.nQT;
        } else {
            u -= nbank;
            b = 0;
        }
    } else if constexpr (MODE == MODE_DUAL) {
        b = 1;
        seg = u % nseg;
        u /= nseg;
    } else {
        b = 0;
    }
    f = u / p.nQT;
    qt = u - f * p.nQT;
    const int bq = (p.inject && b > 0) ? 0 : b;  // branch whose q and k are used (tokenflow_utils.py:124-130)
    const bool split = nseg > 1 && b > 0;
    const int f_lo = b == 0 ? p.q_frame0 + f : (seg * K) / nseg;
    const int n_fr = b == 0 ? 1 : ((seg + 1) * K) / nseg - f_lo;
    const int tpf = (S + KT - 1) / KT;  // staged tiles per frame
    const int ntiles = n_fr * tpf;
    const bool ragged = (S % KT) != 0;

    const E* qg = reinterpret_cast<const E*>(p.q);
    const E* kg = reinterpret_cast<const E*>(p.k) + bq * p.k_bs + h * DH;
    const int64_t vt_row = vt_row_stride(K, p.Spad);
    const E* vg[NB];
#pragma unroll
    for (int vb = 0; vb < NB; ++vb)
        vg[vb] = reinterpret_cast<const E*>(p.vt) + ((int64_t)((b + vb) * H + h) * DH) * vt_row;

    // ---- LDS pads, written once and never staged over: K columns DH..DKP-1 = 0,
    //      V^T rows DH..VROWS-1 = 0 except row DH = 1 (denominator row) when ONES.
    if constexpr (C::DKP > DH) {
        for (int id = tid; id < NBUFS * KT * (C::DKP - DH); id += NT) {
            const int bufi = id / (KT * (C::DKP - DH));
            const int r = (id / (C::DKP - DH)) % KT, cidx = id % (C::DKP - DH);
            sK(bufi)[r * C::KROW + DH + cidx] = (E)((FOLD && cidx == 0) ? 1.f : 0.f);
        }
    }
    if constexpr (PACK) {
        for (int id = tid; id < NBUFS * (VIMG_ROWS - NB * DH) * KT; id += NT) {
            const int bufi = id / ((VIMG_ROWS - NB * DH) * KT);
            const int r = (id / KT) % (VIMG_ROWS - NB * DH), cidx = id % KT;
            sV(bufi, 0)[(NB * DH + r) * C::VROW + cidx] = (E)(r == 0 ? 1.f : 0.f);
        }
    } else if constexpr (C::VROWS > DH) {
        for (int id = tid; id < NBUFS * NB * (C::VROWS - DH) * KT; id += NT) {
            const int bv = id / ((C::VROWS - DH) * KT);
            const int r = (id / KT) % (C::VROWS - DH), cidx = id % KT;
            sV(bv / NB, bv % NB)[(DH + r) * C::VROW + cidx] = (E)((ONES && r == 0) ? 1.f : 0.f);
        }
    }

    // ---- Q fragments (B operand of S^T = K Q^T), resident for the whole kernel
    int q_row[QT];
    bool q_ok[QT];
    vec8 qf[QT][C::KS];
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
        q_row[qi] = qt * (32 * QT * NW) + (wave * QT + qi) * 32 + l31;
        q_ok[qi] = q_row[qi] < S;
        const E* qp = qg + bq * p.q_bs + f * p.q_fs + (int64_t)(q_ok[qi] ? q_row[qi] : S - 1) * p.ld_q + h * DH;
#pragma unroll
        for (int t = 0; t < C::KS; ++t) {
            const int col = 16 * t + 8 * hi;
            qf[qi][t] = __builtin_bit_cast(vec8, col < DH ? ld16(qp + col) : u32x4{0, 0, 0, 0});
            if constexpr (FOLD) {   // q * (scale*log2 e), rounded once to the MFMA input type
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[qi][t][j] = (E)((float)qf[qi][t][j] * p.c);
            }
        }
    }
    float s_bound[QT] = {};   // BOUND: upper bound of q.k*c (log2 units) over every key of the bank (1.001 covers fp32 rounding)
    if constexpr (BOUND) {
        const int ppf = p.Spad / 64;   // 64-key blocks per frame; this problem sees frames f_lo .. f_lo + n_fr - 1
        const float* part = p.knorm2 + ((int64_t)(bq * H + h) * K + f_lo) * ppf;
        float kn2 = 0.f;
        for (int i = lane; i < n_fr * ppf; i += 64) kn2 = fmaxf(kn2, part[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kn2 = fmaxf(kn2, __shfl_xor(kn2, o));
        const float kn = __builtin_sqrtf(kn2) * 1.001f;
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) {
            float q2 = 0.f;
#pragma unroll
            for (int t = 0; t < C::KS; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) q2 = fmaf((float)qf[qi][t][j], (float)qf[qi][t][j], q2);
            q2 += __shfl_xor(q2, 32);   // the two lanes of a query hold disjoint halves of its columns
            s_bound[qi] = __builtin_sqrtf(q2) * kn * (FOLD ? 1.f : p.c);   // log2 units in both forms
        }
    }

    // ---- staging: per-thread piece offsets are loop-invariant; a tile only moves uniform base pointers
    // The loads are branch-free (one straight-line path, no exec masking): a lane without a piece re-loads
    // the last piece, a row past S is clamped by a select.  Any control flow around the loads makes the
    // compiler merge the two definitions of the staging registers with v_mov copies, and those copies
    // need the data: an s_waitcnt vmcnt(0) right behind the loads that exposes the full L2 latency on
    // every tile (measured ~0.8 ms of a 4.2 ms launch).
    u32x4 rk[NPK], rv[NB][NPV];
    int k_row[NPK], k_col[NPK], k_goff[NPK], k_loff[NPK], v_goff[NPV], v_loff[NPV];
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
        const int id = min(tid + NT * i, KT * C::PPR - 1);
        k_row[i] = id / C::PPR;
        k_col[i] = (id - k_row[i] * C::PPR) * 8;
        k_goff[i] = k_row[i] * (int)p.ld + k_col[i];
        k_loff[i] = k_row[i] * C::KROW + k_col[i];
    }
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
        const int id = min(tid + NT * i, DH * C::VPR - 1);
        v_goff[i] = (id / C::VPR) * (int)vt_row + (id % C::VPR) * 8;
        v_loff[i] = (id / C::VPR) * C::VROW + (id % C::VPR) * 8;
    }
    // Tile cursors (see ext_attn_pp_kernel): uniform pointer bumps, no division per tile.
    const int k_wrap = S - (tpf - 1) * KT, v_wrap = p.Spad - (tpf - 1) * KT;
    const int64_t k_wrap_off = p.k_fs - (int64_t)(tpf - 1) * KT * p.ld;   // last tile of a frame -> first tile of the next
    const E* k_next = kg + f_lo * p.k_fs;
    const E* v_next[NB];
#pragma unroll
    for (int vb = 0; vb < NB; ++vb) v_next[vb] = vg[vb] + (int64_t)f_lo * p.Spad;
    int ld_tt = 0;
    auto stage_load = [&]() {
        const bool wrap = ld_tt == tpf - 1;
        const int rlim = wrap ? k_wrap - 1 : KT - 1;   // last valid key row of this tile (rows past S are masked later)
        const int clamp_off = rlim * (int)p.ld;
#pragma unroll
        for (int i = 0; i < NPK; ++i) rk[i] = ld16(k_next + (k_row[i] <= rlim ? k_goff[i] : clamp_off + k_col[i]));
#pragma unroll
        for (int vb = 0; vb < NB; ++vb) {
#pragma unroll
            for (int i = 0; i < NPV; ++i) rv[vb][i] = ld16(v_next[vb] + v_goff[i]);
            v_next[vb] += wrap ? v_wrap : KT;
        }
        k_next += wrap ? k_wrap_off : (int64_t)KT * p.ld;
        ld_tt = wrap ? 0 : ld_tt + 1;
    };
    auto stage_write = [&](int buf) {
        E* kb = sK(buf);
#pragma unroll
        for (int i = 0; i < NPK; ++i)
            if (tid + NT * i < KT * C::PPR) st16(kb + k_loff[i], rk[i]);
#pragma unroll
        for (int vb = 0; vb < NB; ++vb) {
            E* vbp = sV(buf, vb);
#pragma unroll
            for (int i = 0; i < NPV; ++i)
                if (tid + NT * i < DH * C::VPR) st16(vbp + v_loff[i], rv[vb][i]);
        }
    };

    f32x16 o[NB][QT][C::MT];
    float m_run[QT], l_run[QT];  // running max of the RAW scores (scale > 0); this lane's share of the denominator
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
        m_run[qi] = FOLD ? 0.f : -INFINITY;   // FOLD: the current shift
        l_run[qi] = 0.f;
#pragma unroll
        for (int vb = 0; vb < NB; ++vb)
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[vb][qi][mt][r] = 0.f;
    }
    const float c = p.c;
    const f32x2 c2 = {c, c};

    stage_load();
    __syncthreads();  // pad fill visible before anything reads; staging regions are disjoint from the pads
    stage_write(0);
    __syncthreads();

    int tt_cur = 0;   // tile index within the frame of the tile being computed
    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = SB ? 0 : tile & 1;
        const bool has_next = tile + 1 < ntiles;
        if (has_next) stage_load();

#pragma unroll
        for (int sub = 0; sub < C::SUB; ++sub) {
            const int key0 = tt_cur * KT + sub * 64;  // first key (within the frame) of this 64-key sub-tile
            if (C::SUB > 1 && ragged && key0 >= S) break;   // nothing but padding left in this tile
            // Program order per tile: QK(q0) QK(q1) | softmax(q0) PV(q0) | softmax(q1) PV(q1).
            // MFMAs execute asynchronously behind the in-order issue, so the softmax VALU of one query
            // tile runs while the matrix pipe works on the other one's QK^T / P.V.
            f32x16 s[QT][2];  // S^T tiles: 64 keys x 32 queries each
#pragma unroll
            for (int qi = 0; qi < QT; ++qi)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[qi][kt][r] = 0.f;
            // k-step outermost: consecutive MFMAs hit DIFFERENT accumulators (two MFMAs on the same accumulator
            // with other instructions between them cost ~43 extra cycles, MI355X_MICROARCH.md cycle constants)
#pragma unroll
            for (int t = 0; t < C::KS; ++t)
#pragma unroll
                for (int qi = 0; qi < QT; ++qi)
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) {
                        const E* krow = sK(buf) + (sub * 64 + kt * 32 + l31) * C::KROW + 8 * hi;
                        s[qi][kt] = T::mfma32(__builtin_bit_cast(vec8, ld16(krow + 16 * t)), qf[qi][t], s[qi][kt]);
                    }
            if (ragged && key0 + 64 > S) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (key0 + kt * 32 + cd_row(r, hi) >= S) {
#pragma unroll
                            for (int qi = 0; qi < QT; ++qi) s[qi][kt][r] = -INFINITY;
                        }
            }

#pragma unroll
            for (int qi = 0; qi < QT; ++qi) {
                // ---- online softmax (lane-local; the two lanes of a query share m)
                auto tile_max = [&]() {
                    float mx = s[qi][0][0];
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qi][kt][r]);
                    return max_with_lane_xor32(mx);
                };
                vec8 pf[4];
                if constexpr (FOLD) {
                    // s already is  score*c - shift.  Move the shift only when needed (wave-uniform branches).
                    const bool first = tile == 0 && sub == 0;
                    float delta = 0.f;
                    float mx = 0.f;
                    const bool look = !BOUND || first || __any(s_bound[qi] - m_run[qi] > FOLD_T);
                    if (look) mx = tile_max();
                    if (look && (first || __any(mx > FOLD_T))) {
                        const float sh_old = m_run[qi];          // m_run holds the current shift (0 before tile 0)
                        const float sh_new = (first || mx > FOLD_T) ? (float)(E)(sh_old + mx) : sh_old;
                        delta = sh_new - sh_old;
                        // first tile: O is still zero, and exp2(-delta) overflows to +inf when every score of the
                        // tile is far below zero (0 * inf = NaN) -- nothing to rescale there
                        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
                        m_run[qi] = sh_new;
                        if (hi == SH_HI) qf[qi][SH_T][0] = (E)(-sh_new);
#pragma unroll
                        for (int vb = 0; vb < NB; ++vb)
#pragma unroll
                            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                                for (int r = 0; r < 16; ++r) o[vb][qi][mt][r] *= alpha;
#pragma unroll
                        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) s[qi][kt][r] -= delta;
                    }
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            pf[kt * 2 + (r >> 3)][r & 7] = (E)__builtin_amdgcn_exp2f(s[qi][kt][r]);
                } else {
                    bool move;      // wave-uniform: some query's shift / running maximum changes on this tile
                    float m_new;
                    if constexpr (BOUND) {
                        // m_run = deferred shift (raw-score units; -inf before the first tile, so the first tile
                        // always looks and always moves).  No tile can overflow while (bound - shift) <= FOLD_T.
                        const bool look = __any(s_bound[qi] - m_run[qi] * c > FOLD_T);
                        move = false;
                        m_new = m_run[qi];
                        if (look) {
                            const float mx = tile_max();
                            const bool over = (mx - m_run[qi]) * c > FOLD_T;
                            move = __any(over);
                            if (over) m_new = mx;
                        }
                    } else {
                        // reference point = running maximum with a lag of 8 binades (see ext_attn_il_kernel): per-query
                        // decision, alpha == 1 exactly for a query whose reference stays
                        const float mx = tile_max();
                        const bool over = mx > m_run[qi] + TF_ATTN_LAG / c;
                        move = __any(over);
                        m_new = over ? mx : m_run[qi];
                    }
                    if (move) {
                        const float alpha = __builtin_amdgcn_exp2f((m_run[qi] - m_new) * c);  // exp2(-inf) = 0 on tile 0
                        m_run[qi] = m_new;
                        if constexpr (!ONES) l_run[qi] *= alpha;
#pragma unroll
                        for (int vb = 0; vb < NB; ++vb)
#pragma unroll
                            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                                for (int r = 0; r < 16; ++r) o[vb][qi][mt][r] *= alpha;
                    }
                    const float mc = m_run[qi] * c;
                    const f32x2 mc2 = {mc, mc};
                    float lsum = 0.f;
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            // v_pk_fma_f32: 16 instead of 32 VALU per tile (two scalar v_fma measured +16 % kernel time)
                            const f32x2 x = f32x2{s[qi][kt][r], s[qi][kt][r + 1]} * c2 - mc2;
                            const float p0 = __builtin_amdgcn_exp2f(x[0]);
                            const float p1 = __builtin_amdgcn_exp2f(x[1]);
                            if constexpr (!ONES) lsum += p0 + p1;
                            pf[kt * 2 + (r >> 3)][r & 7] = (E)p0;
                            pf[kt * 2 + (r >> 3)][(r & 7) + 1] = (E)p1;
                        }
                    if constexpr (!ONES) l_run[qi] += lsum;
                }
                // ---- O^T += V^T . P  (once per V bank)
                if constexpr (PACK) {   // 3 M-tiles over the packed image: accumulators o[0][.][0], o[0][.][1], o[1][.][0]
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            const E* vrow = sV(buf, 0) + (g * 32 + l31) * C::VROW + sub * 64 + 8 * hi;
                            o[g >> 1][qi][g & 1] = T::mfma32(__builtin_bit_cast(vec8, ld16(vrow + 16 * ks)), pf[ks],
                                                             o[g >> 1][qi][g & 1]);
                        }
                } else {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)   // k-step outermost: round-robin over the NB * MT accumulators
#pragma unroll
                        for (int vb = 0; vb < NB; ++vb)
#pragma unroll
                            for (int mt = 0; mt < C::MT; ++mt) {
                                const E* vrow = sV(buf, vb) + (mt * 32 + l31) * C::VROW + sub * 64 + 8 * hi;
                                o[vb][qi][mt] = T::mfma32(__builtin_bit_cast(vec8, ld16(vrow + 16 * ks)), pf[ks],
                                                          o[vb][qi][mt]);
                            }
                }
            }

        }

        if constexpr (SB) {
            if (has_next) {
                __syncthreads();   // every wave is done reading the tile before it is overwritten
                stage_write(0);
            }
        } else if (has_next) {
            stage_write(buf ^ 1);
        }
        tt_cur = tt_cur == tpf - 1 ? 0 : tt_cur + 1;
        __syncthreads();
    }

    // ---- epilogue: normalise, round, store 4 consecutive d (8 B) per register group
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
        float l_tot;
        if constexpr (PACK)
            l_tot = __shfl(o[1][qi][0][8], l31);   // image row 80 = row 16 of the third M-tile: register 8, lane half 0
        else if constexpr (ONES)
            l_tot = __shfl(o[0][qi][C::MT - 1][ONES_R], l31);  // row DH lives in lane half 0 of the last M-tile
        else
            l_tot = l_run[qi] + __shfl_xor(l_run[qi], 32);
        const float inv_l = 1.0f / l_tot;
        if (split) {
            // split form: this workgroup saw only a run of the bank's frames -- leave the unnormalised O, the
            // denominator and the shift (log2 domain) for attn_merge_kernel
            if (q_ok[qi]) {
                constexpr int PS = DH + 8;
                const float lshift = FOLD ? m_run[qi] : m_run[qi] * c;
                auto row_ptr = [&](int vb) {
                    const int64_t R = (((int64_t)(b - 1 + vb) * Kq + f) * H + h) * S + q_row[qi];
                    return p.partials + (R * nseg + seg) * PS;
                };
                if constexpr (PACK) {
#pragma unroll
                    for (int g = 0; g < 3; ++g)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const int R = g * 32 + 8 * rg + 4 * hi;
                            if (R < NB * DH) {
                                const int vb = R >= DH ? 1 : 0;
                                f32x4 w;
#pragma unroll
                                for (int i = 0; i < 4; ++i) w[i] = o[g >> 1][qi][g & 1][rg * 4 + i];
                                *reinterpret_cast<f32x4*>(row_ptr(vb) + (R - vb * DH)) = w;
                            }
                        }
                } else {
#pragma unroll
                    for (int vb = 0; vb < NB; ++vb)
#pragma unroll
                        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg) {
                                const int d0 = mt * 32 + 8 * rg + 4 * hi;
                                if (d0 < DH) {
                                    f32x4 w;
#pragma unroll
                                    for (int i = 0; i < 4; ++i) w[i] = o[vb][qi][mt][rg * 4 + i];
                                    *reinterpret_cast<f32x4*>(row_ptr(vb) + d0) = w;
                                }
                            }
                }
                if (hi == 0) {
#pragma unroll
                    for (int vb = 0; vb < NB; ++vb) {
                        row_ptr(vb)[DH] = l_tot;
                        row_ptr(vb)[DH + 1] = lshift;
                    }
                }
            }
        } else if (PACK && q_ok[qi]) {
            const int64_t op0 = b * p.o_bs + f * p.o_fs + (int64_t)q_row[qi] * (H * DH) + h * DH;
            const int64_t branch = p.o_bs;
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int R = g * 32 + 8 * rg + 4 * hi;    // image row of this group of 4 (never straddles a bank)
                    if (R < NB * DH) {
                        const int vb = R >= DH ? 1 : 0;
                        f32x4 w;
#pragma unroll
                        for (int i = 0; i < 4; ++i) w[i] = o[g >> 1][qi][g & 1][rg * 4 + i] * inv_l;
                        store_out4<E, vec4>(p.out, op0 + vb * branch + (R - vb * DH), w, p.out_f32);
                    }
                }
        } else if (q_ok[qi]) {
#pragma unroll
            for (int vb = 0; vb < NB; ++vb) {
                const int64_t op = (b + vb) * p.o_bs + f * p.o_fs + (int64_t)q_row[qi] * (H * DH) + h * DH;
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int d0 = mt * 32 + 8 * rg + 4 * hi;
                        if (d0 < DH) {
                            f32x4 w;
#pragma unroll
                            for (int i = 0; i < 4; ++i) w[i] = o[vb][qi][mt][rg * 4 + i] * inv_l;
                            store_out4<E, vec4>(p.out, op + d0, w, p.out_f32);
                        }
                    }
            }
        }
    }
}

// Split form, second step: out = sum_seg O_seg 2^(sh_seg - M) / sum_seg l_seg 2^(sh_seg - M), M = max_seg sh_seg.
// One thread per (bank, frame, head, query, 4 consecutive d).
template <typename T>
__global__ __launch_bounds__(256) void attn_merge_kernel(const float* __restrict__ partials, void* __restrict__ out,
                                                         int Kq, int S, int H, int DH, int nseg, int out_f32,
                                                         int64_t o_bs, int64_t o_fs) {
    typedef typename T::elem E;
    typedef typename T::vec4 vec4;
    const int PS = DH + 8, dq = DH >> 2;
    const int64_t total = (int64_t)2 * Kq * H * S * dq;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        const int64_t R = g / dq;                 // ((vbank*Kq + f)*H + h)*S + q
        const int d0 = (int)(g - R * dq) * 4;
        const float* pr = partials + R * nseg * PS;
        float M = -INFINITY;
        for (int sg = 0; sg < nseg; ++sg) M = fmaxf(M, pr[sg * PS + DH + 1]);
        f32x4 num = {0.f, 0.f, 0.f, 0.f};
        float den = 0.f;
        for (int sg = 0; sg < nseg; ++sg) {
            const float w = __builtin_amdgcn_exp2f(pr[sg * PS + DH + 1] - M);
            const f32x4 o4 = *reinterpret_cast<const f32x4*>(pr + sg * PS + d0);
            num += o4 * w;
            den = fmaf(pr[sg * PS + DH], w, den);
        }
        const float inv = 1.0f / den;
        const int q = (int)(R % S);
        int64_t t = R / S;
        const int h = (int)(t % H);
        t /= H;
        const int f = (int)(t % Kq), vbank = (int)(t / Kq);
        store_out4<E, vec4>(out, (1 + vbank) * o_bs + f * o_fs + (int64_t)q * (H * DH) + h * DH + d0, num * inv, out_f32);
    }
}

// How many runs of bank frames a bank problem is split into.  The grid of a sharded rank or of a small level has
// too few waves to fill the chip (8-GPU rank at cfg2 level 0: 2 waves per SIMD, level 1: 0.5; single GPU at the
// 16x16 level: 1): split until it has `occ` waves per SIMD, while a run keeps at least 2 tiles.
static int split_plan(int K, int Kq, int S, int H, int Dh, bool inject, int part, bool allow) {
    if (!allow || part == TF_ATTN_SOURCE_ONLY) return 1;
    const bool dual = inject && S >= 256 && Dh != 160;
#ifndef TF_TUNE_OCC160
#define TF_TUNE_OCC160 1   // splitting towards 2 waves per SIMD (the single-buffered tiles would allow two workgroups per
#endif                     // CU) measured slower: 88 vs 85 us at cfg2 level 2, 41 vs 33 us on a rank of 8 (merge included)
    const int occ = Dh == 40 ? 4 : Dh == 160 ? TF_TUNE_OCC160 : dual ? 2 : Dh == 64 ? 4 : 3;   // waves per SIMD the kernels reach
    const int64_t wgs = (int64_t)(dual ? 1 : 2) * Kq * ((S + 127) / 128) * H;   // 4-wave workgroups
    const int tpf = (S + 63) / 64;
    if (K * tpf < 16) return 1;   // a bank of a few tiles: the merge launch costs more than it buys (8x8 level)
    int nseg = 1;
    while (wgs * 4 * nseg < (int64_t)occ * 1024 && nseg * 2 <= K && (K / (nseg * 2)) * tpf >= 2) nseg *= 2;
    // A SHORT bank (<= 8192 keys) that already has half the waves the kernel can hold gains less from the second half than the
    // partial results + merge launch cost: BASELINE config 1, level 0 (K = 4, S = 1024, 2 of 4 waves per SIMD): 0.091 against
    // 0.098 ms unsplit (round 6, one box, TOKENFLOW_ATTN_NSEG A/B); the dual-V launch keeps its split (0.092 against 0.111)
    if (!dual && nseg == 2 && (int64_t)K * S <= 8192) nseg = 1;
    // TOKENFLOW_ATTN_NSEG=n (experiments): force n runs
    static const int forced = [] { const char* e = getenv("TOKENFLOW_ATTN_NSEG"); return e ? atoi(e) : 0; }();
    if (forced > 0) return forced <= K ? forced : K;
    // Large grids: splitting every bank problem into n runs of frames so that the last, nearly empty round of workgroups gets
    // shorter (cfg4 level 0: 3600 bank workgroups on 512 resident slots = 7.03 rounds) was modelled and measured in round 6
    // (profiles/r06_attn_tail_split.txt): n = 2 gains 1.7 % at cfg4 level 0 and 3 % at its level 1, costs 5 % at cfg5 level 0
    // and 3 % at cfg2 level 0; n = 5 loses everywhere (+5..14 %).  The rounds are not in lockstep, the source problems fill
    // the tail, and the partial results' round trip costs more than the model allowed: no planner, the switch above stays.
    // TOKENFLOW_SPLIT_OVER=n (experiments): n further doublings once the chip is full -- shorter workgroups, so that a
    // launch running BESIDE this one (a rank's source branch on an auxiliary stream) is absorbed instead of appended
    static const int over = [] { const char* e = getenv("TOKENFLOW_SPLIT_OVER"); return e ? atoi(e) : 0; }();
    for (int i = 0; i < over && nseg > 1 && nseg * 2 <= K && (K / (nseg * 2)) * tpf >= 2; ++i) nseg *= 2;
    return nseg;
}

// MFMA issue order of one ping-pong region: round-robin over the independent accumulators
// (MT P.V chains over 4 k-steps, 2 QK^T chains over KS k-steps).  Two MFMAs on the SAME accumulator with
// other instructions issued between them cost ~+43 cycles (MI355X_MICROARCH.md, cycle constants), so
// consecutive steps must always hit different accumulators.
template <int MT, int KS>
struct PpSchedule {
    static constexpr int N = 4 * MT + 2 * KS;
    int is_pv[N] = {}, chain[N] = {}, kstep[N] = {};
    constexpr PpSchedule() {
        int i = 0;
        for (int r = 0; r < (KS > 4 ? KS : 4); ++r) {
            for (int mt = 0; mt < MT; ++mt)
                if (r < 4) {
                    is_pv[i] = 1;
                    chain[i] = mt;
                    kstep[i] = r;
                    ++i;
                }
            for (int kt = 0; kt < 2; ++kt)
                if (r < KS) {
                    is_pv[i] = 0;
                    chain[i] = kt;
                    kstep[i] = r;
                    ++i;
                }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Ping-pong variant (head dims whose registers allow two query tiles per wave: 40, 64).
//
// A wave issues in order: a run of back-to-back MFMAs blocks its own VALU until the last one has
// issued, so softmax and matrix work of ONE query tile can never overlap inside a wave.  Here every
// wave owns two query tiles, streams A and B, half a tile apart:
//     R1(t):  exp/round P_A(t)   (VALU)   ||   O_B += V(t-1) P_B(t-1),  S_B(t) = K(t) Q_B     (MFMA)
//     R2(t):  exp/round P_B(t)   (VALU)   ||   O_A += V(t) P_A(t),      S_A(t+1) = K(t+1) Q_A (MFMA)
// and inside a region the instruction stream is forced (sched_group_barrier) to alternate
// 1 MFMA : ~4 VALU/TRANS : 1 LDS fragment read, i.e. the VALU work of one stream rides in the issue
// gaps of the other stream's MFMAs.  K(t) lives in Kbuf[t&1], V(t) in Vbuf[t&1]; K(t+1) and V(t) are
// written at the top of R1(t) from registers loaded one iteration earlier; ONE barrier per tile
// (between R1 and R2) orders all LDS hazards (see the per-line comments).
template <typename T, int DH, int MODE, int MINW, bool FQ = false>
__global__ __launch_bounds__(256, MINW) void ext_attn_pp_kernel(AttnParams p) {
    typedef AttnCfg<DH, 64> C;
    typedef typename T::elem E;
    typedef typename T::vec8 vec8;
    typedef typename T::vec4 vec4;
    constexpr int NT = 256;
    constexpr int NPK = C::npk(NT), NPV = C::npv(NT);
    constexpr int BUF_ELEMS = C::K_ELEMS + C::V_ELEMS;
    constexpr bool ONES = (DH % 32) != 0;
    constexpr int ONES_R = ((DH % 32) & 3) + 4 * ((DH % 32) >> 3);
    constexpr int NMFMA = 2 * C::KS + 4 * C::MT;   // MFMAs per region: one QK^T (64 keys) + one P.V
    // softmax scale and shift folded into the QK^T MFMA (see ext_attn_kernel)
    constexpr bool FOLD = FQ && ONES && (C::DKP > DH);
    constexpr int SH_T = DH / 16, SH_HI = (DH % 16) / 8;
    constexpr float FOLD_T = 8.0f;
    // no score bound here: the two query tiles per wave leave no registers for it (250 VGPRs; with the bound 256 and spills
    // inside the loop: 898 against 971 TF/s at cfg4 level 0, profiles/r06_attn_d64_ab.txt)
    constexpr bool BOUND = false;
    constexpr float BOUND_T = std::is_same<E, _Float16>::value ? 14.0f : 60.0f;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    auto sK = [&](int buf) { return reinterpret_cast<E*>(smem) + buf * BUF_ELEMS; };
    auto sV = [&](int buf) { return reinterpret_cast<E*>(smem) + buf * BUF_ELEMS + C::K_ELEMS; };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hi = lane >> 5;
    const int l31 = lane & 31;
    const int K = p.K, Kq = p.Kq, S = p.S, H = p.H;

    const int h = blockIdx.x % H;
    int u = blockIdx.x / H;
    int b, f, qt;
    if constexpr (MODE == MODE_ALL) {
        const int nbank = 2 * Kq * p.nQT;
        if (u < nbank) {
            b = 1 + u / (Kq * p.nQT);
            u -= (b - 1) * Kq * p.nQT;
        } else {
            u -= nbank;
            b = 0;
        }
    } else {
        b = 0;
    }
    f = u / p.nQT;
    qt = u - f * p.nQT;
    const int bq = (p.inject && b > 0) ? 0 : b;
    const int f_lo = b == 0 ? p.q_frame0 + f : 0;
    const int n_fr = b == 0 ? 1 : K;
    const int tpf = (S + 63) >> 6;
    const int ntiles = n_fr * tpf;
    const bool ragged = (S & 63) != 0;

    const E* qg = reinterpret_cast<const E*>(p.q);
    const E* kg = reinterpret_cast<const E*>(p.k) + bq * p.k_bs + h * DH;
    const int64_t vt_row = vt_row_stride(K, p.Spad);
    const E* vg = reinterpret_cast<const E*>(p.vt) + ((int64_t)(b * H + h) * DH) * vt_row;

    // ---- LDS init: everything zero (the pipeline touches Kbuf[1] / Vbuf[1] before they are staged:
    //      P_B(-1) = 0 times V must not meet NaN bits), then the denominator row of both V^T images.
    for (int id = tid; id < 2 * BUF_ELEMS / 8; id += NT) st16(reinterpret_cast<E*>(smem) + id * 8, u32x4{0, 0, 0, 0});
    __syncthreads();
    if constexpr (ONES)
        for (int id = tid; id < 2 * 64; id += NT) sV(id >> 6)[DH * C::VROW + (id & 63)] = (E)1.f;
    if constexpr (FOLD)
        for (int id = tid; id < 2 * 64; id += NT) sK(id >> 6)[(id & 63) * C::KROW + DH] = (E)1.f;

    // ---- Q fragments of both streams
    int q_row[2];
    bool q_ok[2];
    vec8 qf[2][C::KS];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        q_row[qi] = qt * 256 + (wave * 2 + qi) * 32 + l31;
        q_ok[qi] = q_row[qi] < S;
        const E* qp = qg + bq * p.q_bs + f * p.q_fs + (int64_t)(q_ok[qi] ? q_row[qi] : S - 1) * p.ld_q + h * DH;
#pragma unroll
        for (int t = 0; t < C::KS; ++t) {
            const int col = 16 * t + 8 * hi;
            qf[qi][t] = __builtin_bit_cast(vec8, col < DH ? ld16(qp + col) : u32x4{0, 0, 0, 0});
            if constexpr (FOLD) {
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[qi][t][j] = (E)((float)qf[qi][t][j] * p.c);
            }
        }
    }

    float s_bound[2] = {0.f, 0.f};   // BOUND: upper bound of q.k*c (log2 units) over every key this problem sees
    if constexpr (BOUND && !FOLD) {
        const int ppf = p.Spad / 64;
        const float* part = p.knorm2 + ((int64_t)(bq * H + h) * K + f_lo) * ppf;
        float kn2 = 0.f;
        for (int i = lane; i < n_fr * ppf; i += 64) kn2 = fmaxf(kn2, part[i]);
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) kn2 = fmaxf(kn2, __shfl_xor(kn2, o_));
        const float kn = __builtin_sqrtf(kn2) * 1.001f * p.c;
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            float q2 = 0.f;
#pragma unroll
            for (int t = 0; t < C::KS; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) q2 = fmaf((float)qf[qi][t][j], (float)qf[qi][t][j], q2);
            q2 += __shfl_xor(q2, 32);
            s_bound[qi] = __builtin_sqrtf(q2) * kn;
        }
    }

    // ---- staging registers: rk = K(t+1), rv = V(t) while iteration t starts
    // (loads branch-free for the reason given in ext_attn_kernel)
    u32x4 rk[NPK], rv[NPV];
    int k_row[NPK], k_col[NPK], k_goff[NPK], k_loff[NPK], v_goff[NPV], v_loff[NPV];
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
        const int id = min(tid + NT * i, 64 * C::PPR - 1);
        k_row[i] = id / C::PPR;
        k_col[i] = (id - k_row[i] * C::PPR) * 8;
        k_goff[i] = k_row[i] * (int)p.ld + k_col[i];
        k_loff[i] = k_row[i] * C::KROW + k_col[i];
    }
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
        const int id = min(tid + NT * i, DH * 8 - 1);
        v_goff[i] = (id >> 3) * (int)vt_row + (id & 7) * 8;
        v_loff[i] = (id >> 3) * C::VROW + (id & 7) * 8;
    }
    // Tile cursors: K rows and V^T positions of consecutive tiles are 64 apart, except at a frame
    // boundary of a ragged S (the frame's last tile is short in K, padded to Spad in V^T).  Uniform
    // pointer bumps instead of a tile -> (frame, tile-in-frame) division per load.
    const int k_wrap = S - (tpf - 1) * 64, v_wrap = p.Spad - (tpf - 1) * 64;
    const int64_t k_wrap_off = p.k_fs - (int64_t)(tpf - 1) * 64 * p.ld;
    const E* k_next = kg + f_lo * p.k_fs;   // first row of the next K tile to load
    const E* v_next = vg + (int64_t)f_lo * p.Spad;
    int k_tt = 0, v_tt = 0;                            // its tile index within the frame
    auto load_k = [&]() {
        const bool wrap = k_tt == tpf - 1;
        const int rlim = wrap ? k_wrap - 1 : 63;
        const int clamp_off = rlim * (int)p.ld;
#pragma unroll
        for (int i = 0; i < NPK; ++i) rk[i] = ld16(k_next + (k_row[i] <= rlim ? k_goff[i] : clamp_off + k_col[i]));
        k_next += wrap ? k_wrap_off : (int64_t)64 * p.ld;
        k_tt = wrap ? 0 : k_tt + 1;
    };
    auto load_v = [&]() {
#pragma unroll
        for (int i = 0; i < NPV; ++i) rv[i] = ld16(v_next + v_goff[i]);
        const bool wrap = v_tt == tpf - 1;
        v_next += wrap ? v_wrap : 64;
        v_tt = wrap ? 0 : v_tt + 1;
    };
    auto write_k = [&](int buf) {
        E* kb = sK(buf);
#pragma unroll
        for (int i = 0; i < NPK; ++i)
            if (tid + NT * i < 64 * C::PPR) st16(kb + k_loff[i], rk[i]);
    };
    auto write_v = [&](int buf) {
        E* vb = sV(buf);
#pragma unroll
        for (int i = 0; i < NPV; ++i)
            if (tid + NT * i < DH * 8) st16(vb + v_loff[i], rv[i]);
    };

    f32x16 o[2][C::MT], s[2][2];
    vec8 pf[2][4];
    float m_run[2], l_run[2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        m_run[qi] = FOLD ? 0.f : -INFINITY;   // FOLD: the current shift
        l_run[qi] = 0.f;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qi][mt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[qi][ks][j] = (E)0.f;
    }
    const float c = p.c;

    // S^T (64 keys x 32 queries) of stream qi from K buffer `buf`
    auto qk = [&](auto qi_c, int buf) {
        constexpr int qi = decltype(qi_c)::value;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[qi][kt][r] = 0.f;
            const E* krow = sK(buf) + (kt * 32 + l31) * C::KROW + 8 * hi;
#pragma unroll
            for (int t = 0; t < C::KS; ++t)
                s[qi][kt] = T::mfma32(__builtin_bit_cast(vec8, ld16(krow + 16 * t)), qf[qi][t], s[qi][kt]);
        }
    };
    // O^T += V^T . P of stream qi from V buffer `buf`
    auto pv = [&](auto qi_c, int buf) {
        constexpr int qi = decltype(qi_c)::value;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
            const E* vrow = sV(buf) + (mt * 32 + l31) * C::VROW + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                o[qi][mt] = T::mfma32(__builtin_bit_cast(vec8, ld16(vrow + 16 * ks)), pf[qi][ks], o[qi][mt]);
        }
    };
    // first half of the online softmax: mask, row max, (rare) rescale.  Returns m*c.
    auto sm_head = [&](auto qi_c, int tile, bool first) -> float {   // tile = index within the frame
        constexpr int qi = decltype(qi_c)::value;

        if (ragged) {
            const int tt = tile - (tile / tpf) * tpf;
            if (tt == tpf - 1) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (tt * 64 + kt * 32 + cd_row(r, hi) >= S) s[qi][kt][r] = -INFINITY;
            }
        }
        auto tile_max = [&]() {
            float mx = s[qi][0][0];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qi][kt][r]);
            return max_with_lane_xor32(mx);
        };
        if constexpr (BOUND && !FOLD) {
            // deferred shift under the score bound (see BOUND in ext_attn_kernel): the tile maximum is looked at only
            // while the bound does not exclude an overflow; m_run = -inf before the first tile, which always looks
            if (__any(s_bound[qi] - m_run[qi] * c > BOUND_T)) {
                const float mx = tile_max();
                const bool over = (mx - m_run[qi]) * c > BOUND_T;
                if (__any(over)) {
                    const float m_new = over ? mx : m_run[qi];
                    const float alpha = __builtin_amdgcn_exp2f((m_run[qi] - m_new) * c);   // exp2(-inf) = 0 on tile 0 (O = 0)
                    m_run[qi] = m_new;
                    if constexpr (!ONES) l_run[qi] *= alpha;
#pragma unroll
                    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[qi][mt][r] *= alpha;
                }
            }
            return m_run[qi] * c;
        }
        const float mx = tile_max();
        if constexpr (FOLD) {
            // s already is score*c - shift: move the shift only on the first tile or when the tile maximum
            // exceeds it by more than FOLD_T (wave-uniform branch)
            if (first || __any(mx > FOLD_T)) {
                const float sh_old = m_run[qi];
                const float sh_new = (first || mx > FOLD_T) ? (float)(E)(sh_old + mx) : sh_old;
                const float delta = sh_new - sh_old;
                const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);   // O == 0 on the first tile
                m_run[qi] = sh_new;
                if (hi == SH_HI) qf[qi][SH_T][0] = (E)(-sh_new);
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qi][mt][r] *= alpha;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[qi][kt][r] -= delta;
            }
            return 0.f;
        }
        const bool over = mx > m_run[qi] + TF_ATTN_LAG / c;   // lagged reference point (see ext_attn_il_kernel), per query
        if (__any(over)) {
            const float m_new = over ? mx : m_run[qi];
            const float alpha = __builtin_amdgcn_exp2f((m_run[qi] - m_new) * c);
            m_run[qi] = m_new;
            if constexpr (!ONES) l_run[qi] *= alpha;
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qi][mt][r] *= alpha;
        }
        return m_run[qi] * c;
    };
    // One overlapped region: stream X finishes its softmax (P = exp2(s*c - m*c), rounded to the MFMA
    // input type) on the VALU while stream Y = 1-X runs O_Y += V P_Y (vbuf) and S_Y = K Q_Y (kbuf) on
    // the matrix pipe.  The region is cut into NMFMA steps, each = { LDS fragment read for step i+2,
    // MFMA i, its share of the 16 (pk_fma, 2 exp, cvt_pk) softmax units }, and a sched_barrier(0)
    // after every step pins that order: the wave's in-order issue then alternates matrix and vector work.
    auto region = [&](auto x_c, float mc, int vbuf, int kbuf) {
        constexpr int X = decltype(x_c)::value;
        constexpr int Y = 1 - X;
        constexpr PpSchedule<C::MT, C::KS> sch{};
#ifndef TF_TUNE_PP_PF
#define TF_TUNE_PP_PF 4
#endif
        constexpr int PF = TF_TUNE_PP_PF;   // fragment reads run PF steps ahead of their MFMA (LDS latency)
        float lsum = 0.f;
        const E* vbase = sV(vbuf) + l31 * C::VROW + 8 * hi;
        const E* kbase = sK(kbuf) + l31 * C::KROW + 8 * hi;
        auto frag = [&](int i) -> vec8 {
            if (sch.is_pv[i]) return __builtin_bit_cast(vec8, ld16(vbase + sch.chain[i] * 32 * C::VROW + 16 * sch.kstep[i]));
            return __builtin_bit_cast(vec8, ld16(kbase + sch.chain[i] * 32 * C::KROW + 16 * sch.kstep[i]));
        };
        vec8 fr[NMFMA];
#pragma unroll
        for (int i = 0; i < PF; ++i) fr[i] = frag(i);
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NMFMA; ++i) {
            if (i + PF < NMFMA) fr[i + PF] = frag(i + PF);
            if (sch.is_pv[i]) {
                o[Y][sch.chain[i]] = T::mfma32(fr[i], pf[Y][sch.kstep[i]], o[Y][sch.chain[i]]);
            } else {
                s[Y][sch.chain[i]] =
                    T::mfma32(fr[i], qf[Y][sch.kstep[i]], sch.kstep[i] == 0 ? zero : s[Y][sch.chain[i]]);
            }
#pragma unroll
            for (int un = (i * 16) / NMFMA; un < ((i + 1) * 16) / NMFMA; ++un) {
                const int kt = un >> 3, r = (un & 7) * 2;
                // two scalar v_fma_f32, NOT one v_pk_fma_f32: packed f32 VALU beside MFMAs costs ~+22 cycles each
                const float p0 = __builtin_amdgcn_exp2f(FOLD ? s[X][kt][r] : fmaf(s[X][kt][r], c, -mc));
                const float p1 = __builtin_amdgcn_exp2f(FOLD ? s[X][kt][r + 1] : fmaf(s[X][kt][r + 1], c, -mc));
                if constexpr (!ONES) lsum += p0 + p1;
                pf[X][kt * 2 + (r >> 3)][r & 7] = (E)p0;
                pf[X][kt * 2 + (r >> 3)][(r & 7) + 1] = (E)p1;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!ONES) l_run[X] += lsum;
        // P_X must exist HERE: an empty asm with the registers as read-write operands keeps the compiler
        // from sinking the (register-only) softmax past the next barrier, next to its consumer
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(pf[X][ks]));
    };
    typedef std::integral_constant<int, 0> A;
    typedef std::integral_constant<int, 1> B;

    // ---- prologue: K(0) -> Kbuf[0]; S_A(0); registers <- K(1), V(0)
    load_k();                   // K(0)
    __syncthreads();            // LDS init done before the first staging write
    write_k(0);
    if (ntiles > 1) load_k();   // K(1)   (with a single tile rk keeps K(0): written to Kbuf[1], read by a dead S_A(1))
    load_v();                   // V(0)
    __syncthreads();
    qk(A{}, 0);

    int tt = 0;   // tile index of t within its frame
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        // ================= R1(t) =================
        // Kbuf[nxt] held K(t-1) (last read in R1(t-1)), Vbuf[cur] held V(t-2) (last read in R1(t-1)):
        // every wave has passed the barrier of iteration t-1, which follows R1(t-1) -> free to overwrite.
        write_k(nxt);   // K(t+1)
        write_v(cur);   // V(t)
        if (t + 2 < ntiles) load_k();   // K(t+2)
        if (t + 1 < ntiles) load_v();   // V(t+1)
        // P_A(t) (VALU)  ||  O_B += V(t-1) P_B(t-1) from Vbuf[(t-1)&1],  S_B(t) = K(t) Q_B from Kbuf[t&1] (MFMA)
        region(A{}, sm_head(A{}, tt, t == 0), nxt, cur);
        __syncthreads();   // K(t+1), V(t) visible to all waves; all waves done with R1(t)
        __builtin_amdgcn_sched_barrier(0);
        // ================= R2(t) =================
        // P_B(t) (VALU)  ||  O_A += V(t) P_A(t) from Vbuf[t&1],  S_A(t+1) = K(t+1) Q_A from Kbuf[(t+1)&1]
        // (a dead tile after the last t) (MFMA)
        region(B{}, sm_head(B{}, tt, t == 0), cur, nxt);
        tt = tt == tpf - 1 ? 0 : tt + 1;
    }
    pv(B{}, (ntiles - 1) & 1);         // drain: O_B += V(n-1) P_B(n-1)

    // ---- epilogue
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        float l_tot;
        if constexpr (ONES)
            l_tot = __shfl(o[qi][C::MT - 1][ONES_R], l31);
        else
            l_tot = l_run[qi] + __shfl_xor(l_run[qi], 32);
        const float inv_l = 1.0f / l_tot;
        if (q_ok[qi]) {
            const int64_t op = b * p.o_bs + f * p.o_fs + (int64_t)q_row[qi] * (H * DH) + h * DH;
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int d0 = mt * 32 + 8 * rg + 4 * hi;
                    if (d0 < DH) {
                        f32x4 w;
#pragma unroll
                        for (int i = 0; i < 4; ++i) w[i] = o[qi][mt][rg * 4 + i] * inv_l;
                        store_out4<E, vec4>(p.out, op + d0, w, p.out_f32);
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Half-tile interleaved variant (fp32 score scaling): Dh = 40 (the cfg2 / cfg3 level-0 form) and 80.
//
// The plain kernel runs QK^T -> softmax (64 VALU) -> P.V of one 64-key tile back to back: inside a wave the matrix
// pipe idles during the softmax and the VALU during the MFMAs, and the overlap that independent waves on a SIMD
// provide stops at ~49 % matrix-pipe utilisation at Dh = 40 -- and does not exist at all where the
// registers of the larger head dims leave two waves per SIMD.
// Here ONE query tile per wave is software-pipelined over 32-key half tiles, so that every stretch of the
// instruction stream has INDEPENDENT matrix and vector work, issued alternately (one MFMA, its share of the softmax,
// pinned by sched_barrier(0)):
//     phase 1 of tile t:  O += V0(t) P0(t)  and  S0(t+1) = K0(t+1) Q   (2 MT + KS MFMAs)  ||  P1(t)   = exp2(S1(t) c - m c)
//     phase 2 of tile t:  O += V1(t) P1(t)  and  S1(t+1) = K1(t+1) Q                       ||  P0(t+1) = exp2(S0(t+1) c - m c)
// (7 MFMAs per phase at Dh = 40, 11 at 80, against the same 8 softmax units of 2 fma + 2 exp + 1 cvt.)
// Same LDS images as the plain kernel, K staged one tile earlier (as in the ping-pong kernel): top of iteration t
// writes K(t+1) and V(t) from registers loaded an iteration before, one barrier, then the two phases.
// Online softmax: Dh = 40 uses the score bound (BOUND: a half tile looks at its maximum only when the bound does not
// exclude an overflow); the other head dims take the half tile's maximum every time (they are matrix-bound).  When
// the shift moves, O -- which by then includes the P.V of the half tile that ran beside the softmax, computed against
// the OLD shift -- is rescaled at the END of the phase, before any P at the new shift is multiplied in
// (cdna_hip_programming.md T13: scale everything still at the old maximum exactly once).
// Scope: S a multiple of 64, MODE_ALL / MODE_SOURCE problems (the dual-V form has its own kernel); the split form
// of small grids (runs of bank frames + attn_merge_kernel) as in ext_attn_kernel.
template <int MT, int KS, bool NEXT>
struct IlSchedule {   // MFMA order of one phase: QK^T k-steps (one accumulator chain) alternate with the P.V MFMAs
    static constexpr int N = (NEXT ? KS : 0) + 2 * MT;   // (M-tile round-robin, 2 k-steps): never two MFMAs on one
    int is_pv[N] = {}, a[N] = {}, b[N] = {};              // accumulator next to each other
    constexpr IlSchedule() {
        int i = 0, qk = 0, pv = 0;
        while (i < N) {
            if (NEXT && qk < KS) {
                is_pv[i] = 0, a[i] = qk, b[i] = 0;
                ++qk, ++i;
            }
            if (pv < 2 * MT) {
                is_pv[i] = 1, a[i] = pv % MT, b[i] = pv / MT;   // a = M-tile, b = 16-key k-step of the half
                ++pv, ++i;
            }
        }
    }
};

// MIXED MFMA shapes (Dh = 40, one bank, round 6; off: TF_TUNE_NO_IL40_MIX): QK^T stays 32x32x16 (K = 48), P.V runs as 16x16x32 MFMAs over
// THREE 16-row M-tiles (rows 0-47 of the same V^T image: 40 features, the ones row, 7 zero rows) and the two 16-query halves
// of the wave's tile: 6 short MFMAs (16 clocks each) per 32-key half instead of 4 long ones -- 192 instead of 224 matrix-pipe
// clocks per phase.  A step's fragment is read once per M-tile (fidx: the step whose LDS fragment this step multiplies).
template <int NT, int KS, bool NEXT>
struct IlScheduleMix {   // NT = 16-row M-tiles of P.V (3 at Dh = 40: 48 rows; 4 at Dh = 64)
    static constexpr int N = (NEXT ? KS : 0) + 2 * NT;
    int is_pv[N] = {}, a[N] = {}, b[N] = {}, fidx[N] = {};   // P.V: a = M-tile, b = 16-query half; QK^T: a = k-step
    constexpr IlScheduleMix() {
        int i = 0;
        if (NEXT) {
            // QK0 PV00 PV01 | QK1 PV10 PV11 | QK2 PV20 PV21 ...: the QK^T chain's links lie two short MFMAs (32 clocks) apart
            for (int d = 0; d < NT; ++d) {
                if (d < KS) {
                    is_pv[i] = 0, a[i] = d, fidx[i] = i;
                    ++i;
                }
                is_pv[i] = 1, a[i] = d, b[i] = 0, fidx[i] = i;
                is_pv[i + 1] = 1, a[i + 1] = d, b[i + 1] = 1, fidx[i + 1] = i;
                i += 2;
            }
            for (int t = NT; t < KS; ++t) {
                is_pv[i] = 0, a[i] = t, fidx[i] = i;
                ++i;
            }
        } else {
            // PV00 PV10 PV01 PV11 | PV20 PV21 | PV30 PV31: the first two steps own their fragments (cross-phase prefetch hands over two)
            const int dd[4] = {0, 1, 0, 1}, tt[4] = {0, 0, 1, 1}, ff[4] = {0, 1, 0, 1};
            for (i = 0; i < 4; ++i) is_pv[i] = 1, a[i] = dd[i], b[i] = tt[i], fidx[i] = ff[i];
            for (int d = 2; d < NT; ++d) {
                is_pv[i] = 1, a[i] = d, b[i] = 0, fidx[i] = i;
                is_pv[i + 1] = 1, a[i + 1] = d, b[i + 1] = 1, fidx[i + 1] = i;
                i += 2;
            }
        }
    }
};

// DMA != 0 (non-PACK forms): K and V^T tiles go global -> LDS by `global_load_lds_dwordx4` instead of through registers: no
// staging VGPRs, no ds_write pass.  The DMA writes lane-linearly (wave-uniform LDS base + lane * 16 B per instruction, a
// "piece" of 1 KB), the per-lane SOURCE address is free, so any LDS image whose 16-B slots are filled piece by piece works:
//   DMA = 1 (round 5, Dh = 40; A/B switch TF_TUNE_IL40_DMA, OFF: 0.5 % slower than register staging at cfg2 level 0,
//           profiles/r05_attn_il40_dma_ab.txt): DENSE images, K rows of DH elements (the QK^T k-step that straddles DH reads the
//           next row's first elements against ZERO columns of Q), V^T rows of 64 keys XOR-swizzled with (row & 7) on the source
//           address and on the fragment read -- one address computation per fragment read;
//   DMA = 2 (round 6): the PADDED images of the register-staged form (row strides of an odd number of 16-B slots), so every
//           fragment address stays "per-lane base + immediate".  A lane whose slot is row padding fetches slot 0 of its row
//           (finite data: the K pad columns meet zero columns of Q, the V^T pad columns are never read); the lanes of a last,
//           partial piece past the end of the image are masked off (the constant rows behind it must survive).
// A tile is issued right behind the barrier that frees its buffer and drained (vmcnt(0)) in front of the next one: the same
// distance the register staging had.
template <typename T, int DH, int NW, int MODE, int MINW, int DMA = 0>
__global__ __launch_bounds__(64 * NW, MINW) void ext_attn_il_kernel(AttnParams p) {
    typedef AttnCfg<DH, 64> C;
    typedef typename T::elem E;
    typedef typename T::vec8 vec8;
    typedef typename T::vec4 vec4;
    constexpr int NT = 64 * NW;
    // MODE_DUAL (q/k injection, Dh = 40): uncond and cond share q, k, the scores and P; the two banks' V^T rows are
    // packed into ONE LDS image of 3 M-tiles (rows 0-39 uncond, 40-79 cond, row 80 the common ones row) exactly as in
    // ext_attn_kernel's PACK form, so the only differences to the single-bank kernel are the number of staged V^T rows
    // (VR), the number of P.V M-tiles (MT) and the epilogue's row -> (bank, feature) decode.
    constexpr bool PACK = MODE == MODE_DUAL;
    constexpr bool MIX = DMA == 3;   // DMA = 3: the DMA = 2 staging + mixed MFMA shapes, see IlScheduleMix
    static_assert(!MIX || ((DH == 40 || DH == 64) && !PACK), "mixed MFMA shapes: Dh = 40 or 64, one bank");
    constexpr int NT16 = DH == 40 ? 3 : (DH + 15) / 16;   // MIX: 16-row M-tiles of P.V (Dh = 40: features + the ones row + 7 zero rows)
#ifdef TF_TUNE_IL40_MIX_SWZ
    constexpr bool MIXSWZ = MIX;   // slot swizzle of the mixed form's V^T image: conflict-free and 1 % SLOWER, see dv_goff below
#else
    constexpr bool MIXSWZ = false;
#endif
    static_assert(!PACK || DH == 40 || DH == 64 || DH == 80,
                  "the packed dual-V image: Dh = 40 (3 M-tiles, ones row 80), Dh = 64 (4 full M-tiles) or Dh = 80 (5 full M-tiles)");
    constexpr int VR = PACK ? 2 * DH : DH;              // staged V^T rows per tile
    constexpr int MT = PACK ? (2 * DH + 31) / 32 : C::MT;   // P.V M-tiles
    static_assert(DMA != 1 || (!PACK && (64 * DH * 2) % 1024 == 0 && (VR * 128) % 1024 == 0), "dense DMA form: whole 1 KB pieces");
    static_assert(DMA != 1 || !PACK, "the dense DMA form stages one bank");
    constexpr int KROW = DMA == 1 ? DH : C::KROW;       // LDS row strides (elements): dense images in the DMA = 1 form
    constexpr int VROW = DMA == 1 ? 64 : C::VROW;
    constexpr int K_ELEMS = 64 * KROW;
    constexpr int V_ELEMS = MT * 32 * VROW;
    constexpr int NPK = C::npk(NT), NPV = (VR * 8 + NT - 1) / NT;
    constexpr int BUF_ELEMS = K_ELEMS + V_ELEMS;
    constexpr bool ONES = (VR % 32) != 0;   // denominator from the MFMA (row VR of the V^T image = 1.0)
    constexpr int ONES_R = ((VR % 32) & 3) + 4 * ((VR % 32) >> 3);
    static_assert(!ONES || ((VR % 32) & 4) == 0, "the ones row must live in lane half 0");
    constexpr bool BOUND = attn_has_bound(DH);   // needs the key norms of the pre-pass
    constexpr float BOUND_T = std::is_same<E, _Float16>::value ? 14.0f : 60.0f;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    auto sK = [&](int buf) { return reinterpret_cast<E*>(smem) + buf * BUF_ELEMS; };
    auto sV = [&](int buf) { return reinterpret_cast<E*>(smem) + buf * BUF_ELEMS + K_ELEMS; };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hi = lane >> 5;
    const int l31 = lane & 31;
    const int K = p.K, Kq = p.Kq, S = p.S, H = p.H;

    const int h = blockIdx.x % H;
    int u = blockIdx.x / H;
    int b, f, qt;
    int seg = 0;   // split form (small grids): run of bank frames this workgroup covers, see ext_attn_kernel
    const int nseg = MODE == MODE_SOURCE ? 1 : p.nseg;
    if constexpr (MODE == MODE_ALL) {
        const int nbank = 2 * Kq * p.nQT * nseg;
        if (u < nbank) {
            seg = u % nseg;
            u /= nseg;
            b = 1 + u / (Kq * p.nQT);
            u -= (b - 1) * Kq * p.nQT;
        } else {
            u -= nbank;
            b = 0;
        }
    } else if constexpr (MODE == MODE_DUAL) {
        b = 1;
        seg = u % nseg;
        u /= nseg;
    } else {
        b = 0;
    }
    f = u / p.nQT;
    qt = u - f * p.nQT;
    const int bq = (p.inject && b > 0) ? 0 : b;
    const bool split = nseg > 1 && b > 0;
    const int f_lo = b == 0 ? p.q_frame0 + f : (seg * K) / nseg;
    const int n_fr = b == 0 ? 1 : ((seg + 1) * K) / nseg - f_lo;
    const int tpf = S >> 6;
    const int ntiles = n_fr * tpf;

    const E* qg = reinterpret_cast<const E*>(p.q);
    const E* kg = reinterpret_cast<const E*>(p.k) + bq * p.k_bs + h * DH;
    const int64_t vt_row = vt_row_stride(K, p.Spad);
    const E* vg = reinterpret_cast<const E*>(p.vt) + ((int64_t)(b * H + h) * DH) * vt_row;

    // ---- LDS init: zero everything (pads; the V^T rows past DH), then the denominator row DH of both V^T images
    for (int id = tid; id < 2 * BUF_ELEMS / 8; id += NT) st16(reinterpret_cast<E*>(smem) + id * 8, u32x4{0, 0, 0, 0});
    __syncthreads();
    if constexpr (ONES)
        for (int id = tid; id < 2 * 64; id += NT) sV(id >> 6)[VR * VROW + (id & 63)] = (E)1.f;

    // ---- Q fragments
    const int q_row = qt * (32 * NW) + wave * 32 + l31;
    const bool q_ok = q_row < S;
    vec8 qf[C::KS];
    {
        const E* qp = qg + bq * p.q_bs + f * p.q_fs + (int64_t)(q_ok ? q_row : S - 1) * p.ld_q + h * DH;
#pragma unroll
        for (int t = 0; t < C::KS; ++t) {
            const int col = 16 * t + 8 * hi;
            qf[t] = __builtin_bit_cast(vec8, col < DH ? ld16(qp + col) : u32x4{0, 0, 0, 0});
        }
    }
    const float c = p.c;
    // score bound (log2 units) over every key this problem sees: |q| max|k| c  (see BOUND in ext_attn_kernel)
    float s_bound = 0.f;
    if constexpr (BOUND) {
        const int ppf = p.Spad / 64;
        const float* part = p.knorm2 + ((int64_t)(bq * H + h) * K + f_lo) * ppf;
        float kn2 = 0.f;
        for (int i = lane; i < n_fr * ppf; i += 64) kn2 = fmaxf(kn2, part[i]);
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) kn2 = fmaxf(kn2, __shfl_xor(kn2, o_));
        float q2 = 0.f;
#pragma unroll
        for (int t = 0; t < C::KS; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) q2 = fmaf((float)qf[t][j], (float)qf[t][j], q2);
        q2 += __shfl_xor(q2, 32);
        s_bound = __builtin_sqrtf(q2) * __builtin_sqrtf(kn2) * 1.001f * c;
    }

    // ---- staging: 16-B pieces of K and of V^T per thread and tile (branch-free, see ext_attn_kernel)
    u32x4 rk[NPK], rv[NPV];
    int k_goff[NPK], k_loff[NPK], v_goff[NPV], v_loff[NPV];
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
        const int id = min(tid + NT * i, 64 * C::PPR - 1);
        k_goff[i] = (id / C::PPR) * (int)p.ld + (id % C::PPR) * 8;
        k_loff[i] = (id / C::PPR) * KROW + (id % C::PPR) * 8;
    }
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
        const int id = min(tid + NT * i, VR * 8 - 1);
        const int row = id >> 3;   // image row: bank row / DH (the next branch's rows lie H*DH image rows further), feature row % DH
        v_goff[i] = ((row / DH) * H * DH + row % DH) * (int)vt_row + (id & 7) * 8;
        v_loff[i] = row * VROW + (id & 7) * 8;
    }
    const int v_wrap = p.Spad - (tpf - 1) * 64;
    const int64_t k_wrap_off = p.k_fs - (int64_t)(tpf - 1) * 64 * p.ld;
    const E* k_next = kg + f_lo * p.k_fs;
    const E* v_next = vg + (int64_t)f_lo * p.Spad;
    int k_tt = 0, v_tt = 0;
    auto load_k = [&]() {
#pragma unroll
        for (int i = 0; i < NPK; ++i) rk[i] = ld16(k_next + k_goff[i]);
        const bool wrap = k_tt == tpf - 1;
        k_next += wrap ? k_wrap_off : (int64_t)64 * p.ld;
        k_tt = wrap ? 0 : k_tt + 1;
    };
    auto load_v = [&]() {
#pragma unroll
        for (int i = 0; i < NPV; ++i) rv[i] = ld16(v_next + v_goff[i]);
        const bool wrap = v_tt == tpf - 1;
        v_next += wrap ? v_wrap : 64;
        v_tt = wrap ? 0 : v_tt + 1;
    };
    auto write_k = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NPK; ++i)
            if (tid + NT * i < 64 * C::PPR) st16(sK(buf) + k_loff[i], rk[i]);
    };
    auto write_v = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NPV; ++i)
            if (tid + NT * i < VR * 8) st16(sV(buf) + v_loff[i], rv[i]);
    };
    // DMA forms: a tile = NKP 1-KB pieces of the K image + NVP of the V^T image; wave w issues pieces w, w + NW, .. of each.
    // Per-lane source offsets are unsigned BYTE offsets from the wave-uniform tile pointers, so that the DMA takes the
    // SGPR-base + 32-bit-VGPR-offset form (no 64-bit address pair per lane)
    constexpr int K_IMG = 64 * KROW * 2, V_IMG = VR * VROW * 2;   // staged bytes of one K / V^T image
    constexpr int NKP = DMA ? (K_IMG + 1023) / 1024 : 0, NVP = DMA ? (V_IMG + 1023) / 1024 : 0;
    constexpr int NKS = DMA ? (NKP + NW - 1) / NW : 1, NVS = DMA ? (NVP + NW - 1) / NW : 1;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    uint32_t dk_goff[NKS], dv_goff[NVS];
    bool dk_ok[NKS], dv_ok[NVS];   // this lane's slot lies inside the image (false only in a last, partial piece)
    if constexpr (DMA != 0) {
#pragma unroll
        for (int n = 0; n < NKS; ++n) {
            const int o = (wave_u + NW * n) * 1024 + lane * 16;
            const int row = min(o / (2 * KROW), 63);
            const int pc = (o - row * 2 * KROW) >> 4;          // 16-B slot of the LDS row this lane fills
            dk_ok[n] = o < K_IMG;
            dk_goff[n] = (uint32_t)(row * (int)p.ld + ((pc < DH / 8 ? pc : 0) << 3)) * 2u;
        }
#pragma unroll
        for (int n = 0; n < NVS; ++n) {
            const int o = (wave_u + NW * n) * 1024 + lane * 16;
            const int row = min(o / (2 * VROW), VR - 1);
            const int sl = (o - row * 2 * VROW) >> 4;
            dv_ok[n] = o < V_IMG;
            // image row -> V^T row: bank row / DH (the next branch's rows lie H*DH V^T rows further), feature row % DH
            const int vrow = PACK ? (row / DH) * H * DH + row % DH : row;
            // MIX: ds_read_b128 is serviced in four NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...:
            // MI355X_MICROARCH.md, LDS table), so a group of the mixed form's V^T reads takes rows 0-3 / 12-15 from one 16-lane row
            // of the wave (one k-block = one slot of the image row) and rows 4-11 from the next one (the slot two further): 2-way
            // bank conflicts (SQ_LDS_BANK_CONFLICT 33 % of the LDS-active clocks).  Rows 4-11 of every 16 therefore store their
            // slots with bit 1 flipped (k-blocks of k-steps 0 and 1 exchanged) -- applied here, on the DMA's source side, and in
            // the fragment read: every service group then reads ONE slot position of 16 different rows.  MEASURED: the conflicts go
            // (SQ_LDS_BANK_CONFLICT 0, LDS-active clocks 643 M -> 430 M per launch) and the launch gets 1 % SLOWER (3.61 against
            // 3.57 ms, three alternations on one box, profiles/r06_attn_d40_mix_ab.txt section 8): the LDS is not what this kernel
            // waits for, and the chip is power-limited.  Off by default (TF_TUNE_IL40_MIX_SWZ).
            const int slm = (MIXSWZ && ((row + 4) & 8)) ? (sl ^ 2) : sl;
            dv_goff[n] = (uint32_t)(vrow * (int)vt_row + ((DMA == 1 ? sl ^ (row & 7) : sl < 8 ? slm : 0) << 3)) * 2u;
        }
    }
    auto dma_k = [&](int buf) {      // the next K tile -> Kbuf[buf]
#pragma unroll
        for (int n = 0; n < NKS; ++n) {
            const int q = wave_u + NW * n;
            if (NKP % NW == 0 || q < NKP) {
                uint32_t off = dk_goff[n];
                asm volatile("" : "+v"(off));   // keeps the zero-extension next to the add: SGPR base + 32-bit VGPR offset form
                if (K_IMG % 1024 == 0 || dk_ok[n])
                    __builtin_amdgcn_global_load_lds((glb_ptr)(reinterpret_cast<const char*>(k_next) + off),
                                                     (lds_ptr)(sK(buf) + q * 512), 16, 0, 0);
            }
        }
        const bool wrap = k_tt == tpf - 1;
        k_next += wrap ? k_wrap_off : (int64_t)64 * p.ld;
        k_tt = wrap ? 0 : k_tt + 1;
    };
    auto dma_v = [&](int buf) {      // the next V^T tile -> Vbuf[buf]
#pragma unroll
        for (int n = 0; n < NVS; ++n) {
            const int q = wave_u + NW * n;
            if (NVP % NW == 0 || q < NVP) {
                uint32_t off = dv_goff[n];
                asm volatile("" : "+v"(off));
                if (V_IMG % 1024 == 0 || dv_ok[n])
                    __builtin_amdgcn_global_load_lds((glb_ptr)(reinterpret_cast<const char*>(v_next) + off),
                                                     (lds_ptr)(sV(buf) + q * 512), 16, 0, 0);
            }
        }
        const bool wrap = v_tt == tpf - 1;
        v_next += wrap ? v_wrap : 64;
        v_tt = wrap ? 0 : v_tt + 1;
    };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    f32x16 o[MT], s[2];
    f32x4 o16[NT16][2]; // MIX: O^T as [16-row M-tile][16-query half]: lane l = query l & 15 of the half, rows 4 (l >> 4) + i
    vec8 pf[2][2];      // P of the two 32-key halves, two 16-key k-steps each (MIX: after p_relayout, the two 16-query halves)
    float m_run = -INFINITY;   // BOUND: deferred shift; else the lagged running maximum (raw-score units)
    const float lag = TF_ATTN_LAG / c;   // raw-score units
    float l_run = 0.f;         // !ONES: this lane's share of the denominator
    // !ONES (Dh = 64: both P.V M-tiles are full, no spare row for the denominator): the row sum on the MATRIX pipe.  The 32 v_add of
    // a tile were 2.0 of the loop's 7.8 VALU instructions per MFMA, on an issue port that is the kernel's limiter
    // (profiles/r06_d64_accounting.md); v_mfma_f32_4x4x4 with A = ones adds the 4 rounded P values of a lane's register pair
    // to a lane-local fp32 sum -- 8 short MFMAs (8 clocks of the pipe each) per tile, and the denominator sums exactly the
    // rounded P the numerator multiplies.  TF_TUNE_IL_LSUM_VALU: the v_add form.
#ifndef TF_TUNE_IL_LSUM_VALU
    constexpr bool LSUM_MFMA = !ONES;
#else
    constexpr bool LSUM_MFMA = false;
#endif
    f32x4 lacc = {0.f, 0.f, 0.f, 0.f};
    vec4 ones4;
#pragma unroll
    for (int j = 0; j < 4; ++j) ones4[j] = (E)1.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[mt][r] = 0.f;
#pragma unroll
    for (int d = 0; d < NT16; ++d)
#pragma unroll
        for (int t = 0; t < 2; ++t) o16[d][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[kt][ks][j] = (E)0.f;
    // MIX: P of half X from the 32x32 accumulator layout (lane = query l & 31; pf[X][0] = accumulator registers 0-7, pf[X][1] =
    // 8-15) to the 16x16x32 B layout (lane = query l & 15 of a 16-query half, 8 keys per 16-lane row).  v_permlane16_swap
    // exchanges the odd 16-lane rows of its first operand with the even rows of its second: afterwards pf[X][0] holds, in
    // rows 0 / 1 / 2 / 3, registers 0-7 | 8-15 of lane half 0 and 0-7 | 8-15 of lane half 1 of queries 0-15, pf[X][1] the same of
    // queries 16-31 -- the k order (row g: accumulator registers 8 (g & 1) .. +7 of lane half g >> 1) is the one the V^T
    // fragment read of the mixed form uses.
    auto p_relayout = [&](auto x_c) {
        constexpr int X = decltype(x_c)::value;
        u32x4 a = __builtin_bit_cast(u32x4, pf[X][0]), b2 = __builtin_bit_cast(u32x4, pf[X][1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const auto r = __builtin_amdgcn_permlane16_swap(a[i], b2[i], false, false);
            a[i] = r[0];
            b2[i] = r[1];
        }
        pf[X][0] = __builtin_bit_cast(vec8, a);
        pf[X][1] = __builtin_bit_cast(vec8, b2);
    };
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // decision part of the online softmax of half X: returns alpha (1 = no move) and leaves m_run updated
    auto sm_decide = [&](auto x_c, bool& move) -> float {
        constexpr int X = decltype(x_c)::value;
        move = false;
        float alpha = 1.f;
        auto half_max = [&]() {
            float mx = s[X][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[X][r]);
            return max_with_lane_xor32(mx);
        };
        if constexpr (BOUND) {
            if (__any(s_bound - m_run * c > BOUND_T)) {
                const float mx = half_max();
                const bool over = (mx - m_run) * c > BOUND_T;
                if (__any(over)) {
                    move = true;
                    const float m_new = over ? mx : m_run;
                    alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);   // exp2(-inf) = 0 on the first half tile (O is 0)
                    m_run = m_new;
                }
            }
        } else {
            // m_run = the query's reference point: it follows the running maximum with a lag of TF_ATTN_LAG binades (P <= 2^8,
            // in range for f16 too).  With the exact maximum a wave of 32 queries rescaled O on ~40 % of its half tiles
            // (some query almost always sees a new maximum); per-query decision: alpha = 1 exactly where it did not move.
            const float mx = half_max();
            const bool over = mx > m_run + lag;   // -inf + lag = -inf: the first half tile always sets the reference
            if (__any(over)) {
                move = true;
                const float m_new = over ? mx : m_run;
                alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
                m_run = m_new;
            }
        }
        return alpha;
    };
    auto rescale = [&](float alpha) {
        if constexpr (MIX) {
            // alpha belongs to query l & 31; O^T holds queries l & 15 (half 0) and 16 + (l & 15) (half 1): one row swap delivers both
            const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(alpha), __float_as_uint(alpha), false, false);
            const float a0 = __uint_as_float(r[0]), a1 = __uint_as_float(r[1]);
#pragma unroll
            for (int d = 0; d < NT16; ++d) {
                o16[d][0] *= a0;
                o16[d][1] *= a1;
            }
            return;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[mt][r] *= alpha;
    };
    // one softmax unit: two scores of half X -> P (8 units per half).  Beside MFMAs hipcc emits most of these
    // multiply-adds as two scalar v_fma instead of one v_pk_fma_f32 -- rightly: forcing the packed form (inline asm)
    // measured +8 % (4.30 vs 3.97 ms), the packed f32 VALU delays the MFMAs issued around it.
    auto sm_unit = [&](auto x_c, int un, f32x2 c2, f32x2 mc2, float& lsum) {
        constexpr int X = decltype(x_c)::value;
        const int r = un * 2;
        const f32x2 x = f32x2{s[X][r], s[X][r + 1]} * c2 - mc2;
        const float p0 = __builtin_amdgcn_exp2f(x[0]), p1 = __builtin_amdgcn_exp2f(x[1]);
        if constexpr (LSUM_MFMA) {
            // the register pair (4 values of P) completed by the PREVIOUS two units goes onto the lane's running sum: one unit
            // late, so that the conversion that wrote the pair is not the instruction in front of the MFMA that reads it
            // (VALU write -> MFMA read wait states); the last pair of a half is added by lsum_tail
            if (un >= 2 && !(un & 1)) {
                const vec8 v = pf[X][(un - 2) >> 2];
                lacc = T::mfma4(ones4, ((un - 2) & 2) ? v.hi : v.lo, lacc);
            }
            pf[X][r >> 3][r & 7] = (E)p0;
            pf[X][r >> 3][(r & 7) + 1] = (E)p1;
            return;
        }
        if constexpr (!ONES) {
            lsum += p0 + p1;
#ifndef TF_TUNE_IL_LSUM_SINK
            // the running sum must exist HERE: otherwise the 16 adds of a phase sink to its end as one dependent chain
            // behind the last MFMA (nothing of this wave on the matrix pipe meanwhile) instead of riding in the MFMA gaps
            asm volatile("" : "+v"(lsum));
#endif
        }
        pf[X][r >> 3][r & 7] = (E)p0;
        pf[X][r >> 3][(r & 7) + 1] = (E)p1;
    };

    // One phase: the MFMAs of P.V half Hh of the current tile (V^T buffer vbuf) and -- NEXT -- of QK^T half Hh of the
    // next tile (K buffer kbuf), interleaved in program order with the softmax of half 1 - Hh (SM: there is one).
#ifndef TF_TUNE_IL_PF
#define TF_TUNE_IL_PF 2
#endif
    constexpr int PF = TF_TUNE_IL_PF;   // fragment reads run PF steps ahead of their MFMA (register-staged Dh = 40 with 3: 132 VGPRs)
    // LDS fragment i of the MFMA sequence of phase (Hh, NEXT): a P.V fragment of V^T buffer vbuf or a QK^T fragment of K buffer kbuf
    auto frag = [&](auto h_c, auto next_c, int i, int vbuf, int kbuf) -> vec8 {
        constexpr int Hh = decltype(h_c)::value;
        if constexpr (MIX) {
            constexpr IlScheduleMix<NT16, C::KS, decltype(next_c)::value> schm{};
            if (schm.is_pv[i])   // 16 rows x 32 keys of M-tile a: lane row g reads the image columns of k-step g & 1, lane half g >> 1
                return __builtin_bit_cast(vec8, ld16(sV(vbuf) + (schm.a[i] * 16 + (lane & 15)) * VROW + Hh * 32 +
                                                     16 * (((lane >> 4) ^ (MIXSWZ ? (lane + 4) >> 3 : 0)) & 1) + 8 * hi));   // rows 4-11: slot bit 1 flipped
            return __builtin_bit_cast(vec8, ld16(sK(kbuf) + (Hh * 32 + l31) * KROW + 8 * hi + 16 * schm.a[i]));
        }
        constexpr IlSchedule<MT, C::KS, decltype(next_c)::value> sch{};
        if (sch.is_pv[i]) {
            const E* vbase = sV(vbuf) + l31 * VROW + (DMA == 1 ? 0 : Hh * 32 + 8 * hi);
            if constexpr (DMA == 1)   // dense image: 16-B piece index of row (32 a + l31) is XOR-ed with row & 7
                return __builtin_bit_cast(vec8, ld16(vbase + sch.a[i] * 32 * VROW + (((4 * Hh + hi + 2 * sch.b[i]) ^ (l31 & 7)) << 3)));
            return __builtin_bit_cast(vec8, ld16(vbase + sch.a[i] * 32 * VROW + 16 * sch.b[i]));
        }
        return __builtin_bit_cast(vec8, ld16(sK(kbuf) + (Hh * 32 + l31) * KROW + 8 * hi + 16 * sch.a[i]));
    };
    // XPF (cross-phase prefetch): the first PF fragments of a phase that follows another one WITHOUT a barrier between them
    // (the second phase of a tile: same buffers) are read during the last steps of its predecessor and handed over in
    // fr_carry -- a phase otherwise opens with PF reads and a full LDS round trip in front of its first MFMA.
#ifndef TF_TUNE_IL_NO_XPF
    constexpr bool XPF = true;
#else
    constexpr bool XPF = false;
#endif
    vec8 fr_carry[PF];
    auto phase = [&](auto h_c, auto next_c, auto sm_c, auto pre_in_c, auto pre_out_c, int vbuf, int kbuf) {
        constexpr int Hh = decltype(h_c)::value;
        constexpr int X = 1 - Hh;
        constexpr bool NEXT = decltype(next_c)::value, SM = decltype(sm_c)::value;
        constexpr bool PRE_IN = XPF && decltype(pre_in_c)::value;     // fragments 0 .. PF-1 arrive in fr_carry
        constexpr bool PRE_OUT = XPF && decltype(pre_out_c)::value;   // the following phase (half 1 - Hh, same NEXT, same buffers) gets its first PF
        constexpr std::conditional_t<MIX, IlScheduleMix<NT16, C::KS, NEXT>, IlSchedule<MT, C::KS, NEXT>> sch{};
        constexpr int NM = sch.N;
        static_assert(PF <= NM, "prefetch distance beyond one phase");
        bool move = false;
        float alpha = 1.f, lsum = 0.f;
        f32x2 c2 = {c, c}, mc2 = {0.f, 0.f};
        if constexpr (SM) {
            alpha = sm_decide(std::integral_constant<int, X>{}, move);
            const float mc = m_run * c;
            mc2 = f32x2{mc, mc};
            // the matrix-pipe denominator holds sums at the OLD shift only (every earlier half tile, the other half of this tile
            // included) and takes this phase's P -- at the NEW shift -- as the phase goes: rescale it NOW, before the first of
            // them is added.  (O is rescaled at the END of the phase: its P.V of this phase still multiplies P at the old shift.)
            if constexpr (LSUM_MFMA)
                if (move) lacc *= alpha;
        }
        vec8 fr[NM];
#pragma unroll
        for (int i = 0; i < PF; ++i) fr[i] = PRE_IN ? fr_carry[i] : frag(h_c, next_c, i, vbuf, kbuf);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if constexpr (MIX) {
                // a step reads its own fragment only (fidx == i); the first PF steps of every mixed schedule own theirs
                if (i + PF < NM) {
                    if (sch.fidx[i + PF] == i + PF) fr[i + PF] = frag(h_c, next_c, i + PF, vbuf, kbuf);
                } else if constexpr (PRE_OUT) {
                    fr_carry[i + PF - NM] = frag(std::integral_constant<int, X>{}, next_c, i + PF - NM, vbuf, kbuf);
                }
            } else {
                if (i + PF < NM) fr[i + PF] = frag(h_c, next_c, i + PF, vbuf, kbuf);
                else if constexpr (PRE_OUT) fr_carry[i + PF - NM] = frag(std::integral_constant<int, X>{}, next_c, i + PF - NM, vbuf, kbuf);
            }
            if constexpr (MIX) {
                if (sch.is_pv[i])
                    o16[sch.a[i]][sch.b[i]] = T::mfma16(fr[sch.fidx[i]], pf[Hh][sch.b[i]], o16[sch.a[i]][sch.b[i]]);
                else
                    s[Hh] = T::mfma32(fr[i], qf[sch.a[i]], sch.a[i] == 0 ? zero : s[Hh]);
            } else if (sch.is_pv[i]) {
                o[sch.a[i]] = T::mfma32(fr[i], pf[Hh][sch.b[i]], o[sch.a[i]]);
            } else {
                s[Hh] = T::mfma32(fr[i], qf[sch.a[i]], sch.a[i] == 0 ? zero : s[Hh]);
            }
            if constexpr (SM) {
#pragma unroll
                for (int un = (i * 8) / NM; un < ((i + 1) * 8) / NM; ++un)
                    sm_unit(std::integral_constant<int, X>{}, un, c2, mc2, lsum);
            }
            // The non-mixed forms pin every step (1 MFMA : its share of the softmax): without the pins they lose 1-2 % at every head
            // dim (profiles/r06_attn_d40_mix_ab.txt, nosb rows).  The mixed form is faster when hipcc places the softmax itself
            // (it moves the six short P.V MFMAs to the front of the phase, beside the multiply-adds, and the exponentials beside the
            // three long QK^T MFMAs): 3.53 against 3.63 ms pinned, 3.65 the non-mixed kernel.  TF_TUNE_IL40_MIX_PINNED: pinned.
#ifdef TF_TUNE_IL40_MIX_PINNED
            __builtin_amdgcn_sched_barrier(0);
#else
            if constexpr (!MIX) __builtin_amdgcn_sched_barrier(0);
#endif
        }
        if constexpr (SM) {
            // P of half X must exist HERE (keeps the register-only softmax from sinking towards its consumer)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(pf[X][ks]));
            if constexpr (LSUM_MFMA) lacc = T::mfma4(ones4, pf[X][1].hi, lacc);   // the last pair of the half (units 6, 7)
            if constexpr (MIX) {   // (behind the denominator's last pair: it sums the lane's OWN P values)
                p_relayout(std::integral_constant<int, X>{});
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(pf[X][ks]));
            }
            // the shift moved: O (now including this phase's P.V, computed against the old shift) -- and the part of
            // the denominator accumulated so far, all of it at the old shift -- is rescaled before any P at the new
            // shift is multiplied in / added
            if (move) {
                rescale(alpha);
                if constexpr (!ONES && !LSUM_MFMA) l_run *= alpha;
            }
            if constexpr (!ONES && !LSUM_MFMA) l_run += lsum;
        }
    };
    typedef std::integral_constant<int, 0> H0;
    typedef std::integral_constant<int, 1> H1;
    typedef std::true_type Yes;
    typedef std::false_type No;

    // ---- prologue: K(0) -> Kbuf[0]; S(0) = K(0) Q; P0(0); registers <- K(1), V(0)
    if constexpr (DMA != 0) {
        __syncthreads();        // LDS init done before the first DMA lands
        dma_k(0);               // K(0)
        dma_wait();
        __syncthreads();
    } else {
        load_k();
        __syncthreads();            // LDS init done before the first staging write
        write_k(0);
        if (ntiles > 1) load_k();   // K(1)
        load_v();                   // V(0)
        __syncthreads();
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const E* krow = sK(0) + (kt * 32 + l31) * KROW + 8 * hi;
#pragma unroll
        for (int t = 0; t < C::KS; ++t)
            s[kt] = T::mfma32(__builtin_bit_cast(vec8, ld16(krow + 16 * t)), qf[t], t == 0 ? zero : s[kt]);
    }
    {
        bool move;
        (void)sm_decide(H0{}, move);       // first half tile: sets the shift; O and l are zero, nothing to rescale
        const float mc = m_run * c;
        const f32x2 c2 = {c, c}, mc2 = {mc, mc};
        float lsum = 0.f;
#pragma unroll
        for (int un = 0; un < 8; ++un) sm_unit(H0{}, un, c2, mc2, lsum);
        if constexpr (LSUM_MFMA) lacc = T::mfma4(ones4, pf[0][1].hi, lacc);
        if constexpr (MIX) p_relayout(H0{});
        if constexpr (!ONES && !LSUM_MFMA) l_run = lsum;
    }

#ifdef TF_TUNE_IL_PRIO
    // A/B switch: static priority for the second-dispatched half of the workgroup's waves (cdna_hip_programming.md T5)
    if (__builtin_amdgcn_readfirstlane(tid) >= NT / 2) __builtin_amdgcn_s_setprio(1);
#endif
    // All tiles but the last: every phase also runs the QK^T half of the NEXT tile.  The last tile is peeled (no
    // branch on "is there a next tile" inside the loop: the two shapes of the body would otherwise make the
    // compiler keep two copies of the O accumulators and copy between them).
    if constexpr (DMA != 0) {
        // Kbuf[1] and Vbuf[0] hold nothing yet: K(1), V(0) may be issued at once (every wave is past the LDS init)
        if (ntiles > 1) dma_k(1);
        dma_v(0);
    }
    for (int t = 0; t + 1 < ntiles; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        if constexpr (DMA != 0) {
#if defined(TF_TUNE_IL_NOBARRIER_EXPERIMENT)   // timing experiments only (results are WRONG): what the per-tile rendezvous costs ...
#elif defined(TF_TUNE_IL_NOWAIT_EXPERIMENT)      // ... and what the DMA drain in front of it costs (raw barrier, no vmcnt wait)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#else
            dma_wait();                   // this wave's pieces of K(t+1), V(t) have landed ...
            __syncthreads();              // ... everybody's have; every wave has left iteration t-1, whose phases were the
#endif
            __builtin_amdgcn_sched_barrier(0);   // last readers of Kbuf[cur] (K(t)) and Vbuf[nxt] (V(t-1)): free to refill
            if (t + 2 < ntiles) dma_k(cur);      // K(t+2)
            dma_v(nxt);                          // V(t+1)
            phase(H0{}, Yes{}, Yes{}, No{}, Yes{}, cur, nxt);
            phase(H1{}, Yes{}, Yes{}, Yes{}, No{}, cur, nxt);
            continue;
        }
        // Kbuf[nxt] held K(t-1) (last read by QK(t-1) in iteration t-2), Vbuf[cur] held V(t-2) (last read in iteration
        // t-2): every wave has passed the barrier of iteration t-1, which follows iteration t-2 -> free to overwrite.
        write_k(nxt);                 // K(t+1)
        write_v(cur);                 // V(t)
        if (t + 2 < ntiles) load_k(); // K(t+2)
        load_v();                     // V(t+1)
        __syncthreads();              // K(t+1), V(t) visible to all waves
        __builtin_amdgcn_sched_barrier(0);
        phase(H0{}, Yes{}, Yes{}, No{}, Yes{}, cur, nxt);   // O += V0(t) P0(t), S0(t+1)   ||  P1(t)
        phase(H1{}, Yes{}, Yes{}, Yes{}, No{}, cur, nxt);   // O += V1(t) P1(t), S1(t+1)   ||  P0(t+1)
    }
    {
        const int cur = (ntiles - 1) & 1;
        if constexpr (DMA != 0) dma_wait();
        else write_v(cur);            // V(n-1)
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        phase(H0{}, No{}, Yes{}, No{}, Yes{}, cur, cur);    // O += V0 P0   ||  P1
        phase(H1{}, No{}, No{}, Yes{}, No{}, cur, cur);     // O += V1 P1
    }

    // ---- epilogue
    if constexpr (MIX) {
        // O^T[16 d + 4 g + i][16 t + (l & 15)] = o16[d][t][i], g = l >> 4; the ones row (40 = 16 * 2 + 4 * 2 + 0) is register 0 of
        // M-tile 2 in the lanes of row g = 2
        const int g = lane >> 4, n16 = lane & 15;
        float l_t[2];
        if constexpr (ONES) {
#pragma unroll
            for (int t = 0; t < 2; ++t) l_t[t] = __shfl(o16[2][t][0], 32 + n16);
        } else {   // Dh = 64: the matrix-pipe denominator of query l & 31 (both lane halves hold a part)
            const float lq = LSUM_MFMA ? lacc[0] + __shfl_xor(lacc[0], 32) : l_run + __shfl_xor(l_run, 32);
#pragma unroll
            for (int t = 0; t < 2; ++t) l_t[t] = __shfl(lq, 16 * t + n16);
        }
        if (split) {
            constexpr int PS = DH + 8;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int qr = qt * (32 * NW) + wave * 32 + 16 * t + n16;
                if (qr < S) {
                    const int64_t R = (((int64_t)(b - 1) * Kq + f) * H + h) * S + qr;
                    float* row = p.partials + (R * nseg + seg) * PS;
#pragma unroll
                    for (int d = 0; d < NT16; ++d)
                        if (16 * d + 4 * g < DH) *reinterpret_cast<f32x4*>(row + 16 * d + 4 * g) = o16[d][t];
                    if (g == 0) row[DH] = l_t[t];
                }
            }
            if (hi == 0 && q_ok) {   // the shift is this lane's own query's (l & 31)
                const int64_t R = (((int64_t)(b - 1) * Kq + f) * H + h) * S + q_row;
                p.partials[(R * nseg + seg) * PS + DH + 1] = m_run * c;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int qr = qt * (32 * NW) + wave * 32 + 16 * t + n16;
                if (qr < S) {
                    const float inv = 1.0f / l_t[t];
                    const int64_t op = b * p.o_bs + f * p.o_fs + (int64_t)qr * (H * DH) + h * DH;
#pragma unroll
                    for (int d = 0; d < NT16; ++d)
                        if (16 * d + 4 * g < DH) store_out4<E, vec4>(p.out, op + 16 * d + 4 * g, o16[d][t] * inv, p.out_f32);
                }
            }
        }
        return;
    }
    float l_tot;
    if constexpr (ONES)
        l_tot = __shfl(o[MT - 1][ONES_R], l31);   // row VR of the V^T image is 1.0: sum of P from the MFMA
    else if constexpr (LSUM_MFMA)
        l_tot = lacc[0] + __shfl_xor(lacc[0], 32);
    else
        l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv_l = 1.0f / l_tot;
    if (split) {
        // split form: unnormalised O, denominator and shift (log2 domain) of this run of frames for attn_merge_kernel
        if (q_ok) {
            constexpr int PS = DH + 8;
            auto row_ptr = [&](int vb) {
                const int64_t R = (((int64_t)(b - 1 + vb) * Kq + f) * H + h) * S + q_row;
                return p.partials + (R * nseg + seg) * PS;
            };
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int r0 = mt * 32 + 8 * rg + 4 * hi;   // image row of this group of 4 (never straddles a bank)
                    if (r0 < VR) {
                        const int vb = r0 / DH;
                        f32x4 w;
#pragma unroll
                        for (int i = 0; i < 4; ++i) w[i] = o[mt][rg * 4 + i];
                        *reinterpret_cast<f32x4*>(row_ptr(vb) + (r0 - vb * DH)) = w;
                    }
                }
            if (hi == 0) {
#pragma unroll
                for (int vb = 0; vb < (PACK ? 2 : 1); ++vb) {
                    row_ptr(vb)[DH] = l_tot;
                    row_ptr(vb)[DH + 1] = m_run * c;
                }
            }
        }
    } else if (q_ok) {
        const int64_t op = b * p.o_bs + f * p.o_fs + (int64_t)q_row * (H * DH) + h * DH;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int r0 = mt * 32 + 8 * rg + 4 * hi;
                if (r0 < VR) {
                    const int vb = r0 / DH;
                    f32x4 w;
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = o[mt][rg * 4 + i] * inv_l;
                    store_out4<E, vec4>(p.out, op + vb * p.o_bs + (r0 - vb * DH), w, p.out_f32);
                }
            }
    }
}

template <typename T, int DH, int NW, int MODE, int MINW, int DMA = 0>
int launch_il(AttnParams p, hipStream_t st) {
    typedef AttnCfg<DH, 64> C;
    constexpr size_t lds = DMA == 1            ? 2 * (size_t)(64 * DH + C::MT * 32 * 64) * 2 + 16   // dense images (+ the K over-read)
                           : MODE == MODE_DUAL ? 2 * (size_t)(C::K_ELEMS + ((2 * DH + 31) / 32) * 32 * C::VROW) * 2   // packed dual-V image
                                               : C::lds_bytes(1);
    auto kern = ext_attn_il_kernel<T, DH, NW, MODE, MINW, DMA>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    p.nQT = (p.S + 32 * NW - 1) / (32 * NW);
    const int per_branch = p.Kq * p.nQT * p.H;
    const unsigned grid = (unsigned)(MODE == MODE_ALL    ? (2 * p.nseg + (p.part == TF_ATTN_BANK_ONLY ? 0 : 1)) * per_branch
                                     : MODE == MODE_DUAL ? p.nseg * per_branch
                                                         : per_branch);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, st, p);
    TF_LAUNCH_CHECK("tf_ext_attn_fwd");
    return 0;
}

template <typename T, int DH, int MODE, int MINW>
int launch_pp(AttnParams p, hipStream_t st) {
    typedef AttnCfg<DH, 64> C;
    constexpr size_t lds = C::lds_bytes(1);
    auto kern = ext_attn_pp_kernel<T, DH, MODE, MINW>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    p.nQT = (p.S + 255) / 256;
    const int per_branch = p.Kq * p.nQT * p.H;
    // bank problems are decoded first: a bank-only launch simply stops before the source problems
    const unsigned grid = (unsigned)(MODE == MODE_ALL ? (p.part == TF_ATTN_BANK_ONLY ? 2 : 3) * per_branch : per_branch);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, p);
    TF_LAUNCH_CHECK("tf_ext_attn_fwd");
    return 0;
}

template <typename T, int DH, int QT, int NW, int MODE, int MINW, bool FQ = true, int KT = 64, bool SB = false>
int launch_one(AttnParams p, hipStream_t st) {
    typedef AttnCfg<DH, KT> C;
    constexpr size_t lds = ((MODE == MODE_DUAL && DH == 40) ? 2 * (size_t)(C::K_ELEMS + 96 * C::VROW) * 2   // PACK
                                                             : C::lds_bytes(MODE == MODE_DUAL ? 2 : 1)) / (SB ? 2 : 1);
    auto kern = ext_attn_kernel<T, DH, QT, NW, MODE, MINW, KT, FQ, SB>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    p.nQT = (p.S + 32 * QT * NW - 1) / (32 * QT * NW);
    const int per_branch = p.Kq * p.nQT * p.H;
    // bank problems are decoded first: a bank-only launch simply stops before the source problems
    const int ns = MODE == MODE_SOURCE ? 1 : p.nseg;
    const unsigned grid = (unsigned)(MODE == MODE_ALL ? (2 * ns + (p.part == TF_ATTN_BANK_ONLY ? 0 : 1)) * per_branch
                                                      : ns * per_branch);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, st, p);
    TF_LAUNCH_CHECK("tf_ext_attn_fwd");
    return 0;
}

// Geometry per head dim (A/B-measured on MI355X, tools/attn_microbench.py): what matters is the number of
// INDEPENDENT waves per SIMD (softmax VALU of one wave overlaps MFMAs of another) and how many waves share
// one staged tile.  Dh=40: 1 query tile/wave, 8 waves/workgroup, 111 VGPRs -> 4 waves/SIMD.
// Dh=64: 2 query tiles/wave (each LDS fragment feeds 2 MFMAs).  Dh=80/160: register-bound, 1 tile/wave.
template <typename T, int DH>
int launch_attn(const AttnParams& p, const void* v, hipStream_t st) {
    typedef typename T::elem E;
    const bool src_only = p.part == TF_ATTN_SOURCE_ONLY, bank_only = p.part == TF_ATTN_BANK_ONLY;
    {   // pre-pass: V -> transposed, key-permuted, per-frame padded bank (only the branches this call computes)
        const int b_lo = bank_only ? 1 : 0, b_hi = src_only ? 1 : 3;
        dim3 grid((unsigned)(p.Spad / 64), (unsigned)p.H, (unsigned)((b_hi - b_lo) * p.K));
        const size_t lds = (size_t)64 * (DH + 2) * sizeof(E);
        // the Dh = 40 kernels also need the key norm bounds (score bound, see BOUND)
        const bool bound = attn_has_bound(DH);
        hipLaunchKernelGGL(vt_pack_kernel<T>, grid, dim3(256), lds, st, reinterpret_cast<const E*>(v),
                           reinterpret_cast<E*>(const_cast<void*>(p.vt)),
                           bound ? reinterpret_cast<const E*>(p.k) : nullptr, const_cast<float*>(p.knorm2),
                           p.inject, b_lo * p.K, p.K, p.S, p.H, DH, p.Spad, p.ld, p.v_bs, p.v_fs, p.k_bs, p.k_fs);
        TF_LAUNCH_CHECK("tf_ext_attn_fwd(vt_pack)");
    }
    // Every head dim has three forms: ALL (one launch, bank problems then source problems), DUAL (injection:
    // uncond + cond share QK^T and the softmax; pays from S = 256 on) and SOURCE (the source branch alone).
    // A full call is ALL, or DUAL followed by SOURCE; a bank-only call drops the source part, a source-only
    // call is SOURCE alone.
    auto merge = [&]() -> int {   // split form: fold the per-run partial results into the output
        if (p.nseg <= 1) return 0;
        const int64_t total = (int64_t)2 * p.Kq * p.H * p.S * (DH / 4);
        const int64_t blocks = (total + 255) / 256;
        hipLaunchKernelGGL(attn_merge_kernel<T>, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, st,
                           p.partials, p.out, p.Kq, p.S, p.H, DH, p.nseg, p.out_f32, p.o_bs, p.o_fs);
        TF_LAUNCH_CHECK("tf_ext_attn_fwd(merge)");
        return 0;
    };
    auto compose = [&](auto all, auto dual, auto source) -> int {
        if (src_only) return source();
        if (!p.inject || p.S < 256) {   // short frames: the ALL form reads the source q, k itself
            const int rc = all();
            return rc ? rc : merge();
        }
        int rc = dual();
        if (!rc) rc = merge();
        return (rc || bank_only) ? rc : source();
    };
    if constexpr (DH == 40) {
        // 8-wave (256-query) workgroups while they still give 3 workgroups per CU (split runs included: measured
        // -8..11 % on a sharded rank's level 0 against the 4-wave form); below that the 4-wave form (twice the
        // workgroups).  S < 256: always 4 waves.
        const bool big = p.S >= 256 && (int64_t)3 * p.Kq * ((p.S + 255) / 256) * p.H * p.nseg >= 768;
        if (!p.fold) {   // fp32 score scaling: the default
#ifndef TF_TUNE_IL40_MIN_WGS
#define TF_TUNE_IL40_MIN_WGS 256   // one 8-wave workgroup per CU: a W = 8 rank's one-pass level 0 (384 workgroups) runs
#endif                             // 586 us interleaved against 672 us in the plain 4-wave form (profiles/r03_rank_shard.txt)
#ifndef TF_TUNE_NO_IL40
            // half-tile interleaved form (ext_attn_il_kernel)
            const bool il = p.S >= 256 && p.S % 64 == 0 &&
                            (int64_t)3 * p.Kq * ((p.S + 255) / 256) * p.H * p.nseg >= TF_TUNE_IL40_MIN_WGS;
#else
            const bool il = false;
#endif
            // Round 6: K / V^T tiles by LDS-DMA into the padded images (117 instead of 128 VGPRs, no ds_write pass, fragment
            // addresses unchanged): 3.57 against 3.64 ms at cfg2 level 0 (profiles/r06_attn_d40_ab.txt; round 5's dense swizzled
            // DMA images, TF_TUNE_IL40_DMA=1, cost an address computation per fragment read and measured -0.5 %).
            // TF_TUNE_IL40_DMA=0: register staging.
#ifndef TF_TUNE_IL40_DMA
#define TF_TUNE_IL40_DMA 2
#endif
#ifndef TF_TUNE_IL40_NW
#define TF_TUNE_IL40_NW 8
#endif
#ifndef TF_TUNE_IL40_MINW
#define TF_TUNE_IL40_MINW 4
#endif
            // Round 6, last session: mixed MFMA shapes (DMA = 3: QK^T 32x32x16, P.V 16x16x32 over 16-row M-tiles, see IlScheduleMix) --
            // level-0 launch 3.57 -> 3.43 ms, cfg2 step 25.04 -> 24.48 ms on one box (profiles/r06_attn_d40_mix_ab.txt).  Its softmax is
            // placed by hipcc, in coarser alternation with the MFMAs than the pinned steps: launches of ONE round of workgroups, whose
            // waves run in lockstep, lose with it (a rank of 8, level 0: 3.96-4.01 against 3.88-3.89 ms per rank step, section 9 of the
            // same file) -> only launches of >= 2 rounds (1024 eight-wave workgroups), and never in the bit-stable mode, whose kernel
            // choice must be a function of the shape alone (a rank and the single GPU must agree bit for bit there).
#ifndef TF_TUNE_IL40_MIX_MIN_WGS
#define TF_TUNE_IL40_MIX_MIN_WGS 1024
#endif
#ifndef TF_TUNE_NO_IL40_MIX
            const int64_t per_branch = (int64_t)p.Kq * ((p.S + 255) / 256) * p.H;
            const bool mix_all = p.mix || (!p.bit_stable && per_branch * (2 * p.nseg + (bank_only ? 0 : 1)) >= TF_TUNE_IL40_MIX_MIN_WGS);
            const bool mix_src = p.mix || (!p.bit_stable && per_branch >= TF_TUNE_IL40_MIX_MIN_WGS);
#else
            const bool mix_all = false, mix_src = false;
#endif
#if TF_TUNE_IL40_DMA != 0
            if (il)
                return compose([&] { return mix_all ? launch_il<T, 40, TF_TUNE_IL40_NW, MODE_ALL, TF_TUNE_IL40_MINW, 3>(p, st)
                                                    : launch_il<T, 40, TF_TUNE_IL40_NW, MODE_ALL, TF_TUNE_IL40_MINW, TF_TUNE_IL40_DMA>(p, st); },
                               [&] {
#ifndef TF_TUNE_NO_IL40_DUAL
                                   // Round 6: the packed dual-V kernel with LDS-DMA staging, 8-wave workgroups, FOUR waves per SIMD (128
                                   // VGPRs; the register-staged 4-wave form needs 168 = 3 per SIMD): 2.54 against 2.81 ms at cfg2 level 0
                                   // (profiles/r06_attn_d40_dual_ab.txt; DMA alone at 3 waves per SIMD: 2.70)
#ifndef TF_TUNE_IL40_DUAL_DMA
#define TF_TUNE_IL40_DUAL_DMA 2
#endif
#ifndef TF_TUNE_IL40_DUAL_NW
#define TF_TUNE_IL40_DUAL_NW 8
#endif
#ifndef TF_TUNE_IL40_DUAL_MINW
#define TF_TUNE_IL40_DUAL_MINW 4
#endif
                                   if (p.S >= 256 && p.S % 64 == 0)
                                       return launch_il<T, 40, TF_TUNE_IL40_DUAL_NW, MODE_DUAL, TF_TUNE_IL40_DUAL_MINW, TF_TUNE_IL40_DUAL_DMA>(p, st);
#endif
                                   return launch_one<T, DH, 1, 4, MODE_DUAL, 3, false>(p, st);
                               },
                               [&] { return mix_src ? launch_il<T, 40, 8, MODE_SOURCE, 4, 3>(p, st)
                                                    : launch_il<T, 40, 8, MODE_SOURCE, 4, TF_TUNE_IL40_DMA>(p, st); });
#endif
            return compose([&] {
#ifdef TF_TUNE_IL40_NW4_SMALL
                               // A/B switch: fewer than two 8-wave workgroups per CU (a rank's one-pass level 0: 256) as
                               // twice as many 4-wave workgroups -- two barrier groups per CU instead of one
                               if (il && (int64_t)(p.part == TF_ATTN_BANK_ONLY ? 2 : 3) * p.Kq * ((p.S + 255) / 256) * p.H * p.nseg < 512)
                                   return launch_il<T, 40, 4, MODE_ALL, 4>(p, st);
#endif
                               return il    ? launch_il<T, 40, 8, MODE_ALL, 4>(p, st)
                                        : big ? launch_one<T, DH, 1, 8, MODE_ALL, 2, false>(p, st)
                                              : launch_one<T, DH, 1, 4, MODE_ALL, 2, false>(p, st); },
                           [&] {
#ifndef TF_TUNE_NO_IL40_DUAL
                               if (p.S >= 256 && p.S % 64 == 0) return launch_il<T, 40, 4, MODE_DUAL, 3>(p, st);
#endif
                               return launch_one<T, DH, 1, 4, MODE_DUAL, 3, false>(p, st);
                           },
                           [&] {
#ifdef TF_TUNE_IL40_SRC4
                               // A/B switch, off: a source-only call with fewer than 256 8-wave workgroups (a sharded
                               // rank's own frames: 128 at cfg2 level 0, 2 waves per SIMD on half the CUs) as 4-wave
                               // workgroups on every CU (1 wave per SIMD) measured SLOWER, 98 vs 61 us per level-0 block
                               // of a rank (profiles/r05_rank_step_src4_ab.txt): half the waves share each staged tile
                               // and a lone wave per SIMD hides nothing.
                               if (il && (int64_t)p.Kq * ((p.S + 255) / 256) * p.H < 256)
                                   return launch_il<T, 40, 4, MODE_SOURCE, 4>(p, st);
#endif
                               return il    ? launch_il<T, 40, 8, MODE_SOURCE, 4>(p, st)
                                      : big ? launch_one<T, DH, 1, 8, MODE_SOURCE, 2, false>(p, st)
                                            : launch_one<T, DH, 1, 4, MODE_SOURCE, 2, false>(p, st);
                           });
        }
        return compose([&] { return big ? launch_one<T, DH, 1, 8, MODE_ALL, 2>(p, st)
                                        : launch_one<T, DH, 1, 4, MODE_ALL, 2>(p, st); },
                       [&] { return launch_one<T, DH, 1, 4, MODE_DUAL, 3>(p, st); },   // 151 VGPRs: 3 workgroups per CU
                       [&] { return big ? launch_one<T, DH, 1, 8, MODE_SOURCE, 2>(p, st)
                                        : launch_one<T, DH, 1, 4, MODE_SOURCE, 2>(p, st); });
    } else if constexpr (DH == 64) {
        // Round 6: the half-tile interleaved kernel with its K / V^T tiles staged by LDS-DMA into the padded images (no staging
        // registers: 124 VGPRs, FOUR waves per SIMD) and the score bound -- 1067 / 1082 TF/s at cfg4 / cfg5 level 0 against
        // 971 / 982 of the ping-pong kernel on the same box (profiles/r06_attn_d64_ab.txt; the register-staged interleaved form
        // of round 5 needed 136 VGPRs = 3 waves per SIMD and lost to it).  TF_TUNE_NO_IL64: the round-5 dispatch.
#ifndef TF_TUNE_IL64_NW
#define TF_TUNE_IL64_NW 8
#endif
#ifndef TF_TUNE_IL64_MINW
#define TF_TUNE_IL64_MINW 4
#endif
#ifndef TF_TUNE_IL64_DMA
#define TF_TUNE_IL64_DMA 2
#endif
#ifndef TF_TUNE_NO_IL64
        const bool il = p.S % 64 == 0 && p.S >= 512;
#else
        const bool il = false;
#endif
#ifdef TF_TUNE_IL64_MIX
        // A/B switch: the mixed MFMA shapes at Dh = 64 (P.V as 16x16x32 over four 16-row M-tiles: the same matrix-pipe clocks, no
        // padding to remove here; the question is the short shape's power efficiency), launches of >= 1024 workgroups
        const int64_t per_branch64 = (int64_t)p.Kq * ((p.S + 255) / 256) * p.H;
        const bool mix64 = il && (p.mix || (!p.bit_stable && per_branch64 * (2 * p.nseg + (bank_only ? 0 : 1)) >= 1024));
        const bool mix64s = il && (p.mix || (!p.bit_stable && per_branch64 >= 1024));
#else
        const bool mix64 = false, mix64s = false;
#endif
        return compose([&] { return mix64 ? launch_il<T, DH, TF_TUNE_IL64_NW, MODE_ALL, TF_TUNE_IL64_MINW, 3>(p, st)
                                  : il ? launch_il<T, DH, TF_TUNE_IL64_NW, MODE_ALL, TF_TUNE_IL64_MINW, TF_TUNE_IL64_DMA>(p, st)
                                  : (p.S >= 512 && p.nseg == 1) ? launch_pp<T, DH, MODE_ALL, 2>(p, st)   // ragged frames: ping-pong
                                                                : launch_one<T, DH, 1, 4, MODE_ALL, 2>(p, st); },
                       [&] {
#ifndef TF_TUNE_NO_IL64_DUAL
#ifndef TF_TUNE_IL64_DUAL_NW
#define TF_TUNE_IL64_DUAL_NW 4
#endif
                                // Round 6: q/k injection in the interleaved kernel too -- both V banks in one 4-M-tile image (uncond rows
                                // 0-63, cond 64-127), 12 MFMAs per phase against the same 8 softmax units, 162 VGPRs, 2 workgroups of 4 waves
                                // per CU: 18.99 against 20.27 ms at cfg4 level 0, 2.47 / 2.61 at level 1, 0.359 / 0.380 at level 2
                                // (profiles/r06_attn_d64_dual_ab.txt; 8-wave workgroups: 2.35 ms at level 1 but 0.46 at level 2)
                                if (il) return launch_il<T, DH, TF_TUNE_IL64_DUAL_NW, MODE_DUAL, 2, 2>(p, st);
#endif
                                return launch_one<T, DH, 1, 4, MODE_DUAL, 2>(p, st); },
                       [&] { return mix64s ? launch_il<T, DH, TF_TUNE_IL64_NW, MODE_SOURCE, TF_TUNE_IL64_MINW, 3>(p, st)
                                    : il ? launch_il<T, DH, TF_TUNE_IL64_NW, MODE_SOURCE, TF_TUNE_IL64_MINW, TF_TUNE_IL64_DMA>(p, st)
                                       : launch_one<T, DH, 1, 4, MODE_SOURCE, 2>(p, st); });
    } else if constexpr (DH == 80) {
        // Round 6: LDS-DMA staging (padded images) frees the staging registers: 146 VGPRs with 4-wave workgroups, THREE of which
        // fit a CU (3 waves per SIMD, 3 x 50 KB of LDS) -- 0.442 against 0.482 ms at cfg2 level 1 (profiles/r06_attn_d80_ab.txt;
        // register-staged: 8 waves, 166 VGPRs, 2 waves per SIMD; DMA with 8-wave workgroups: no change, 0.479)
#ifndef TF_TUNE_IL80_NW
#define TF_TUNE_IL80_NW 4
#endif
#ifndef TF_TUNE_IL80_MINW
#define TF_TUNE_IL80_MINW 3
#endif
#ifndef TF_TUNE_IL80_DMA
#define TF_TUNE_IL80_DMA 2
#endif
#ifndef TF_TUNE_NO_IL80
        const bool il = p.S % 64 == 0 && p.S >= 256;   // half-tile interleaved form (ext_attn_il_kernel)
#else
        const bool il = false;
#endif
        return compose([&] { return il ? launch_il<T, DH, TF_TUNE_IL80_NW, MODE_ALL, TF_TUNE_IL80_MINW, TF_TUNE_IL80_DMA>(p, st)
                                       : launch_one<T, DH, 1, 4, MODE_ALL, 2>(p, st); },
                       [&] {
#ifndef TF_TUNE_NO_IL80_DUAL
                                // Round 6: q/k injection in the interleaved kernel at d = 80 too -- both V banks in one 5-M-tile image
                                // (uncond rows 0-79, cond 80-159), 184 VGPRs, 2 workgroups of 4 waves per CU: 0.332-0.344 against
                                // 0.374 ms at cfg2 level 1 on one box (profiles/r06_attn_d80_dual_ab.txt)
                                if (il) return launch_il<T, DH, 4, MODE_DUAL, 2, 2>(p, st);
#endif
                                return launch_one<T, DH, 1, 4, MODE_DUAL, 2>(p, st); },
                       [&] { return il ? launch_il<T, DH, TF_TUNE_IL80_NW, MODE_SOURCE, TF_TUNE_IL80_MINW, TF_TUNE_IL80_DMA>(p, st)
                                       : launch_one<T, DH, 1, 4, MODE_SOURCE, 2>(p, st); });
    } else {
        // Dh=160: the dual (shared-softmax) form needs 160 more accumulator registers and measured slower;
        // under injection the ALL form reads the source q and k for every branch instead.  The interleaved kernel
        // was measured here too (8 waves sharing the 89 KB of tiles, 237 VGPRs): 97 vs 92 us at cfg2 level 2 -- at
        // this head dim every MFMA needs its own 1 KB fragment from LDS, whose read rate (128 B/clk per CU) equals
        // the matrix pipes' demand, and the level has one wave per SIMD whatever the kernel.
#ifndef TF_TUNE_DB160
        // single LDS buffer: 44.5 instead of 89 KB per workgroup.  85 vs 95 us at cfg2 level 2 (profiles/r03_attn_sb160.txt):
        // at one wave per SIMD the second buffer bought no overlap, and two workgroups now fit a CU where the grid has them
        if (src_only) return launch_one<T, DH, 1, 4, MODE_SOURCE, 1, true, 64, true>(p, st);
        const int rc = launch_one<T, DH, 1, 4, MODE_ALL, 1, true, 64, true>(p, st);
#else
        if (src_only) return launch_one<T, DH, 1, 4, MODE_SOURCE, 1>(p, st);
        const int rc = launch_one<T, DH, 1, 4, MODE_ALL, 1>(p, st);
#endif
        return rc ? rc : merge();
    }
}

template <typename T>
int dispatch_dh(int Dh, const AttnParams& p, const void* v, hipStream_t st) {
    switch (Dh) {
        case 40: return launch_attn<T, 40>(p, v, st);
        case 64: return launch_attn<T, 64>(p, v, st);
        case 80: return launch_attn<T, 80>(p, v, st);
        case 160: return launch_attn<T, 160>(p, v, st);
    }
    return TF_ERR_SHAPE;
}

}  // namespace

extern "C" size_t tf_ext_attn_workspace_bytes(int K, int S, int H, int Dh, int dtype) {
    if (K <= 0 || S <= 0 || H <= 0 || Dh <= 0 || dtype == TF_F32) return 0;
    const size_t Spad = (size_t)((S + 127) / 128) * 128;   // frames padded to the largest staged tile
    size_t part_elems = 0;   // split form: worst case over the number of query frames a caller may pass
    for (int Kq = 1; Kq <= K; ++Kq)
        for (int inj = 0; inj < 2; ++inj) {
            const int ns = split_plan(K, Kq, S, H, Dh, inj != 0, 0, true);
            const size_t e = ns > 1 ? (size_t)2 * Kq * H * S * ns * (Dh + 8) : 0;
            part_elems = e > part_elems ? e : part_elems;
        }
    return ((vt_bytes(K, (int)Spad, H, Dh) + 255) & ~(size_t)255) +
           (((size_t)3 * H * K * (Spad / 64) * sizeof(float) + 255) & ~(size_t)255) +
           part_elems * sizeof(float);   // V^T image | key norm bounds | split-form partial results
}

extern "C" int tf_ext_attn_fwd_strided(const void* q, const void* k, const void* v, void* out, int K, int Kq,
                                       int q_frame0, int S, int H, int Dh, int64_t ld, const int64_t* strides,
                                       float scale, int inject, int dtype, void* ws, size_t ws_bytes, void* stream) {
    TF_ARG(q && k && v && out && ws && strides, TF_ERR_NULL, "tf_ext_attn_fwd: null pointer");
    TF_ARG(dtype == TF_BF16 || dtype == TF_F16, TF_ERR_DTYPE, "tf_ext_attn_fwd: dtype %d (bf16/f16 only)", dtype);
    TF_ARG(Dh == 40 || Dh == 64 || Dh == 80 || Dh == 160, TF_ERR_SHAPE,
           "tf_ext_attn_fwd: head dim %d not in {40,64,80,160}", Dh);
    TF_ARG(K > 0 && S > 0 && H > 0 && ld >= (int64_t)H * Dh && ld % 8 == 0, TF_ERR_SHAPE,
           "tf_ext_attn_fwd: K=%d S=%d H=%d ld=%lld (ld a multiple of 8, >= H*Dh)", K, S, H, (long long)ld);
    TF_ARG(Kq > 0 && q_frame0 >= 0 && q_frame0 + Kq <= K, TF_ERR_SHAPE,
           "tf_ext_attn_fwd: query frames [%d, %d) outside the %d-frame bank", q_frame0, q_frame0 + Kq, K);
    const int64_t ld_q = strides[8];
    TF_ARG(ld_q >= (int64_t)H * Dh && ld_q % 8 == 0, TF_ERR_SHAPE,
           "tf_ext_attn_fwd: q token stride %lld (a multiple of 8, >= H*Dh)", (long long)ld_q);
    for (int i = 0; i < 8; ++i)
        TF_ARG(strides[i] % 8 == 0 &&
                   (i & 1 ? strides[i] >= (int64_t)(S - 1) * (i < 2 ? ld_q : i < 6 ? ld : (int64_t)H * Dh) : true),
               TF_ERR_SHAPE, "tf_ext_attn_fwd: stride %d = %lld (multiples of 8 elements; a frame holds S token rows)", i,
               (long long)strides[i]);
    TF_ARG(tf_aligned16(q) && tf_aligned16(k) && tf_aligned16(v) && tf_aligned16(out) && tf_aligned16(ws),
           TF_ERR_ALIGN, "tf_ext_attn_fwd: tensors not 16-byte aligned");
    TF_ARG(ws_bytes >= tf_ext_attn_workspace_bytes(K, S, H, Dh, dtype), TF_ERR_WORKSPACE,
           "tf_ext_attn_fwd: workspace %zu < %zu bytes", ws_bytes, tf_ext_attn_workspace_bytes(K, S, H, Dh, dtype));
    const int part_bits = inject & (TF_ATTN_BANK_ONLY | TF_ATTN_SOURCE_ONLY);
    TF_ARG(part_bits != (TF_ATTN_BANK_ONLY | TF_ATTN_SOURCE_ONLY), TF_ERR_SHAPE,
           "tf_ext_attn_fwd: TF_ATTN_BANK_ONLY and TF_ATTN_SOURCE_ONLY exclude each other");
    {   // small problems: one fused launch, no pre-pass, no merge (csrc/ext_attn_fused.hip)
        TfAttnSet a{};
        a.q = q, a.k = k, a.v = v, a.out = out;
        a.q_bs = strides[0], a.q_fs = strides[1], a.ld_q = ld_q;
        a.k_bs = strides[2], a.k_fs = strides[3], a.v_bs = strides[4], a.v_fs = strides[5], a.ld = ld;
        a.o_bs = strides[6], a.o_fs = strides[7];
        a.H = H, a.Kq = Kq, a.q_frame0 = q_frame0, a.Kb = K;
        a.b0 = part_bits == TF_ATTN_BANK_ONLY ? 1 : 0;
        a.nb = part_bits == TF_ATTN_BANK_ONLY ? 2 : part_bits == TF_ATTN_SOURCE_ONLY ? 1 : 3;
        const TfFusedPlan plan = tf_attn_fused_plan(&a, 1, S, Dh, dtype, inject);
        if (plan.use)
            return tf_attn_fused_launch(&a, 1, S, Dh, scale, inject, dtype, plan, reinterpret_cast<hipStream_t>(stream));
    }
    AttnParams p{};
    p.q = q;
    p.k = k;
    p.vt = ws;
    p.knorm2 = reinterpret_cast<const float*>(static_cast<const unsigned char*>(ws) +
                                              ((vt_bytes(K, ((S + 127) / 128) * 128, H, Dh) + 255) & ~(size_t)255));
    p.out = out;
    p.K = K;
    p.Kq = Kq;
    p.q_frame0 = q_frame0;
    p.S = S;
    p.H = H;
    p.Spad = ((S + 127) / 128) * 128;
    p.nQT = (S + 127) / 128;
    p.inject = (inject & TF_ATTN_INJECT) ? 1 : 0;
    p.part = inject & (TF_ATTN_BANK_ONLY | TF_ATTN_SOURCE_ONLY);
    p.fold = (inject & TF_ATTN_FOLD_SCALE) ? 1 : 0;
    p.out_f32 = (inject & TF_ATTN_OUT_F32) ? 1 : 0;
    p.nseg = split_plan(K, Kq, S, H, Dh, p.inject != 0, p.part, !(inject & TF_ATTN_NO_SPLIT));
    p.bit_stable = (inject & TF_ATTN_NO_SPLIT) ? 1 : 0;
    p.mix = (inject & TF_ATTN_HINT_MIX) ? 1 : 0;
    p.partials = reinterpret_cast<float*>(
        reinterpret_cast<unsigned char*>(const_cast<float*>(p.knorm2)) +
        (((size_t)3 * H * K * (((S + 127) / 128) * 128 / 64) * sizeof(float) + 255) & ~(size_t)255));
    p.ld = ld;
    p.ld_q = ld_q;
    p.q_bs = strides[0];
    p.q_fs = strides[1];
    p.k_bs = strides[2];
    p.k_fs = strides[3];
    p.v_bs = strides[4];
    p.v_fs = strides[5];
    p.o_bs = strides[6];
    p.o_fs = strides[7];
    p.c = (float)((double)scale * 1.4426950408889634);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dtype == TF_BF16 ? dispatch_dh<BF16>(Dh, p, v, st) : dispatch_dh<F16>(Dh, p, v, st);
}

extern "C" int tf_ext_attn_fwd(const void* q, const void* k, const void* v, void* out, int K, int Kq, int q_frame0,
                               int S, int H, int Dh, int64_t ld, float scale, int inject, int dtype, void* ws,
                               size_t ws_bytes, void* stream) {
    // dense [3, frames, S, ld] tensors; out [3, Kq, S, H*Dh]
    const int64_t fs = (int64_t)S * ld, ofs = (int64_t)S * H * Dh;
    const int64_t strides[9] = {Kq * fs, fs, K * fs, fs, K * fs, fs, Kq * ofs, ofs, ld};
    return tf_ext_attn_fwd_strided(q, k, v, out, K, Kq, q_frame0, S, H, Dh, ld, strides, scale, inject, dtype, ws,
                                   ws_bytes, stream);
}
