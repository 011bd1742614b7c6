// Extended (cross-keyframe) attention forward for gfx950.
// Replaces tokenflow_utils.py:124-197 / 234-279 of omerbt/TokenFlow: the per-head
// bmm -> *scale -> softmax -> bmm loops over a K-times replicated key/value bank.
//
// Flash-style: one workgroup = 128 queries of one (branch, frame, head); it streams the
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
// key-permuted and zero-padded per frame to 64 keys into caller-provided scratch
// (1 read + 1 write of V, <1% of the attention time at the sizes that matter).
// LDS: double-buffered K [64][DKP+8] and V^T [32*MT][64+8] tiles; the +8 element pad makes
// every row stride an odd number of 16-B slots -> conflict-free ds_read_b128.
// Pipeline: tile i+1 is fetched global->registers before the MFMAs of tile i and written
// to the other LDS buffer after them; one barrier per tile.
// Block order: head = blockIdx % H, so with H = 8 every XCD (block b runs on XCD b % 8)
// serves one head and its L2 holds only that head's bank; bank problems are queued
// before the short source problems so the tail of the grid is filled with short work.
#include <type_traits>

#include "tf_common.h"

namespace {

template <int DH>
struct AttnCfg {
    static constexpr int KS = (DH + 15) / 16;   // QK^T k-steps over the head dim
    static constexpr int DKP = KS * 16;         // head dim padded for QK^T (zero columns)
    static constexpr int KROW = DKP + 8;        // K row stride in LDS (elements)
    static constexpr int MT = (DH + 31) / 32;   // PV M-tiles over the head dim
    static constexpr int VROWS = MT * 32;       // V^T rows in LDS (rows >= DH stay constant)
    static constexpr int VROW = 64 + 8;         // V^T row stride in LDS (elements)
    static constexpr int PPR = DH / 8;          // 16-B pieces per K row
    static constexpr int K_ELEMS = 64 * KROW;
    static constexpr int V_ELEMS = VROWS * VROW;
    static constexpr int npk(int nt) { return (64 * PPR + nt - 1) / nt; }   // K pieces per thread
    static constexpr int npv(int nt) { return (DH * 8 + nt - 1) / nt; }     // V^T pieces per thread
    static constexpr size_t lds_bytes(int nb) { return 2 * (size_t)(K_ELEMS + nb * V_ELEMS) * 2; }
};

enum { MODE_ALL = 0, MODE_SOURCE = 1, MODE_DUAL = 2 };

struct AttnParams {
    const void* q;
    const void* k;
    const void* vt;
    void* out;
    int K, Kq, q_frame0, S, H, Spad, nQT, inject;  // K bank frames; queries = frames q_frame0 .. +Kq
    int64_t ld;
    float c;  // scale * log2(e)
};

__device__ __forceinline__ int swap23(int x) { return (x & ~12) | ((x & 4) << 1) | ((x & 8) >> 1); }

// V [3,K,S,H*DH] (token stride ld) -> Vt [3][H][DH][K*Spad], position = f*Spad + swap23(key in frame),
// zero for keys >= S.  grid = (Spad/64, H, 3*K), 256 threads.
template <typename T>
__global__ __launch_bounds__(256) void vt_pack_kernel(const typename T::elem* __restrict__ v,
                                                      typename T::elem* __restrict__ vt, int K, int S, int H, int DH,
                                                      int Spad, int64_t ld) {
    typedef typename T::elem E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    E* tile = reinterpret_cast<E*>(smem);  // [64][DH + 2]
    const int row = DH + 2;
    const int tt = blockIdx.x, h = blockIdx.y, bf = blockIdx.z;  // bf = b*K + f
    const int b = bf / K, f = bf - b * K;
    const E* src = v + ((int64_t)bf * S) * ld + h * DH;
    for (int id = threadIdx.x; id < 64 * DH; id += 256) {
        const int key = id / DH, d = id - key * DH;
        const int kk = tt * 64 + key;
        tile[key * row + d] = kk < S ? src[(int64_t)kk * ld + d] : (E)0.f;
    }
    __syncthreads();
    E* dst = vt + ((int64_t)(b * H + h) * DH) * ((int64_t)K * Spad) + (int64_t)f * Spad + tt * 64;
    for (int id = threadIdx.x; id < 64 * DH; id += 256) {
        const int d = id >> 6, pos = id & 63;
        dst[(int64_t)d * ((int64_t)K * Spad) + pos] = tile[swap23(pos) * row + d];
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
template <typename T, int DH, int QT, int NW, int MODE, int MINW>
__global__ __launch_bounds__(64 * NW, MINW) void ext_attn_kernel(AttnParams p) {
    typedef AttnCfg<DH> C;
    typedef typename T::elem E;
    typedef typename T::vec8 vec8;
    typedef typename T::vec4 vec4;
    constexpr int NT = 64 * NW;
    constexpr int NB = MODE == MODE_DUAL ? 2 : 1;   // V banks handled by this workgroup
    constexpr int NPK = C::npk(NT), NPV = C::npv(NT);
    constexpr int BUF_ELEMS = C::K_ELEMS + NB * C::V_ELEMS;
    // When the head dim is not a multiple of 32 the last PV M-tile has unused rows: row DH of the
    // V^T image is set to 1.0, so that accumulator row collects sum_k P[k] -- the softmax
    // denominator comes out of the MFMA for free, summed over the SAME rounded P as the numerator.
    constexpr bool ONES = (DH % 32) != 0;
    constexpr int ONES_R = ((DH % 32) & 3) + 4 * ((DH % 32) >> 3);  // C/D register of row DH%32 (lane half 0)
    static_assert(!ONES || ((DH % 32) & 4) == 0, "row DH must live in lane half 0");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    auto sK = [&](int buf) { return reinterpret_cast<E*>(smem) + buf * BUF_ELEMS; };
    auto sV = [&](int buf, int vb) { return reinterpret_cast<E*>(smem) + buf * BUF_ELEMS + C::K_ELEMS + vb * C::V_ELEMS; };

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
    if constexpr (MODE == MODE_ALL) {   // bank problems (uncond, cond) first, then the short source ones
        const int nbank = 2 * Kq * p.nQT;
        if (u < nbank) {
            b = 1 + u / (Kq * p.nQT);
            u -= (b - 1) * Kq * p.nQT;
        } else {
            u -= nbank;
            b = 0;
        }
    } else {
        b = MODE == MODE_DUAL ? 1 : 0;
    }
    f = u / p.nQT;
    qt = u - f * p.nQT;
    const int bq = (p.inject && b > 0) ? 0 : b;  // branch whose q and k are used (tokenflow_utils.py:124-130)
    const int f_lo = b == 0 ? p.q_frame0 + f : 0;
    const int n_fr = b == 0 ? 1 : K;
    const int tpf = (S + 63) >> 6;  // 64-key tiles per frame
    const int ntiles = n_fr * tpf;
    const bool ragged = (S & 63) != 0;

    const E* qg = reinterpret_cast<const E*>(p.q);
    const E* kg = reinterpret_cast<const E*>(p.k) + ((int64_t)bq * K * S) * p.ld + h * DH;
    const int64_t vt_row = (int64_t)K * p.Spad;
    const E* vg[NB];
#pragma unroll
    for (int vb = 0; vb < NB; ++vb)
        vg[vb] = reinterpret_cast<const E*>(p.vt) + ((int64_t)((b + vb) * H + h) * DH) * vt_row;

    // ---- LDS pads, written once and never staged over: K columns DH..DKP-1 = 0,
    //      V^T rows DH..VROWS-1 = 0 except row DH = 1 (denominator row) when ONES.
    if constexpr (C::DKP > DH) {
        for (int id = tid; id < 2 * 64 * (C::DKP - DH); id += NT) {
            const int bufi = id / (64 * (C::DKP - DH));
            const int r = (id / (C::DKP - DH)) % 64, cidx = id % (C::DKP - DH);
            sK(bufi)[r * C::KROW + DH + cidx] = (E)0.f;
        }
    }
    if constexpr (C::VROWS > DH) {
        for (int id = tid; id < 2 * NB * (C::VROWS - DH) * 64; id += NT) {
            const int bv = id / ((C::VROWS - DH) * 64);
            const int r = (id >> 6) % (C::VROWS - DH), cidx = id & 63;
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
        const E* qp = qg + (((int64_t)bq * Kq + f) * S + (q_ok[qi] ? q_row[qi] : S - 1)) * p.ld + h * DH;
#pragma unroll
        for (int t = 0; t < C::KS; ++t) {
            const int col = 16 * t + 8 * hi;
            qf[qi][t] = __builtin_bit_cast(vec8, col < DH ? ld16(qp + col) : u32x4{0, 0, 0, 0});
        }
    }

    // ---- staging: per-thread piece offsets are loop-invariant; a tile only moves uniform base pointers
    u32x4 rk[NPK], rv[NB][NPV];
    int k_goff[NPK], k_loff[NPK], v_goff[NPV], v_loff[NPV];
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
        const int id = tid + NT * i;
        const int r = id / C::PPR, pc = id - r * C::PPR;
        k_goff[i] = r * (int)p.ld + pc * 8;
        k_loff[i] = r * C::KROW + pc * 8;
    }
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
        const int id = tid + NT * i;
        v_goff[i] = (id >> 3) * (int)vt_row + (id & 7) * 8;
        v_loff[i] = (id >> 3) * C::VROW + (id & 7) * 8;
    }
    auto stage_load = [&](int tile) {
        const int fi = tile / tpf;
        const int tt = tile - fi * tpf;
        const int fk = f_lo + fi;
        const E* kt = kg + ((int64_t)fk * S + tt * 64) * p.ld;
        if (ragged && tt == tpf - 1) {  // keys past S: clamp the row (masked later), keep the load in bounds
#pragma unroll
            for (int i = 0; i < NPK; ++i) {
                const int id = tid + NT * i;
                if (id < 64 * C::PPR) {
                    const int r = id / C::PPR, pc = id - r * C::PPR;
                    const int key = tt * 64 + r < S ? r : S - 1 - tt * 64;
                    rk[i] = ld16(kt + (int64_t)key * p.ld + pc * 8);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NPK; ++i)
                if (tid + NT * i < 64 * C::PPR) rk[i] = ld16(kt + k_goff[i]);
        }
#pragma unroll
        for (int vb = 0; vb < NB; ++vb) {
            const E* vt = vg[vb] + (int64_t)fk * p.Spad + tt * 64;
#pragma unroll
            for (int i = 0; i < NPV; ++i)
                if (tid + NT * i < DH * 8) rv[vb][i] = ld16(vt + v_goff[i]);
        }
    };
    auto stage_write = [&](int buf) {
        E* kb = sK(buf);
#pragma unroll
        for (int i = 0; i < NPK; ++i)
            if (tid + NT * i < 64 * C::PPR) st16(kb + k_loff[i], rk[i]);
#pragma unroll
        for (int vb = 0; vb < NB; ++vb) {
            E* vbp = sV(buf, vb);
#pragma unroll
            for (int i = 0; i < NPV; ++i)
                if (tid + NT * i < DH * 8) st16(vbp + v_loff[i], rv[vb][i]);
        }
    };

    f32x16 o[NB][QT][C::MT];
    float m_run[QT], l_run[QT];  // running max of the RAW scores (scale > 0); this lane's share of the denominator
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
        m_run[qi] = -INFINITY;
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

    stage_load(0);
    __syncthreads();  // pad fill visible before anything reads; staging regions are disjoint from the pads
    stage_write(0);
    __syncthreads();

    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        const bool has_next = tile + 1 < ntiles;
        if (has_next) stage_load(tile + 1);

        // Program order per tile: QK(q0) QK(q1) | softmax(q0) PV(q0) | softmax(q1) PV(q1).
        // MFMAs execute asynchronously behind the in-order issue, so the softmax VALU of one query
        // tile runs while the matrix pipe works on the other one's QK^T / P.V.
        f32x16 s[QT][2];  // S^T tiles: 64 keys x 32 queries each
#pragma unroll
        for (int qi = 0; qi < QT; ++qi)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[qi][kt][r] = 0.f;
                const E* krow = sK(buf) + (kt * 32 + l31) * C::KROW + 8 * hi;
#pragma unroll
                for (int t = 0; t < C::KS; ++t)
                    s[qi][kt] = T::mfma32(__builtin_bit_cast(vec8, ld16(krow + 16 * t)), qf[qi][t], s[qi][kt]);
            }
        if (ragged) {
            const int tt = tile - (tile / tpf) * tpf;
            if (tt == tpf - 1) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (tt * 64 + kt * 32 + cd_row(r, hi) >= S) {
#pragma unroll
                            for (int qi = 0; qi < QT; ++qi) s[qi][kt][r] = -INFINITY;
                        }
            }
        }

#pragma unroll
        for (int qi = 0; qi < QT; ++qi) {
            // ---- online softmax (lane-local; the two lanes of a query share m)
            float mx = s[qi][0][0];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qi][kt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            // rescale only when some query of this wave saw a new maximum: alpha == 1 exactly otherwise
            if (__any(mx > m_run[qi])) {
                const float m_new = fmaxf(m_run[qi], mx);
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
            vec8 pf[4];
            float lsum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 x = f32x2{s[qi][kt][r], s[qi][kt][r + 1]} * c2 - mc2;  // v_pk_fma_f32
                    const float p0 = __builtin_amdgcn_exp2f(x[0]);
                    const float p1 = __builtin_amdgcn_exp2f(x[1]);
                    if constexpr (!ONES) lsum += p0 + p1;
                    pf[kt * 2 + (r >> 3)][r & 7] = (E)p0;
                    pf[kt * 2 + (r >> 3)][(r & 7) + 1] = (E)p1;
                }
            if constexpr (!ONES) l_run[qi] += lsum;
            // ---- O^T += V^T . P  (once per V bank)
#pragma unroll
            for (int vb = 0; vb < NB; ++vb)
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt) {
                    const E* vrow = sV(buf, vb) + (mt * 32 + l31) * C::VROW + 8 * hi;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        o[vb][qi][mt] =
                            T::mfma32(__builtin_bit_cast(vec8, ld16(vrow + 16 * ks)), pf[ks], o[vb][qi][mt]);
                }
        }

        if (has_next) stage_write(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: normalise, round, store 4 consecutive d (8 B) per register group
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
        float l_tot;
        if constexpr (ONES)
            l_tot = __shfl(o[0][qi][C::MT - 1][ONES_R], l31);  // row DH lives in lane half 0 of the last M-tile
        else
            l_tot = l_run[qi] + __shfl_xor(l_run[qi], 32);
        const float inv_l = 1.0f / l_tot;
        if (q_ok[qi]) {
#pragma unroll
            for (int vb = 0; vb < NB; ++vb) {
                E* op = reinterpret_cast<E*>(p.out) +
                        (((int64_t)(b + vb) * Kq + f) * S + q_row[qi]) * ((int64_t)H * DH) + h * DH;
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int d0 = mt * 32 + 8 * rg + 4 * hi;
                        if (d0 < DH) {
                            vec4 w;
#pragma unroll
                            for (int i = 0; i < 4; ++i) w[i] = (E)(o[vb][qi][mt][rg * 4 + i] * inv_l);
                            *reinterpret_cast<u32x2*>(op + d0) = __builtin_bit_cast(u32x2, w);
                        }
                    }
            }
        }
    }
}

template <typename T, int DH, int QT, int NW, int MODE, int MINW>
int launch_one(AttnParams p, hipStream_t st) {
    typedef AttnCfg<DH> C;
    constexpr size_t lds = C::lds_bytes(MODE == MODE_DUAL ? 2 : 1);
    auto kern = ext_attn_kernel<T, DH, QT, NW, MODE, MINW>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    p.nQT = (p.S + 32 * QT * NW - 1) / (32 * QT * NW);
    const int per_branch = p.Kq * p.nQT * p.H;
    const unsigned grid = (unsigned)(MODE == MODE_ALL ? 3 * per_branch : per_branch);
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
    {   // pre-pass: V -> transposed, key-permuted, per-frame padded bank
        dim3 grid((unsigned)(p.Spad / 64), (unsigned)p.H, (unsigned)(3 * p.K));
        const size_t lds = (size_t)64 * (DH + 2) * sizeof(E);
        hipLaunchKernelGGL(vt_pack_kernel<T>, grid, dim3(256), lds, st, reinterpret_cast<const E*>(v),
                           reinterpret_cast<E*>(const_cast<void*>(p.vt)), p.K, p.S, p.H, DH, p.Spad, p.ld);
        TF_LAUNCH_CHECK("tf_ext_attn_fwd(vt_pack)");
    }
    if constexpr (DH == 40) {
        if (p.S >= 256) {
            if (p.inject) {   // 151 VGPRs: 4-wave workgroups, 3 per CU
                const int rc = launch_one<T, DH, 1, 4, MODE_DUAL, 3>(p, st);
                return rc ? rc : launch_one<T, DH, 1, 8, MODE_SOURCE, 2>(p, st);
            }
            return launch_one<T, DH, 1, 8, MODE_ALL, 2>(p, st);
        }
        return launch_one<T, DH, 1, 4, MODE_ALL, 2>(p, st);
    } else if constexpr (DH == 64) {
        if (p.inject && p.S >= 256) {
            const int rc = launch_one<T, DH, 1, 4, MODE_DUAL, 2>(p, st);
            return rc ? rc : launch_one<T, DH, 1, 4, MODE_SOURCE, 2>(p, st);
        }
        if (p.S >= 512) return launch_one<T, DH, 2, 4, MODE_ALL, 2>(p, st);
        return launch_one<T, DH, 1, 4, MODE_ALL, 2>(p, st);
    } else if constexpr (DH == 80) {
        if (p.inject && p.S >= 256) {
            const int rc = launch_one<T, DH, 1, 4, MODE_DUAL, 2>(p, st);
            return rc ? rc : launch_one<T, DH, 1, 4, MODE_SOURCE, 2>(p, st);
        }
        return launch_one<T, DH, 1, 4, MODE_ALL, 2>(p, st);
    } else {
        // Dh=160: the dual (shared-softmax) form needs 160 more accumulator registers and measured slower
        return launch_one<T, DH, 1, 4, MODE_ALL, 1>(p, st);
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
    const size_t Spad = (size_t)((S + 63) / 64) * 64;
    return (size_t)3 * H * Dh * K * Spad * 2;
}

extern "C" int tf_ext_attn_fwd(const void* q, const void* k, const void* v, void* out, int K, int Kq, int q_frame0,
                               int S, int H, int Dh, int64_t ld, float scale, int inject, int dtype, void* ws,
                               size_t ws_bytes, void* stream) {
    TF_ARG(q && k && v && out && ws, TF_ERR_NULL, "tf_ext_attn_fwd: null pointer");
    TF_ARG(dtype == TF_BF16 || dtype == TF_F16, TF_ERR_DTYPE, "tf_ext_attn_fwd: dtype %d (bf16/f16 only)", dtype);
    TF_ARG(Dh == 40 || Dh == 64 || Dh == 80 || Dh == 160, TF_ERR_SHAPE,
           "tf_ext_attn_fwd: head dim %d not in {40,64,80,160}", Dh);
    TF_ARG(K > 0 && S > 0 && H > 0 && S % 8 == 0 && ld >= (int64_t)H * Dh && ld % 8 == 0, TF_ERR_SHAPE,
           "tf_ext_attn_fwd: K=%d S=%d H=%d ld=%lld (S, ld multiples of 8; ld >= H*Dh)", K, S, H, (long long)ld);
    TF_ARG(Kq > 0 && q_frame0 >= 0 && q_frame0 + Kq <= K, TF_ERR_SHAPE,
           "tf_ext_attn_fwd: query frames [%d, %d) outside the %d-frame bank", q_frame0, q_frame0 + Kq, K);
    TF_ARG(tf_aligned16(q) && tf_aligned16(k) && tf_aligned16(v) && tf_aligned16(out) && tf_aligned16(ws),
           TF_ERR_ALIGN, "tf_ext_attn_fwd: tensors not 16-byte aligned");
    TF_ARG(ws_bytes >= tf_ext_attn_workspace_bytes(K, S, H, Dh, dtype), TF_ERR_WORKSPACE,
           "tf_ext_attn_fwd: workspace %zu < %zu bytes", ws_bytes, tf_ext_attn_workspace_bytes(K, S, H, Dh, dtype));
    AttnParams p;
    p.q = q;
    p.k = k;
    p.vt = ws;
    p.out = out;
    p.K = K;
    p.Kq = Kq;
    p.q_frame0 = q_frame0;
    p.S = S;
    p.H = H;
    p.Spad = ((S + 63) / 64) * 64;
    p.nQT = (S + 127) / 128;
    p.inject = inject ? 1 : 0;
    p.ld = ld;
    p.c = (float)((double)scale * 1.4426950408889634);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dtype == TF_BF16 ? dispatch_dh<BF16>(Dh, p, v, st) : dispatch_dh<F16>(Dh, p, v, st);
}
