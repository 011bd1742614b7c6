// Fused small-problem form of the extended attention (tokenflow_utils.py:124-197 / 234-279 of omerbt/TokenFlow) for
// gfx950: the coarse UNet levels (S <= 256), BASELINE config 1, and a frame-sharded rank's share of the middle levels.
//
// The streaming kernels of ext_attn.hip spend a V^T pre-pass, (on small grids) a split + merge launch pair and -- on a
// sharded rank -- a separate source-branch launch around the attention itself.  At these sizes each of those launches
// costs as much as the attention (4-30 us each).  This kernel needs none of them:
//   * ONE launch over every (tensor set, branch, frame, head, 32*QW-query tile) problem; a sharded rank passes two
//     tensor sets -- the bank branches on the buffer its all-to-all delivered and the source branch of its own frames;
//   * V is staged ROW-major, exactly as it lies in HBM, and the P.V MFMA's A operand (V^T: rows = features, k = keys)
//     is read with ds_read_b64_tr_b16: a 16-lane group reads a [4 keys][16 features] block and every lane receives
//     the 4 keys of ITS feature.  The 4-key runs it delivers (keys 4hi..4hi+3 and 8+4hi..) are exactly the key order
//     of the S^T accumulator registers, so P still goes from the QK^T accumulator into the P.V MFMA without a shuffle;
//   * the key sequence of a problem is split over the KW wave groups of the workgroup itself (sub-tile j of 32 keys
//     goes to group j % KW) and the KW partial results are merged through LDS in the epilogue: the parallelism of the
//     split form without partials in HBM, without a merge launch and without any cross-workgroup protocol;
//   * PREC: P is carried as hi + lo bf16 (two P.V MFMAs on the same V^T fragment), which removes the 2^-9 rounding of
//     P -- the dominant error term where few keys are averaged (the reference rounds P to fp16, 2^-11, at the same
//     point: tokenflow_utils.py:177-179 under run_tokenflow_pnp.py:220).  f16 inputs carry P as f16 (11 bits).
// MFMA mapping as in ext_attn.hip: S^T = K Q^T (a lane owns one query and 16 of the sub-tile's 32 keys), O^T = V^T P.
// Online softmax per 32-key sub-tile against a per-query reference point that follows the running maximum with a lag
// of TF_FUSED_LAG binades (P <= 2^8: in range for f16 as well); O is rescaled only when a reference moves (softmax_p).
// Arithmetic of a (query, head) depends on KW and PREC only -- never on QW or on the grid -- so a rank reproduces the
// single-GPU result bit for bit whenever both take this kernel with the same KW (tf_attn_fused_plan: shape-only rules
// in TF_ATTN_NO_SPLIT mode).
#include <cstdlib>
#include <type_traits>

#include "attn_fused.h"

namespace {

typedef __bf16 bf16x4_vs __attribute__((__vector_size__(8)));
typedef __fp16 fp16x4_vs __attribute__((__vector_size__(8)));
#define TF_LDS_AS __attribute__((address_space(3)))

// ds_read_b64_tr_b16: within a 16-lane group, lane i passes the address of block[i >> 2][4 * (i & 3) .. +3] of a
// [4][16] block of 16-bit elements and receives block[0..3][i]  (cdna_hip_programming.md T10)
template <typename T>
struct TrRead;
template <>
struct TrRead<BF16> {
    static __device__ __forceinline__ u32x2 rd(const __bf16* p) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((TF_LDS_AS bf16x4_vs*)(p)));
    }
};
template <>
struct TrRead<F16> {
    static __device__ __forceinline__ u32x2 rd(const _Float16* p) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((TF_LDS_AS fp16x4_vs*)(p)));
    }
};

template <int DH>
struct FusedCfg {
    static constexpr int KS = (DH + 15) / 16;    // QK^T k-steps over the head dim
    static constexpr int DKP = KS * 16;
    static constexpr int KROW = DKP + 8;         // K row stride in LDS (elements): an odd number of 16-B slots
    static constexpr int MT = (DH + 31) / 32;    // P.V M-tiles over the head dim
    // V row stride (elements).  The transpose read serves 32 lanes per LDS cycle: 4 rows x 64 B; they fall on disjoint
    // bank ranges when the row stride is 64 or 192 (mod 256) bytes.
    static constexpr int VS = DH == 160 ? 160 : 96;
    static constexpr int PPR = DH / 8;           // 16-B pieces per K / V row
    static constexpr int K_ELEMS = 32 * KROW;
    static_assert(VS >= MT * 32, "a transpose read must stay inside its V row");
    static_assert((VS * 2) % 256 == 64 || (VS * 2) % 256 == 192, "bank-conflict-free transpose reads");
};

struct FusedSet {   // device form of TfAttnSet
    const void* q;
    const void* k;
    const void* v;
    void* out;
    int64_t q_bs, q_fs, ld_q, k_bs, k_fs, v_bs, v_fs, ld, o_bs, o_fs;
    int H, Kq, q_frame0, Kb, b0, nb, n_wg, pad_;
};

struct FusedParams {
    FusedSet set[2];
    int n_sets, S, nQT, tpf;   // nQT = query tiles per frame, tpf = 32-key sub-tiles per frame
    unsigned tpf_magic;        // j / tpf = umulhi(j, tpf_magic) for tpf > 1
    int inject, out_f32;
    float c;                   // scale * log2(e)
    float lag;                 // TF_FUSED_LAG / c: raw score units the sub-tile maximum may exceed a query's reference by
};

__device__ __forceinline__ float max_xor32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

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

// Pieces shared by the two kernel forms below -------------------------------------------------------------------------

struct Problem {   // one (set, branch, frame, head, query tile) problem, decoded from blockIdx.x
    int si, h, qt, f, b, bq, f_lo, n_fr;
};
__device__ __forceinline__ Problem decode_problem(const FusedParams& p) {
    // sets in order (the long bank problems of set 0 first), head fastest (H = 8: one head per XCD)
    Problem pr;
    int u = blockIdx.x;
    pr.si = 0;
    if (p.n_sets > 1 && u >= p.set[0].n_wg) {
        u -= p.set[0].n_wg;
        pr.si = 1;
    }
    const FusedSet& st = p.set[pr.si];
    pr.h = u % st.H;
    u /= st.H;
    pr.qt = u % p.nQT;
    u /= p.nQT;
    pr.f = u % st.Kq;
    const int bi = u / st.Kq;
    // a full set (source + two bank branches) runs its bank branches first
    pr.b = (st.b0 == 0 && st.nb == 3) ? (bi == 2 ? 0 : bi + 1) : st.b0 + bi;
    pr.bq = (p.inject && pr.b > 0) ? 0 : pr.b;   // branch whose q and k are used (tokenflow_utils.py:124-130)
    pr.f_lo = pr.b == 0 ? st.q_frame0 + pr.f : 0;
    pr.n_fr = pr.b == 0 ? 1 : st.Kb;
    return pr;
}

// Wave-uniform cursor over the sub-tiles j0, j0 + step, ... of a problem's key sequence (frames of tpf sub-tiles);
// past the end it stays on the last sub-tile, so that the branch-free staging loads always have a valid address.
struct Cursor {
    int fr, tt, j;
    __device__ __forceinline__ void init(int j0, int tpf, int nst) {
        j = j0 < nst ? j0 : nst - 1;
        fr = 0, tt = j;
        while (tt >= tpf) tt -= tpf, ++fr;
    }
    __device__ __forceinline__ void advance(int step, int tpf, int nst) {
        if (j + step < nst) {
            j += step, tt += step;
            while (tt >= tpf) tt -= tpf, ++fr;
        }
    }
};

// online softmax of one 32-key sub-tile (lane-local; the two lanes of a query share the maximum): moves the query's
// reference point m_run -- and rescales O -- only when the sub-tile's maximum exceeds it by more than `lag` raw score units
// (= TF_FUSED_LAG binades of P), leaves P of the two 16-key k-steps (registers 0-7 / 8-15 of the accumulator) in ph (and
// the rounding remainders in pl with PREC).
// The reference point need not be the running maximum: any m gives the same softmax as long as numerator and denominator
// use it, and P <= 2^TF_FUSED_LAG is harmless in bf16 / f16 / fp32 (P keeps its relative precision).  With the exact
// running maximum a wave of 32 queries rescaled on 54 of its 64 sub-tiles (some query almost always sees a new maximum:
// expected count 32 + 32 ln 2) -- 24 packed multiplies, an exponential and a dependent chain each time; with the lag the
// reference moves once or twice per query.  The decision is PER QUERY (a query whose maximum did not pass its own
// reference keeps alpha = 1 exactly), so the arithmetic of a query never depends on its neighbours in the wave.
#ifndef TF_FUSED_LAG
#define TF_FUSED_LAG 8
#endif
template <typename T, int DH, bool PREC>
__device__ __forceinline__ void softmax_p(f32x16& s, f32x16 (&o)[FusedCfg<DH>::MT], float& m_run, float& l_run, float c,
                                          float lag, typename T::vec8 (&ph)[2], typename T::vec8 (&pl)[2]) {
    typedef FusedCfg<DH> C;
    typedef typename T::elem E;
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = max_xor32(mx);
    const bool move = mx > m_run + lag;   // -inf + lag = -inf: the first sub-tile always sets the reference
    if (__any(move)) {
        const float m_new = move ? mx : m_run;
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);   // exp2(-inf) = 0 on the first sub-tile; 1 if !move
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[mt][r] *= alpha;
    }
    const float mc = m_run * c;
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], c, -mc));
        lsum += pr;
        const E e = (E)pr;
        ph[r >> 3][r & 7] = e;
        if constexpr (PREC) pl[r >> 3][r & 7] = (E)(pr - (float)e);
    }
    l_run += lsum;
}

// O^T += V^T P for the QB query blocks of a wave: every V^T fragment is transpose-read ONCE from the row-major V image
// `vtr` (this lane's address for k-step 0, M-tile 0) and multiplied into each block's accumulator
template <typename T, int DH, int QB, bool PREC>
__device__ __forceinline__ void pv_acc(f32x16 (&o)[QB][FusedCfg<DH>::MT], typename T::vec8 (&ph)[QB][2],
                                       typename T::vec8 (&pl)[QB][2], const typename T::elem* vtr) {
    typedef FusedCfg<DH> C;
    typedef typename T::vec8 vec8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
            const u32x2 a0 = TrRead<T>::rd(vtr + (16 * ks) * C::VS + mt * 32);
            const u32x2 a1 = TrRead<T>::rd(vtr + (16 * ks + 8) * C::VS + mt * 32);
            const vec8 a = __builtin_bit_cast(vec8, u32x4{a0[0], a0[1], a1[0], a1[1]});
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                o[qb][mt] = T::mfma32(a, ph[qb][ks], o[qb][mt]);
                if constexpr (PREC) o[qb][mt] = T::mfma32(a, pl[qb][ks], o[qb][mt]);
            }
        }
}

// Epilogue: merge the KW key groups of every query wave through LDS (the staging area is free: the caller has passed a
// workgroup barrier behind the last read of it) and store.  Group order of the sums is fixed (k = 0 .. KW-1), so the
// result does not depend on timing; an idle group has m = -inf, l = 0 and weight 0.
template <typename T, int DH, int QW, int KW>
__device__ __forceinline__ void merge_store(unsigned char* smem, f32x16 (&o)[FusedCfg<DH>::MT], float m_run, float l_run,
                                            float c, int wave, int lane, void* out, int64_t out_row, bool q_ok, int out_f32) {
    typedef FusedCfg<DH> C;
    typedef typename T::elem E;
    typedef typename T::vec4 vec4;
    constexpr int NW = QW * KW;
    const int hi = lane >> 5, l31 = lane & 31;
    const int qw = wave / KW, kw = wave % KW;
    const float l_wave = l_run + __shfl_xor(l_run, 32);   // both lanes of a query: the wave's denominator
    auto store_tile = [&](int mt, const f32x16& acc, float inv_l) {
        if (!q_ok) return;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int d0 = mt * 32 + 8 * rg + 4 * hi;
            if (d0 < DH) {
                f32x4 w;
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = acc[rg * 4 + i] * inv_l;
                store_out4<E, vec4>(out, out_row + d0, w, out_f32);
            }
        }
    };
    if constexpr (KW == 1) {
        const float inv_l = 1.0f / l_wave;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) store_tile(mt, o[mt], inv_l);
    } else {
        float* stat = reinterpret_cast<float*>(smem + 2 * NW * 16 * 64 * 4);   // [NW][64]: lanes 0-31 m, lanes 32-63 l
        float* tile = reinterpret_cast<float*>(smem);                         // [2][NW][16][64]
        stat[wave * 64 + lane] = hi == 0 ? m_run : l_wave;
        __syncthreads();
        float M = -INFINITY;
#pragma unroll
        for (int k2 = 0; k2 < KW; ++k2) M = fmaxf(M, stat[(qw * KW + k2) * 64 + l31]);
        float L = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < KW; ++k2) {
            const float mk = stat[(qw * KW + k2) * 64 + l31], lk = stat[(qw * KW + k2) * 64 + 32 + l31];
            L = fmaf(lk, __builtin_amdgcn_exp2f((mk - M) * c), L);
        }
        const float w_own = __builtin_amdgcn_exp2f((m_run - M) * c);
        const float inv_l = 1.0f / L;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
            float* buf = tile + (mt & 1) * (NW * 16 * 64);
#pragma unroll
            for (int r = 0; r < 16; ++r) buf[(wave * 16 + r) * 64 + lane] = o[mt][r] * w_own;
            __syncthreads();   // round mt written; the sums of round mt-1 (other buffer) are complete by program order
            if (mt % KW == kw) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float a = buf[((qw * KW) * 16 + r) * 64 + lane];
#pragma unroll
                    for (int k2 = 1; k2 < KW; ++k2) a += buf[((qw * KW + k2) * 16 + r) * 64 + lane];
                    acc[r] = a;
                }
                store_tile(mt, acc, inv_l);
            }
        }
    }
}

template <int NW, int KW>
constexpr int merge_bytes() { return KW > 1 ? 2 * NW * 16 * 64 * 4 + NW * 64 * 4 : 0; }

// ---------------------------------------------------------------------------------------------------------------------
// Shared-tile form (QW >= 2 query waves share every staged sub-tile; large grids of small frames).
// QW   query waves per workgroup: the workgroup covers 32*QW queries of one (branch, frame, head)
// KW   key groups: sub-tile j (32 keys) of the problem's key sequence is computed by the waves of group j % KW; the QW
//      waves of group kw stage slot kw together (wave-uniform base pointer + per-thread constant offsets)
// PREC P as hi + lo (bf16 only)
template <typename T, int DH, int QW, int KW, bool PREC>
__global__ __launch_bounds__(64 * QW * KW, 1) void ext_attn_fused_kernel(FusedParams p) {
    typedef FusedCfg<DH> C;
    typedef typename T::elem E;
    typedef typename T::vec8 vec8;
    constexpr int NW = QW * KW, NT = 64 * NW;
    constexpr int SLOT_PIECES = 32 * C::PPR;                      // 16-B pieces of K (and of V) per sub-tile
    constexpr int NP = (SLOT_PIECES + 64 * QW - 1) / (64 * QW);   // pieces of K (and of V) per thread and iteration
    constexpr int SLOT_ELEMS = C::K_ELEMS + 32 * C::VS;
    constexpr int STAGE_BYTES = KW * SLOT_ELEMS * 2;
    static_assert((STAGE_BYTES > merge_bytes<NW, KW>() ? STAGE_BYTES : merge_bytes<NW, KW>()) <= 160 * 1024, "LDS");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    E* lds = reinterpret_cast<E*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qw = wave / KW, kw = wave % KW;
    const int hi = lane >> 5, l31 = lane & 31;
    const int S = p.S, tpf = p.tpf;
    E* sK = lds + kw * SLOT_ELEMS;
    E* sV = sK + C::K_ELEMS;

    const Problem pr = decode_problem(p);
    const FusedSet& st = p.set[pr.si];
    const int nst = pr.n_fr * tpf;                // sub-tiles of this problem
    const int nit = (nst + KW - 1) / KW;
    const E* kg = reinterpret_cast<const E*>(st.k) + pr.bq * st.k_bs + pr.f_lo * st.k_fs + pr.h * DH;
    const E* vg = reinterpret_cast<const E*>(st.v) + pr.b * st.v_bs + pr.f_lo * st.v_fs + pr.h * DH;
    const int64_t k_fs = st.k_fs, v_fs = st.v_fs;
    const int ld = (int)st.ld;

    // ---- LDS: zero once (the K pad columns DH..DKP-1 must be zero; everything else is staged before it is read)
    for (int id = tid; id < STAGE_BYTES / 16; id += NT) st16(smem + id * 16, u32x4{0, 0, 0, 0});

    // ---- Q fragments (B operand of S^T = K Q^T), resident for the whole kernel
    const int q_row = pr.qt * (32 * QW) + qw * 32 + l31;
    const bool q_ok = q_row < S;
    vec8 qf[C::KS];
    {
        const E* qp = reinterpret_cast<const E*>(st.q) + pr.bq * st.q_bs + pr.f * st.q_fs +
                      (int64_t)(q_ok ? q_row : S - 1) * st.ld_q + pr.h * DH;
#pragma unroll
        for (int t = 0; t < C::KS; ++t) {
            const int col = 16 * t + 8 * hi;
            qf[t] = __builtin_bit_cast(vec8, col < DH ? ld16(qp + col) : u32x4{0, 0, 0, 0});
        }
    }

    // ---- staging of this wave's slot: piece -> (key row, column) is fixed per thread; an iteration moves a
    //      wave-uniform base.  Branch-free loads: a key row past S re-loads the last valid row (such keys are masked),
    //      a slot past the end of the sequence re-loads the last sub-tile (never computed).
    u32x4 rk[NP], rv[NP];
    int s_row[NP], s_col[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int id = min(qw * 64 + lane + 64 * QW * i, SLOT_PIECES - 1);
        s_row[i] = id / C::PPR;
        s_col[i] = (id - s_row[i] * C::PPR) * 8;
    }
    Cursor ldc, cc;   // sub-tile being loaded / computed
    ldc.init(kw, tpf, nst);
    cc.init(kw, tpf, nst);
    auto stage_load = [&]() {
        const int nvalid = min(32, S - ldc.tt * 32);
        const E* kb = kg + ldc.fr * k_fs + (int64_t)(ldc.tt * 32) * ld;
        const E* vb = vg + ldc.fr * v_fs + (int64_t)(ldc.tt * 32) * ld;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int off = min(s_row[i], nvalid - 1) * ld + s_col[i];
            rk[i] = ld16(kb + off);
            rv[i] = ld16(vb + off);
        }
        ldc.advance(KW, tpf, nst);
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (qw * 64 + lane + 64 * QW * i < SLOT_PIECES) {
                st16(sK + s_row[i] * C::KROW + s_col[i], rk[i]);
                st16(sV + s_row[i] * C::VS + s_col[i], rv[i]);
            }
    };

    f32x16 o[1][C::MT];
    float m_run = -INFINITY;   // running maximum of the raw scores of this wave's sub-tiles
    float l_run = 0.f;         // this lane's share of the denominator
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[0][mt][r] = 0.f;
    const float c = p.c, lag = p.lag;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // fragment addresses inside this wave's slot
    const E* kfrag = sK + l31 * C::KROW + 8 * hi;
    // transpose read: 16-lane group g = lane >> 4 covers features 16*(g & 1) .. +15 of the M-tile and -- lane half
    // hi = g >> 1 -- keys 4hi .. 4hi+3 (+8 for the second read) of the 16-key k-step; lane i passes the address of
    // block[i >> 2][4 * (i & 3)]
    const int li = lane & 15, lg = lane >> 4;
    const E* vtr = sV + (4 * hi + (li >> 2)) * C::VS + 16 * (lg & 1) + 4 * (li & 3);

    stage_load();
    __syncthreads();   // zero fill complete before the first staging write

    for (int it = 0; it < nit; ++it) {
        stage_write();
        __syncthreads();   // sub-tiles of this iteration visible
        stage_load();      // next iteration's loads fly under the MFMAs (past the end: a harmless re-load)
        if (it * KW + kw < nst) {
            const int key0 = cc.tt * 32;
            f32x16 s = zero;   // S^T = K Q^T
#pragma unroll
            for (int t = 0; t < C::KS; ++t) s = T::mfma32(__builtin_bit_cast(vec8, ld16(kfrag + 16 * t)), qf[t], s);
            if (key0 + 32 > S) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (key0 + cd_row(r, hi) >= S) s[r] = -INFINITY;
            }
            vec8 ph[1][2], pl[1][2];
            softmax_p<T, DH, PREC>(s, o[0], m_run, l_run, c, lag, ph[0], pl[0]);
            pv_acc<T, DH, 1, PREC>(o, ph, pl, vtr);
            cc.advance(KW, tpf, nst);
        }
        __syncthreads();   // every wave is done with this iteration's sub-tiles
    }
    const int64_t out_row = pr.b * st.o_bs + pr.f * st.o_fs + (int64_t)q_row * (st.H * DH) + pr.h * DH;
    merge_store<T, DH, QW, KW>(smem, o[0], m_run, l_run, c, wave, lane, st.out, out_row, q_ok, p.out_f32);
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-private form (small grids: a sharded rank's share of a level, BASELINE config 1): ONE tile of 32*QB queries per
// workgroup, its key sequence split over KW = 4 or 8 waves, and NOTHING shared between the waves until the merge -- so
// the main loop has no workgroup barrier at all: every wave streams its own sub-tiles at its own pace.
//   * K: the MFMA A fragments (a key row per lane, 16 B per k-step) are loaded straight from global memory into
//     registers, one sub-tile ahead (these problems are L2-resident; LDS would buy coalescing only);
//   * V: loaded one sub-tile ahead (lane = half a key row), written row-major into the wave's private LDS region and
//     transpose-read as V^T fragments.  LDS operations of one wave execute in order, so write -> read -> next write
//     need no barrier;
//   * Q: registers; at Dh = 160 (where the accumulators alone are 80 registers) in LDS, shared by the KW waves;
//   * QB = 2 (head dims <= 80, longer key sequences): a wave owns two 32-query blocks -- every K and V^T fragment it
//     fetches feeds two MFMAs, which halves the K / V re-reads per query.  Opt-in (TF_ATTN_HINT_QB2): on a rank's
//     level 1 it measured 70-72 against 72-75 us -- the launch is bound by its instruction mix, not by the re-reads
//     (profiles/r04_pmc_fused.csv) -- at 250+ registers (one wave per SIMD at Dh = 80).
// Same arithmetic per (query, head) as the shared-tile form with the same KW: bit-identical results.
template <typename T, int DH, int KW, int QB, bool PREC>
__global__ __launch_bounds__(64 * KW, 1) void ext_attn_fused_wp_kernel(FusedParams p) {
    typedef FusedCfg<DH> C;
    typedef typename T::elem E;
    typedef typename T::vec8 vec8;
    constexpr int NT = 64 * KW;
    constexpr bool LQ = DH == 160;                    // Q fragments from LDS
    static_assert(!(LQ && QB > 1), "two query blocks per wave: head dims <= 80");
    constexpr int NPH = (C::PPR + 1) / 2;             // 16-B pieces of V per lane and sub-tile (half a row)
    constexpr int V_ELEMS = 32 * C::VS;
    constexpr int STAGE_BYTES = (KW * V_ELEMS + (LQ ? C::K_ELEMS : 0)) * 2;
    static_assert((STAGE_BYTES > merge_bytes<KW, KW>() ? STAGE_BYTES : merge_bytes<KW, KW>()) <= 160 * 1024, "LDS");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    E* lds = reinterpret_cast<E*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int kw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int S = p.S, tpf = p.tpf;
    E* sV = lds + kw * V_ELEMS;
    E* sQ = lds + KW * V_ELEMS;

    const Problem pr = decode_problem(p);
    const FusedSet& st = p.set[pr.si];
    const int nst = pr.n_fr * tpf;
    const int nmine = kw < nst ? (nst - kw + KW - 1) / KW : 0;   // sub-tiles of this wave: kw, kw + KW, ...
    const E* kg = reinterpret_cast<const E*>(st.k) + pr.bq * st.k_bs + pr.f_lo * st.k_fs + pr.h * DH;
    const E* vg = reinterpret_cast<const E*>(st.v) + pr.b * st.v_bs + pr.f_lo * st.v_fs + pr.h * DH;
    const int64_t k_fs = st.k_fs, v_fs = st.v_fs;
    const int ld = (int)st.ld;

    // ---- Q fragments
    int q_row[QB];
    bool q_ok[QB];
    vec8 qf[QB][LQ ? 1 : C::KS];
    const E* qbase = reinterpret_cast<const E*>(st.q) + pr.bq * st.q_bs + pr.f * st.q_fs + pr.h * DH;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        q_row[qb] = (pr.qt * QB + qb) * 32 + l31;
        q_ok[qb] = q_row[qb] < S;
    }
    if constexpr (LQ) {
        // the 32 x DH query tile, row-major with the K image's row stride: every wave copies 32 / KW rows
        for (int id = tid; id < 32 * C::PPR; id += NT) {
            const int row = id / C::PPR, pc = id - row * C::PPR;
            const int qr = min(pr.qt * 32 + row, S - 1);
            st16(sQ + row * C::KROW + pc * 8, ld16(qbase + (int64_t)qr * st.ld_q + pc * 8));
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const E* qp = qbase + (int64_t)(q_ok[qb] ? q_row[qb] : S - 1) * st.ld_q;
#pragma unroll
            for (int t = 0; t < C::KS; ++t) {
                const int col = 16 * t + 8 * hi;
                qf[qb][t] = __builtin_bit_cast(vec8, col < DH ? ld16(qp + col) : u32x4{0, 0, 0, 0});
            }
        }
    }
    const E* qfrag = sQ + l31 * C::KROW + 8 * hi;

    // ---- loads of a sub-tile's K and V, each with its own cursor; branch-free (clamped rows / cursor), see the
    //      shared-tile form.  K runs TWO sub-tiles ahead of the softmax (its S^T = K Q^T is issued one sub-tile early,
    //      below), V one.
    u32x4 rk[C::KS], rv[NPH];
    const int v_row = lane >> 1, v_pc0 = (lane & 1) * NPH;
    Cursor ldk, ldv;
    ldk.init(kw, tpf, nst);
    ldv.init(kw, tpf, nst);
    int key0_k = 0;   // first key (inside its frame) of the sub-tile whose K fragments sit in rk
    auto load_k = [&]() {
        const int nvalid = min(32, S - ldk.tt * 32);
        key0_k = ldk.tt * 32;
        const E* kb = kg + ldk.fr * k_fs + (int64_t)(ldk.tt * 32) * ld + min(l31, nvalid - 1) * ld + 8 * hi;
#pragma unroll
        for (int t = 0; t < C::KS; ++t) rk[t] = (16 * t + 8 * hi < DH) ? ld16(kb + 16 * t) : u32x4{0, 0, 0, 0};
        ldk.advance(KW, tpf, nst);
    };
    auto load_v = [&]() {
        const int nvalid = min(32, S - ldv.tt * 32);
        const E* vb = vg + ldv.fr * v_fs + (int64_t)(ldv.tt * 32) * ld + min(v_row, nvalid - 1) * ld;
#pragma unroll
        for (int i = 0; i < NPH; ++i) rv[i] = ld16(vb + min(v_pc0 + i, C::PPR - 1) * 8);
        ldv.advance(KW, tpf, nst);
    };

    f32x16 o[QB][C::MT];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -INFINITY, l_run[qb] = 0.f;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][mt][r] = 0.f;
    }
    const float c = p.c, lag = p.lag;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, lg = lane >> 4;
    const E* vtr = sV + (4 * hi + (li >> 2)) * C::VS + 16 * (lg & 1) + 4 * (li & 3);

    // S^T = K Q^T of the sub-tile in rk (k-step outermost: consecutive MFMAs hit different accumulators)
    f32x16 s_next[QB];
    int key0_next = 0;
    auto qk = [&]() {
        key0_next = key0_k;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s_next[qb] = zero;
#pragma unroll
        for (int t = 0; t < C::KS; ++t)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const vec8 qv = LQ ? __builtin_bit_cast(vec8, ld16(qfrag + 16 * t)) : qf[qb][LQ ? 0 : t];
                s_next[qb] = T::mfma32(__builtin_bit_cast(vec8, rk[t]), qv, s_next[qb]);
            }
    };

    // Software pipeline over this wave's sub-tiles: the QK^T MFMAs of sub-tile i+1 are issued BEFORE the softmax of
    // sub-tile i, so the matrix pipe works through them while the wave's VALU does the exponentials -- without it a
    // barrier-free wave runs its MFMAs and its softmax strictly one after the other (352 + 230 clk per sub-tile at
    // Dh = 80).  Same arithmetic, same order per (query, key): bit-identical results.
    load_k();
    load_v();
    qk();        // sub-tile 0
    load_k();    // sub-tile 1 (past the end: a harmless re-load)
    for (int i = 0; i < nmine; ++i) {
        // V(i) -> this wave's LDS region (the transpose reads of sub-tile i-1 precede these writes in the wave's
        // in-order LDS stream)
#pragma unroll
        for (int j2 = 0; j2 < NPH; ++j2)
            if (v_pc0 + j2 < C::PPR) st16(sV + v_row * C::VS + (v_pc0 + j2) * 8, rv[j2]);
        __builtin_amdgcn_wave_barrier();
        load_v();    // V(i+1) flies under everything below
        f32x16 s[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s[qb] = s_next[qb];
        const int key0 = key0_next;
        if (i + 1 < nmine) {
            qk();        // sub-tile i+1: executes under the softmax of sub-tile i
            load_k();    // sub-tile i+2
        }
        vec8 ph[QB][2], pl[QB][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (key0 + 32 > S) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (key0 + cd_row(r, hi) >= S) s[qb][r] = -INFINITY;
            }
            softmax_p<T, DH, PREC>(s[qb], o[qb], m_run[qb], l_run[qb], c, lag, ph[qb], pl[qb]);
        }
        pv_acc<T, DH, QB, PREC>(o, ph, pl, vtr);
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        __syncthreads();   // every wave has left its loop (or the previous block's merge): the LDS becomes the merge area
        const int64_t out_row = pr.b * st.o_bs + pr.f * st.o_fs + (int64_t)q_row[qb] * (st.H * DH) + pr.h * DH;
        merge_store<T, DH, 1, KW>(smem, o[qb], m_run[qb], l_run[qb], c, kw, lane, st.out, out_row, q_ok[qb], p.out_f32);
    }
}

template <typename T, int DH, int QW, int KW, bool PREC>
int launch_fused(const FusedParams& p, unsigned grid, hipStream_t st) {
    typedef FusedCfg<DH> C;
    constexpr int NW = QW * KW;
    constexpr int STAGE_BYTES = KW * (C::K_ELEMS + 32 * C::VS) * 2;
    constexpr int lds = STAGE_BYTES > merge_bytes<NW, KW>() ? STAGE_BYTES : merge_bytes<NW, KW>();
    auto kern = ext_attn_fused_kernel<T, DH, QW, KW, PREC>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, st, p);
    TF_LAUNCH_CHECK("tf_ext_attn_fwd(fused)");
    return 0;
}

template <typename T, int DH, int KW, int QB, bool PREC>
int launch_fused_wp(const FusedParams& p, unsigned grid, hipStream_t st) {
    typedef FusedCfg<DH> C;
    constexpr int STAGE_BYTES = (KW * 32 * C::VS + (DH == 160 ? C::K_ELEMS : 0)) * 2;
    constexpr int lds = STAGE_BYTES > merge_bytes<KW, KW>() ? STAGE_BYTES : merge_bytes<KW, KW>();
    auto kern = ext_attn_fused_wp_kernel<T, DH, KW, QB, PREC>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * KW), lds, st, p);
    TF_LAUNCH_CHECK("tf_ext_attn_fwd(fused)");
    return 0;
}

template <typename T, int DH, bool PREC>
int dispatch_geom(const FusedParams& p, unsigned grid, int qw, int kw, int qb, hipStream_t st) {
    if constexpr (DH <= 80) {
        if (qw == 1 && kw == 4 && qb == 2) return launch_fused_wp<T, DH, 4, 2, PREC>(p, grid, st);
    }
    if (qw == 1 && kw == 4 && qb == 1) return launch_fused_wp<T, DH, 4, 1, PREC>(p, grid, st);
    if (qw == 1 && kw == 8 && qb == 1) return launch_fused_wp<T, DH, 8, 1, PREC>(p, grid, st);
    if (qb != 1) {
        tf_set_error("tf_ext_attn_fwd(fused): two query blocks per wave exist for the wave-private form with 4 key groups, "
                     "head dims <= 80");
        return TF_ERR_SHAPE;
    }
    if (qw == 2 && kw == 4) return launch_fused<T, DH, 2, 4, PREC>(p, grid, st);
    if (qw == 4 && kw == 2) return launch_fused<T, DH, 4, 2, PREC>(p, grid, st);
    if (qw == 4 && kw == 1) return launch_fused<T, DH, 4, 1, PREC>(p, grid, st);
    tf_set_error("tf_ext_attn_fwd(fused): geometry (%d query waves, %d key groups) is not built", qw, kw);
    return TF_ERR_SHAPE;
}

template <typename T, int DH>
int dispatch_prec(const FusedParams& p, unsigned grid, const TfFusedPlan& plan, hipStream_t st) {
    constexpr bool bf = std::is_same<T, BF16>::value;
    if (bf && plan.prec) return dispatch_geom<T, DH, bf>(p, grid, plan.qw, plan.kw, plan.qb, st);
    return dispatch_geom<T, DH, false>(p, grid, plan.qw, plan.kw, plan.qb, st);
}

template <typename T>
int dispatch_dh(int Dh, const FusedParams& p, unsigned grid, const TfFusedPlan& plan, hipStream_t st) {
    switch (Dh) {
        case 40: return dispatch_prec<T, 40>(p, grid, plan, st);
        case 64: return dispatch_prec<T, 64>(p, grid, plan, st);
        case 80: return dispatch_prec<T, 80>(p, grid, plan, st);
        case 160: return dispatch_prec<T, 160>(p, grid, plan, st);
    }
    return TF_ERR_SHAPE;
}

int64_t n_problems(const TfAttnSet* sets, int n_sets) {
    int64_t n = 0;
    for (int i = 0; i < n_sets; ++i) n += (int64_t)sets[i].nb * sets[i].Kq * sets[i].H;
    return n;
}

}  // namespace

// Which calls take the fused kernel, and in which geometry (measured: profiles/r04_fused_microbench.txt).
//   * KW (the in-workgroup key split) and PREC change the arithmetic of a (query, head); QW, the staging form
//     (wave-private / shared tiles) and the grid do not.
//   * TF_ATTN_NO_SPLIT (bit-stable mode): S <= 256 -> KW = 4, whatever the grid, the bank size or the branches of the
//     call -- a function of (S, Dh, dtype) only, so a sharded rank's calls and the single-GPU call agree and one-pass
//     results stay bit-identical across grid sizes.  Larger frames keep the streaming kernels' one-pass form.
//   * otherwise (free to choose per grid): S <= 256 -> small grids (<= 1024 32-query tiles: a sharded rank, BASELINE
//     config 1, the 8x8 level) take the wave-private form with KW = 4; larger grids the shared-tile form without a
//     key split (4 query waves share every sub-tile), unless the bank is long (> 4096 keys: the hi + lo P.V costs
//     more there than the pre-pass it saves, and the rounding of P averages out).  256 < S <= 1024 -> only small
//     grids (a sharded rank's level 1): wave-private, KW = 4.
//   * PREC: bf16 and S <= 256 (shape only, and independent of which branches a call computes: the parts of a
//     sharded rank's pass must round exactly as the single-GPU call does).
#ifndef TF_TUNE_FUSED_SMALL_GRID
#define TF_TUNE_FUSED_SMALL_GRID 1024
#endif
TfFusedPlan tf_attn_fused_plan(const TfAttnSet* sets, int n_sets, int S, int Dh, int dtype, int flags) {
    TfFusedPlan pl{};
    if (flags & TF_ATTN_NO_FUSED) return pl;
    if (!(Dh == 40 || Dh == 64 || Dh == 80 || Dh == 160) || (dtype != TF_BF16 && dtype != TF_F16)) return pl;
    if (flags & TF_ATTN_FOLD_SCALE) return pl;   // the folded-scale opt-in is a Dh = 40 streaming-kernel form
    const int64_t n_qt = n_problems(sets, n_sets) * ((S + 31) / 32);   // 32-query tiles of the launch
    int Kb = 1;
    for (int i = 0; i < n_sets; ++i)
        if (sets[i].b0 + sets[i].nb > 1 && sets[i].Kb > Kb) Kb = sets[i].Kb;
    const bool small_grid = n_qt <= TF_TUNE_FUSED_SMALL_GRID;
    if (flags & TF_ATTN_NO_SPLIT) {
        pl.use = S <= 256;
        pl.kw = 4;
        pl.qw = small_grid ? 1 : 2;
    } else if (S <= 256) {
        pl.use = small_grid || (int64_t)Kb * S <= 4096;
        pl.kw = small_grid ? 4 : 1;
        pl.qw = small_grid ? 1 : 4;
    } else {
        // TOKENFLOW_FUSED_MAX_S (experiments): frames above it keep the streaming kernels on small grids too
        static const int max_s = [] { const char* e = getenv("TOKENFLOW_FUSED_MAX_S"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
        pl.use = S <= max_s && small_grid;
        pl.kw = 4;
        pl.qw = 1;
    }
    pl.qb = 1;
    if (flags & TF_ATTN_FUSED) pl.use = 1;
    if (!pl.use) return pl;
    pl.prec = dtype == TF_BF16 && S <= 256;
    // hints (development / A-B measurements; 0 = automatic)
    const int hq = (flags >> 8) & 7, hk = (flags >> 11) & 7;
    if (hq) pl.qw = 1 << (hq - 1);
    if (hk) pl.kw = 1 << (hk - 1);   // codes 1..4 = 1, 2, 4, 8 key groups
    if (flags & TF_ATTN_HINT_QB2) pl.qb = 2;
    if (flags & TF_ATTN_PRECISE_P) pl.prec = dtype == TF_BF16;
    if (flags & TF_ATTN_NO_PRECISE_P) pl.prec = 0;
    return pl;
}

int tf_attn_fused_launch(const TfAttnSet* sets, int n_sets, int S, int Dh, float scale, int flags, int dtype,
                         const TfFusedPlan& plan, hipStream_t st) {
    TF_ARG(sets && n_sets >= 1 && n_sets <= 2, TF_ERR_SHAPE, "tf_ext_attn_fwd(fused): %d tensor sets", n_sets);
    FusedParams p{};
    p.n_sets = n_sets;
    p.S = S;
    p.nQT = (S + 32 * plan.qw * plan.qb - 1) / (32 * plan.qw * plan.qb);
    p.tpf = (S + 31) / 32;
    p.tpf_magic = p.tpf > 1 ? (unsigned)(((uint64_t)1 << 32) / (unsigned)p.tpf + 1) : 0u;
    p.inject = (flags & TF_ATTN_INJECT) ? 1 : 0;
    p.out_f32 = (flags & TF_ATTN_OUT_F32) ? 1 : 0;
    p.c = (float)((double)scale * 1.4426950408889634);
    p.lag = (float)TF_FUSED_LAG / p.c;
    int64_t grid = 0;
    for (int i = 0; i < n_sets; ++i) {
        const TfAttnSet& a = sets[i];
        FusedSet& d = p.set[i];
        TF_ARG(a.q && a.k && a.v && a.out, TF_ERR_NULL, "tf_ext_attn_fwd(fused): null pointer in set %d", i);
        TF_ARG(a.H > 0 && a.Kq > 0 && a.Kb > 0 && a.nb >= 1 && a.b0 >= 0 && a.b0 + a.nb <= 3 && a.q_frame0 >= 0,
               TF_ERR_SHAPE, "tf_ext_attn_fwd(fused): set %d: H=%d Kq=%d Kb=%d branches [%d, %d)", i, a.H, a.Kq, a.Kb, a.b0,
               a.b0 + a.nb);
        d.q = a.q, d.k = a.k, d.v = a.v, d.out = a.out;
        d.q_bs = a.q_bs, d.q_fs = a.q_fs, d.ld_q = a.ld_q, d.k_bs = a.k_bs, d.k_fs = a.k_fs, d.v_bs = a.v_bs, d.v_fs = a.v_fs;
        d.ld = a.ld, d.o_bs = a.o_bs, d.o_fs = a.o_fs;
        d.H = a.H, d.Kq = a.Kq, d.q_frame0 = a.q_frame0, d.Kb = a.Kb, d.b0 = a.b0, d.nb = a.nb;
        d.n_wg = a.nb * a.Kq * a.H * p.nQT;
        grid += d.n_wg;
    }
    TF_ARG(grid > 0 && grid < ((int64_t)1 << 31), TF_ERR_SHAPE, "tf_ext_attn_fwd(fused): grid of %lld workgroups",
           (long long)grid);
    // the magic division is exact for j * tpf < 2^32: the sub-tile index stays far below that
    return dtype == TF_BF16 ? dispatch_dh<BF16>(Dh, p, (unsigned)grid, plan, st)
                            : dispatch_dh<F16>(Dh, p, (unsigned)grid, plan, st);
}
