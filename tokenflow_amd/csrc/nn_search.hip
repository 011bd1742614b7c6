// Nearest-neighbour token search for gfx950:
//   idx[p, t] = argmax_j <tgt[t], piv[kf[p], j]> * inv_norm[kf[p], j]
// replaces util.py:61-69 (batch_cosine_sim) + tokenflow_utils.py:335-343 (chunk + argmax)
// of omerbt/TokenFlow.  The [n*S, P*S] similarity matrix never leaves registers.
//
// Structure: a GEMM with an argmax epilogue.  M = pivots (MFMA A operand, rows),
// N = targets (B operand, columns), contraction over D.  With the 32x32 C/D map a
// lane owns ONE target column and 16 pivot rows per tile, so the running
// (max, argmax) over pivots is lane-local; lanes l and l+32 are merged once at the
// end, then the two pivot-half waves through LDS.  A workgroup owns TN targets and
// sweeps ALL S pivots of its keyframe, so there is no cross-workgroup merge.
//   workgroup = 256 threads = 4 waves as 2 (pivot halves) x 2 (target halves)
//   tile      = 128 pivots x TN targets x 64 (D chunk), TN = 64*WN... see below
//   LDS       = double-buffered A/B chunk images, 128-B rows, 16-B slots XOR-swizzled
//               with ((row >> 1) & 7) -> conflict-free ds_read_b128 fragment reads
//   pipeline  = global->register prefetch of chunk i+1 issued before the MFMAs of
//               chunk i, LDS write after them, one barrier per chunk.
// Two kernels share that epilogue:
//   nn_search_rb_kernel<T, DK>  (D = 16*DK <= 320): the wave's 32 target rows live in REGISTERS as
//               MFMA B fragments for the whole kernel (80 VGPRs at D = 320), so only the pivot
//               tiles [32][D] stream through LDS (one ds_read_b128 per MFMA, no target restaging).
//   nn_search_kernel<T, WN>     (any D): both operands staged in 64-wide D chunks.
// Parallelism: grid = (target panels, P keyframes, SPLITS of the pivot range).  With SPLITS > 1
// every workgroup writes its (best score, index) per target to scratch and nn_finalize_kernel
// merges them in ascending split order (strict '>', so the first index still wins on ties).
#include <stdlib.h>

#include <type_traits>

#include "tf_common.h"

namespace {

// byte offset of 16-B piece `piece` of row `row` in a [rows][BK] 16-bit tile, BK = 64, 128 or 256 (D chunk).
// 128-B rows: two rows share a 256-B bank row -> XOR by (row >> 1) & 7; 256-B and 512-B rows: XOR by row & 15
// (16 consecutive rows reading the same piece fall into 16 different 16-B bank groups).
template <int BK>
__device__ __forceinline__ int swz_off(int row, int piece) {
    if constexpr (BK == 64) return row * 128 + ((piece ^ ((row >> 1) & 7)) << 4);
    return row * (BK * 2) + ((piece ^ (row & 15)) << 4);
}

// ---------------------------------------------------------------------------
// inv_norm[r] = 1 / ||piv[r]||_2, one wave per row.
template <typename T>
__global__ __launch_bounds__(256) void pivot_inv_norm_kernel(const typename T::elem* __restrict__ piv,
                                                             float* __restrict__ inv_norm, int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float inv = tf_row_inv_norm<T>(piv + r * D, D, lane);
        if (lane == 0) inv_norm[r] = inv;
    }
}

// ---------------------------------------------------------------------------
// WN = 32-target sub-tiles per wave (1 or 2): workgroup covers TN = 64*WN targets.
// BK = D chunk per barrier interval: 64, or 128 / 256 for small problems (few workgroups, where the per-iteration
//      global-load latency is exposed: halving the iteration count halves the run time).
// TM = pivots per tile: 128, or 64 with BK = 256 (the LDS holds two (TM + TN) x BK stages).
template <typename T, int WN, int BK, int TM>
__global__ __launch_bounds__(256) void nn_search_kernel(const typename T::elem* __restrict__ tgt,
                                                        const typename T::elem* __restrict__ piv,
                                                        const float* __restrict__ inv_norm,
                                                        int32_t* __restrict__ idx_out,
                                                        NnPartial* __restrict__ part_out, int64_t n_tgt, int S,
                                                        int D, int kf0, int kf1, int tiles_per_split, NnChunks ch) {
    typedef typename T::vec8 vec8;
    constexpr int TN = 64 * WN;
    constexpr int NI = TM / 64;              // 32-pivot sub-tiles per wave
    constexpr int WR = 32 * NI;              // pivot rows per wave half
    constexpr int PPR = BK / 8;              // 16-B pieces per chunk row
    constexpr int A_BYTES = TM * BK * 2;
    constexpr int B_BYTES = TN * BK * 2;
    constexpr int NPA = TM * PPR / 256;      // 16-B pieces per thread, A chunk
    constexpr int NPB = TN * PPR / 256;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    auto sA = [&](int b) { return smem + b * (A_BYTES + B_BYTES); };
    auto sB = [&](int b) { return smem + b * (A_BYTES + B_BYTES) + A_BYTES; };
    float* sInv = reinterpret_cast<float*>(smem + 2 * (A_BYTES + B_BYTES));  // [2][TM]
    float* sBestV = sInv + 2 * TM;                                           // [2][TN]
    int* sBestI = reinterpret_cast<int*>(sBestV + 2 * TN);                   // [2][TN]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1;  // pivot half  (rows wr*WR .. +WR-1 of the tile)
    const int wc = wave & 1;   // target half (cols wc*32*WN .. )
    const int hi = lane >> 5;
    const int l31 = lane & 31;

    const int p = blockIdx.y;
    // chunk of the video this target panel belongs to (0 in the single-chunk call): chunk j matches keyframe
    // slots kf0 + j and kf1 + j; the first chunk of the video has ONE keyframe (tokenflow_utils.py:331-333)
    const int chunk = blockIdx.x / ch.ppc;
    if (p == 1 && chunk == 0 && ch.first_single) return;
    const int kf = (p == 0 ? kf0 : kf1) + chunk;
    const typename T::elem* pv = piv + (int64_t)kf * S * D;
    const float* inv = inv_norm + (int64_t)kf * S;
    const int64_t t0 = chunk * ch.nS + (int64_t)(blockIdx.x - chunk * ch.ppc) * TN;
    const int64_t t_end = (chunk + 1) * ch.nS;   // targets of this chunk: [chunk * nS, t_end)

    const int n_mt_all = (S + TM - 1) / TM;
    const int mt0 = blockIdx.z * tiles_per_split;           // this workgroup's slice of the pivot tiles
    const int n_mt = min(tiles_per_split, n_mt_all - mt0);
    const int n_kc = (D + BK - 1) / BK;
    const int total = n_mt * n_kc;

    // per-thread staging pieces: piece id = tid + 256*i -> row = id / PPR, col piece = id % PPR.
    // TWO register sets: the loads of chunk it+2 are issued during chunk it and written to LDS at the end of
    // chunk it+1, so a global-load latency is spread over two iterations (the loop is a serial chain of
    // load -> LDS -> barrier -> MFMA steps; with one set every step paid the full latency, ~1.7 us, which
    // is all the time there is at the small levels).
    u32x4 ra2[2][NPA], rb2[2][NPB];
    float rinv2[2] = {0.f, 0.f};

    auto stage_load = [&](int it, auto SET) {
        constexpr int set = decltype(SET)::value;
        u32x4(&ra)[NPA] = ra2[set];
        u32x4(&rb)[NPB] = rb2[set];
        float& rinv = rinv2[set];
        const int mt = it / n_kc, kc = it - mt * n_kc;
        const int col0 = kc * BK;
        // Branch-free: every lane loads from a valid (clamped) address; columns past D are zeroed when the
        // piece is written to LDS.  A predicated load (col < D ? load : 0) becomes an exec-masked branch per
        // load and makes the compiler drain vmcnt(0) before the next batch -- no load ever overlapped a MFMA.
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int id = tid + 256 * i;
            const int r = id / PPR, pc = id % PPR;
            const int row = min((mt0 + mt) * TM + r, S - 1);   // clamped duplicates can never win (see epilogue)
            const int col = min(col0 + pc * 8, D - 8);
            ra[i] = ld16(pv + (int64_t)row * D + col);
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const int id = tid + 256 * i;
            const int r = id / PPR, pc = id % PPR;
            const int64_t row = min(t0 + r, t_end - 1);
            const int col = min(col0 + pc * 8, D - 8);
            rb[i] = ld16(tgt + row * D + col);
        }
        rinv = inv[min((mt0 + mt) * TM + min(tid, TM - 1), S - 1)];
    };
    auto stage_write = [&](int it, auto SET) {
        constexpr int set = decltype(SET)::value;
        u32x4(&ra)[NPA] = ra2[set];
        u32x4(&rb)[NPB] = rb2[set];
        const float rinv = rinv2[set];
        const int mt = it / n_kc, kc = it - mt * n_kc;
        const int b = it & 1;
        const int col0 = kc * BK;
        const u32x4 zero = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int id = tid + 256 * i;
            st16(sA(b) + swz_off<BK>(id / PPR, id % PPR), col0 + (id % PPR) * 8 < D ? ra[i] : zero);
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const int id = tid + 256 * i;
            st16(sB(b) + swz_off<BK>(id / PPR, id % PPR), col0 + (id % PPR) * 8 < D ? rb[i] : zero);
        }
        if (kc == 0 && tid < TM) sInv[(mt & 1) * TM + tid] = rinv;
    };

    f32x16 acc[NI][WN];
    float best_v[WN];
    int best_i[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        best_v[j] = -INFINITY;
        best_i[j] = 0;
    }

    typedef std::integral_constant<int, 0> Set0;
    typedef std::integral_constant<int, 1> Set1;
    stage_load(0, Set0{});
    stage_write(0, Set0{});
    if (total > 1) stage_load(1, Set1{});
    __syncthreads();

    // chunk `it` sits in LDS buffer it & 1; register set (it + 1) & 1 holds chunk it + 1, set it & 1 is free
    auto step = [&](int it, auto CUR, auto NXT) {
        const int mt = it / n_kc, kc = it - mt * n_kc;
        if (it + 2 < total) stage_load(it + 2, CUR);
        if (kc == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        const unsigned char* a = sA(it & 1);
        const unsigned char* b = sB(it & 1);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            vec8 fa[NI], fb[WN];
#pragma unroll
            for (int i = 0; i < NI; ++i)
                fa[i] = __builtin_bit_cast(vec8, ld16(a + swz_off<BK>(wr * WR + i * 32 + l31, ks * 2 + hi)));
#pragma unroll
            for (int j = 0; j < WN; ++j)
                fb[j] = __builtin_bit_cast(vec8, ld16(b + swz_off<BK>((wc * WN + j) * 32 + l31, ks * 2 + hi)));
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = T::mfma32(fa[i], fb[j], acc[i][j]);
        }
        if (kc == n_kc - 1) {
            // argmax epilogue: rows visited in ascending order, strict '>' keeps the first maximum
            const float* si = sInv + (mt & 1) * TM;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wr * WR + i * 32 + cd_row(r, hi);
                    const float w = si[rl];
                    const int gi = (mt0 + mt) * TM + rl;
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        const float sc = acc[i][j][r] * w;
                        if (sc > best_v[j]) {
                            best_v[j] = sc;
                            best_i[j] = gi;
                        }
                    }
                }
            }
        }
        if (it + 1 < total) stage_write(it + 1, NXT);
        __syncthreads();
    };
    for (int it = 0; it < total; it += 2) {   // unrolled by two so the register sets are compile-time names
        step(it, Set0{}, Set1{});
        if (it + 1 < total) step(it + 1, Set1{}, Set0{});
    }

    // merge lane l with lane l+32 (interleaved row sets): tie -> smaller index
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const float ov = __shfl_xor(best_v[j], 32);
        const int oi = __shfl_xor(best_i[j], 32);
        if (ov > best_v[j] || (ov == best_v[j] && oi < best_i[j])) {
            best_v[j] = ov;
            best_i[j] = oi;
        }
        if (hi == 0) {
            const int col = (wc * WN + j) * 32 + l31;
            sBestV[wr * TN + col] = best_v[j];
            sBestI[wr * TN + col] = best_i[j];
        }
    }
    __syncthreads();
    if (tid < TN) {
        float v0 = sBestV[tid], v1 = sBestV[TN + tid];
        int i0 = sBestI[tid], i1 = sBestI[TN + tid];
        if (v1 > v0 || (v1 == v0 && i1 < i0)) {
            i0 = i1;
            v0 = v1;
        }
        i0 = i0 < S ? i0 : S - 1;  // a clamped duplicate of row S-1 maps back to S-1
        const int64_t t = t0 + tid;
        if (t < t_end) {
            if (part_out)
                part_out[((int64_t)blockIdx.z * gridDim.y + p) * n_tgt + t] = NnPartial{v0, i0};
            else
                idx_out[(int64_t)p * n_tgt + t] = i0;
        }
    }
}

// ---------------------------------------------------------------------------
// LDS-DMA variant for D % 64 == 0, D >= 512 (levels 1-2 of SD: D = 640, 1280) on grids that fill the chip.
// The generic kernel above is LDS-bound there: a wave's 64 x 64 tile reads 4 fragments of 1 KB per 4 MFMAs and
// re-stages (TM + TN) x BK per interval through registers -- 1.5 KB of LDS traffic per MFMA against the 1 KB the LDS
// pipe (128 B/clk per CU) delivers in the 32 clocks an MFMA occupies one of the CU's four matrix pipes (counters:
// profiles/r04_pmc_level1.csv).  Here a wave owns 64 pivots x 128 targets (6 fragment reads per 8 MFMAs) inside a
// 256 x 256 workgroup tile of 8 waves (4 pivot quarters x 2 target halves): 0.75 + 0.25 KB per MFMA.  The 128
// accumulator registers leave no room for two register staging sets (profiles/r04_nn_wide4_ab.txt: spills), so the
// tiles go global -> LDS directly (`global_load_lds_dwordx4`: no staging registers, no ds_write pass).  The DMA
// writes lane-linearly (wave-uniform base + lane * 16 B): the XOR swizzle of the image is applied on the SOURCE side
// (the lane that fills slot s of row r fetches piece s ^ ((r >> 1) & 7) of that row) and on the read side, the same
// involution (swz_off<64>).  Two LDS buffers; the DMA of chunk it+1 is issued before the MFMAs of chunk it and
// drained (vmcnt(0)) in front of the interval's barrier.  The pivots' inverse norms of a tile arrive the same way.
// Arithmetic per (target, pivot) and the first-index rule are those of nn_search_kernel.
// SH (round 6, last session): the wave's 64 x 128 tile as 4 x 8 sub-tiles of v_mfma_f32_16x16x32 instead of 2 x 4 of 32x32x16 -- same
// images, same bytes through the LDS, same matrix-pipe clocks and accumulator registers; the short shape sustains more on these
// power-limited boxes (see nn_search_rbs_kernel).  32 features per MFMA: scores differ from the SH = false kernel's in the last bit.
template <typename T, bool SH = false>
__global__ __launch_bounds__(512, 2) void nn_search_glds_kernel(const typename T::elem* __restrict__ tgt,
                                                                const typename T::elem* __restrict__ piv,
                                                                const float* __restrict__ inv_norm,
                                                                int32_t* __restrict__ idx_out,
                                                                NnPartial* __restrict__ part_out, int64_t n_tgt, int S,
                                                                int D, int kf0, int kf1, int tiles_per_split, NnChunks ch) {
    typedef typename T::vec8 vec8;
    typedef typename T::elem E;
    constexpr int TM = 256, TN = 256, BK = 64, NI = 2, WN = 4;
    constexpr int A_BYTES = TM * BK * 2, B_BYTES = TN * BK * 2;   // 32 KB each
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    auto sA = [&](int b) { return smem + b * (A_BYTES + B_BYTES); };
    auto sB = [&](int b) { return smem + b * (A_BYTES + B_BYTES) + A_BYTES; };
    float* sInv = reinterpret_cast<float*>(smem + 2 * (A_BYTES + B_BYTES));   // [2][TM]
    float* sBestV = sInv + 2 * TM;                                            // [4][TN]
    int* sBestI = reinterpret_cast<int*>(sBestV + 4 * TN);                    // [4][TN]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1;   // pivot quarter (rows wr*64 .. +63 of the tile)
    const int wc = wave & 1;    // target half   (cols wc*128 .. +127)
    const int hi = lane >> 5;
    const int l31 = lane & 31;
    const int g = lane >> 4, n16 = lane & 15;   // SH: 16-lane row (k-block of A / B, row group of C), row / column in a sub-tile
    constexpr int NB = SH ? 2 * WN : WN;        // running (max, argmax) pairs per lane: one per target sub-tile

    const int p = blockIdx.y;
    const int chunk = blockIdx.x / ch.ppc;
    if (p == 1 && chunk == 0 && ch.first_single) return;
    const int kf = (p == 0 ? kf0 : kf1) + chunk;
    const E* pv = piv + (int64_t)kf * S * D;
    const float* inv = inv_norm + (int64_t)kf * S;
    const int64_t t0 = chunk * ch.nS + (int64_t)(blockIdx.x - chunk * ch.ppc) * TN;
    const int64_t t_end = (chunk + 1) * ch.nS;

    const int n_mt_all = (S + TM - 1) / TM;
    const int mt0 = blockIdx.z * tiles_per_split;
    const int n_mt = min(tiles_per_split, n_mt_all - mt0);
    const int n_kc = D / BK;
    const int total = n_mt * n_kc;

    // DMA pieces of this lane: wave w fills rows (w*4 + j)*8 .. +7 of both images, j = 0..3; lane -> row l/8, slot l%8
    const int sub = lane >> 3, slot = lane & 7;
    const E* b_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 8 + sub;
        const int64_t row = min(t0 + r, t_end - 1);            // clamped duplicates: never written back
        b_src[j] = tgt + row * D + ((slot ^ ((r >> 1) & 7)) << 3);
    }
    int a_off[4];   // element offset of this lane's piece inside the pivot tile (re-clamped per tile)
    auto set_tile = [&](int mt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (wave * 4 + j) * 8 + sub;
            const int row = min((mt0 + mt) * TM + r, S - 1);   // clamped duplicates can never win (see epilogue)
            a_off[j] = row * D + ((slot ^ ((r >> 1) & 7)) << 3);
        }
    };
    // DMA pieces j = 0..3 of chunk `it` (one A and one B piece each), issued as one block in front of the k-loop of the
    // chunk in flight.  Spreading them over the k-steps, each pair behind a k-step's 8 MFMAs and pinned there with
    // sched_barriers (TF_TUNE_GLDS_SPREAD_ISSUE), measured 1-4 % slower (profiles/r05_nn_glds_ab.txt): with two waves per
    // SIMD the other wave's MFMAs already cover a wave's issue time, and the pinning costs the compiler its own order.
#ifdef TF_TUNE_GLDS_SPREAD_ISSUE
    constexpr bool SPREAD = true;
#else
    constexpr bool SPREAD = false;
#endif
    auto stage_piece = [&](int it, int j) {
        const int mt = it / n_kc, kc = it - mt * n_kc;
        const int b = it & 1;
        if (j == 0 && kc == 0) {
            set_tile(mt);
            if (wave == 0) {   // the tile's inverse norms: 256 floats = one 1 KB DMA piece
                const int row = min((mt0 + mt) * TM + lane * 4, S - 4);
                __builtin_amdgcn_global_load_lds((glb_ptr)(inv + row), (lds_ptr)(sInv + (mt & 1) * TM), 16, 0, 0);
            }
        }
        const int col = kc * BK;
        __builtin_amdgcn_global_load_lds((glb_ptr)(pv + a_off[j] + col), (lds_ptr)(sA(b) + (wave * 4 + j) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr)(b_src[j] + col), (lds_ptr)(sB(b) + (wave * 4 + j) * 1024), 16, 0, 0);
    };
    auto stage = [&](int it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) stage_piece(it, j);
    };

    f32x16 acc[NI][WN];
    f32x4 acc16[2 * NI][2 * WN];   // SH
    float best_v[NB];
    int best_i[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        best_v[j] = -INFINITY;
        best_i[j] = 0;
    }

    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int it = 0; it < total; ++it) {
        const int mt = it / n_kc, kc = it - mt * n_kc;
        // buffer (it + 1) & 1 was last read in interval it - 1; every wave has passed the barrier that ended it
        const bool more = it + 1 < total;
        if (!SPREAD && more) stage(it + 1);
        if (kc == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
            for (int i = 0; i < 2 * NI; ++i)
#pragma unroll
                for (int j = 0; j < 2 * WN; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const unsigned char* a = sA(it & 1);
        const unsigned char* b = sB(it & 1);
        if constexpr (SH) {
#pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
                vec8 fa[2 * NI], fb[2 * WN];
#pragma unroll
                for (int i = 0; i < 2 * NI; ++i)
                    fa[i] = __builtin_bit_cast(vec8, ld16(a + swz_off<BK>(wr * 64 + i * 16 + n16, ks * 4 + g)));
#pragma unroll
                for (int j = 0; j < 2 * WN; ++j)
                    fb[j] = __builtin_bit_cast(vec8, ld16(b + swz_off<BK>((wc * 2 * WN + j) * 16 + n16, ks * 4 + g)));
#pragma unroll
                for (int i = 0; i < 2 * NI; ++i)
#pragma unroll
                    for (int j = 0; j < 2 * WN; ++j) acc16[i][j] = T::mfma16(fa[i], fb[j], acc16[i][j]);
            }
        } else
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            vec8 fa[NI], fb[WN];
#pragma unroll
            for (int i = 0; i < NI; ++i)
                fa[i] = __builtin_bit_cast(vec8, ld16(a + swz_off<BK>(wr * 64 + i * 32 + l31, ks * 2 + hi)));
#pragma unroll
            for (int j = 0; j < WN; ++j)
                fb[j] = __builtin_bit_cast(vec8, ld16(b + swz_off<BK>((wc * WN + j) * 32 + l31, ks * 2 + hi)));
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = T::mfma32(fa[i], fb[j], acc[i][j]);
            if (SPREAD) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) stage_piece(it + 1, ks);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (kc == n_kc - 1) {
            // argmax epilogue: rows visited in ascending order, strict '>' keeps the first maximum
            const float* si = sInv + (mt & 1) * TM;
            const int last = S - 1 - (mt0 + mt) * TM;   // rows past S are copies of row S - 1: give them ITS inverse norm
            if constexpr (SH) {
#pragma unroll
                for (int i = 0; i < 2 * NI; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rl = wr * 64 + i * 16 + 4 * g + r;   // ascending within the lane
                        const float w = si[min(rl, last)];
                        const int gi = (mt0 + mt) * TM + rl;
#pragma unroll
                        for (int j = 0; j < 2 * WN; ++j) {
                            const float sc = acc16[i][j][r] * w;
                            if (sc > best_v[j]) {
                                best_v[j] = sc;
                                best_i[j] = gi;
                            }
                        }
                    }
            } else
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wr * 64 + i * 32 + cd_row(r, hi);
                    const float w = si[min(rl, last)];
                    const int gi = (mt0 + mt) * TM + rl;
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        const float sc = acc[i][j][r] * w;
                        if (sc > best_v[j]) {
                            best_v[j] = sc;
                            best_i[j] = gi;
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if constexpr (SH) {   // the four 16-lane rows hold disjoint pivot rows of a target: merge, smaller index on equal scores
#pragma unroll
        for (int j = 0; j < NB; ++j) {
#pragma unroll
            for (int o_ = 16; o_ <= 32; o_ <<= 1) {
                const float ov = __shfl_xor(best_v[j], o_);
                const int oi = __shfl_xor(best_i[j], o_);
                if (ov > best_v[j] || (ov == best_v[j] && oi < best_i[j])) {
                    best_v[j] = ov;
                    best_i[j] = oi;
                }
            }
            if (g == 0) {
                const int col = (wc * NB + j) * 16 + n16;
                sBestV[wr * TN + col] = best_v[j];
                sBestI[wr * TN + col] = best_i[j];
            }
        }
    } else
    // merge lane l with lane l+32 (interleaved row sets): tie -> smaller index
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const float ov = __shfl_xor(best_v[j], 32);
        const int oi = __shfl_xor(best_i[j], 32);
        if (ov > best_v[j] || (ov == best_v[j] && oi < best_i[j])) {
            best_v[j] = ov;
            best_i[j] = oi;
        }
        if (hi == 0) {
            const int col = (wc * WN + j) * 32 + l31;
            sBestV[wr * TN + col] = best_v[j];
            sBestI[wr * TN + col] = best_i[j];
        }
    }
    __syncthreads();
    if (tid < TN) {
        float v0 = sBestV[tid];
        int i0 = sBestI[tid];
#pragma unroll
        for (int q = 1; q < 4; ++q) {   // ascending pivot quarters: strict '>' keeps the first maximum
            const float v1 = sBestV[q * TN + tid];
            const int i1 = sBestI[q * TN + tid];
            if (v1 > v0 || (v1 == v0 && i1 < i0)) {
                i0 = i1;
                v0 = v1;
            }
        }
        i0 = i0 < S ? i0 : S - 1;
        const int64_t t = t0 + tid;
        if (t < t_end) {
            if (part_out)
                part_out[((int64_t)blockIdx.z * gridDim.y + p) * n_tgt + t] = NnPartial{v0, i0};
            else
                idx_out[(int64_t)p * n_tgt + t] = i0;
        }
    }
}

// ---------------------------------------------------------------------------
// Register-B variant for D = 16*DK: one wave = 32 targets whose B fragments stay in registers.
template <typename T, int DK>
__global__ __launch_bounds__(256, 3) void nn_search_rb_kernel(const typename T::elem* __restrict__ tgt,
                                                           const typename T::elem* __restrict__ piv,
                                                           const float* __restrict__ inv_norm,
                                                           int32_t* __restrict__ idx_out,
                                                           NnPartial* __restrict__ part_out, int64_t n_tgt, int S,
                                                           int kf0, int kf1, int tiles_per_split, NnChunks ch) {
    typedef typename T::elem E;
    typedef typename T::vec8 vec8;
    constexpr int D = 16 * DK;
    constexpr int TMR = 32;                  // pivots per tile
    constexpr int RS = D + 8;                // LDS row stride (elements): odd number of 16-B slots
    constexpr int PPR = D / 8;               // 16-B pieces per row
    constexpr int NP = (TMR * PPR + 255) / 256;
    constexpr int A_ELEMS = TMR * RS;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    auto sA = [&](int b) { return reinterpret_cast<E*>(smem) + b * A_ELEMS; };
    float* sInv = reinterpret_cast<float*>(smem + 2 * A_ELEMS * sizeof(E));  // [2][TMR]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hi = lane >> 5;
    const int l31 = lane & 31;
    const int p = blockIdx.y;
    const int chunk = blockIdx.x / ch.ppc;       // see nn_search_kernel
    if (p == 1 && chunk == 0 && ch.first_single) return;
    const int kf = (p == 0 ? kf0 : kf1) + chunk;
    const E* pv = piv + (int64_t)kf * S * D;
    const float* inv = inv_norm + (int64_t)kf * S;
    const int64_t t_end = (chunk + 1) * ch.nS;
    const int64_t t_row = chunk * ch.nS + (int64_t)(blockIdx.x - chunk * ch.ppc) * 128 + wave * 32 + l31;

    const int n_mt_all = (S + TMR - 1) / TMR;
    const int mt0 = blockIdx.z * tiles_per_split;
    const int n_mt = min(tiles_per_split, n_mt_all - mt0);

    vec8 fb[DK];
    {
        const E* tp = tgt + (t_row < t_end ? t_row : t_end - 1) * D + 8 * hi;
#pragma unroll
        for (int t = 0; t < DK; ++t) fb[t] = __builtin_bit_cast(vec8, ld16(tp + 16 * t));
    }

    u32x4 ra[NP];
    float rinv = 0.f;
    auto stage_load = [&](int mt) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int id = tid + 256 * i;
            if (id < TMR * PPR) {
                const int r = id / PPR, pc = id - r * PPR;
                int row = (mt0 + mt) * TMR + r;
                row = row < S ? row : S - 1;
                ra[i] = ld16(pv + (int64_t)row * D + pc * 8);
            }
        }
        if (tid < TMR) {
            int row = (mt0 + mt) * TMR + tid;
            rinv = inv[row < S ? row : S - 1];
        }
    };
    auto stage_write = [&](int mt) {
        E* a = sA(mt & 1);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int id = tid + 256 * i;
            if (id < TMR * PPR) {
                const int r = id / PPR, pc = id - r * PPR;
                st16(a + r * RS + pc * 8, ra[i]);
            }
        }
        if (tid < TMR) sInv[(mt & 1) * TMR + tid] = rinv;
    };

    float best_v = -INFINITY;
    int best_i = 0;
    stage_load(0);
    stage_write(0);
    __syncthreads();
    for (int mt = 0; mt < n_mt; ++mt) {
        const bool has_next = mt + 1 < n_mt;
        if (has_next) stage_load(mt + 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const E* arow = sA(mt & 1) + l31 * RS + 8 * hi;
#pragma unroll
        for (int t = 0; t < DK; ++t) acc = T::mfma32(__builtin_bit_cast(vec8, ld16(arow + 16 * t)), fb[t], acc);
        const float* si = sInv + (mt & 1) * TMR;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = cd_row(r, hi);
            const float sc = acc[r] * si[rl];
            if (sc > best_v) {
                best_v = sc;
                best_i = (mt0 + mt) * TMR + rl;
            }
        }
        if (has_next) stage_write(mt + 1);
        __syncthreads();
    }
    const float ov = __shfl_xor(best_v, 32);
    const int oi = __shfl_xor(best_i, 32);
    if (ov > best_v || (ov == best_v && oi < best_i)) {
        best_v = ov;
        best_i = oi;
    }
    best_i = best_i < S ? best_i : S - 1;
    if (hi == 0 && t_row < t_end) {
        if (part_out)
            part_out[((int64_t)blockIdx.z * gridDim.y + p) * n_tgt + t_row] = NnPartial{best_v, best_i};
        else
            idx_out[(int64_t)p * n_tgt + t_row] = best_i;
    }
}

// ---------------------------------------------------------------------------
// Register-B variant with the pivot tiles staged by LDS-DMA (global_load_lds_dwordx4) instead of through registers:
// no staging VGPRs, no ds_write pass.  The DMA writes lane-linearly, so the image has no row padding; bank conflicts of
// the fragment reads (32 rows, 640-B stride) are avoided by an XOR swizzle of the 16-B piece index with (row >> 1) & 7,
// applied on the SOURCE address of the DMA and on the read (the same involution; it permutes inside aligned groups of
// 8 pieces, D / 8 being a multiple of 8).  Same arithmetic as nn_search_rbg_kernel.
// TT = target tiles (of 32 rows) per wave.  TT = 2 (round 6): a wave keeps 64 targets in registers (160 VGPRs of B fragments),
// every pivot fragment read from LDS feeds TWO MFMAs on two independent accumulator chains, and a barrier interval carries
// twice the matrix work -- half the LDS traffic and half the barriers per MFMA; 2 waves per SIMD instead of 3.  The
// arithmetic of a (target, pivot) pair is unchanged (same contraction order): the kernels are interchangeable bit for bit.
template <typename T, int DK, int TT = 1>
__global__ __launch_bounds__(256, TT == 1 ? 3 : 2) void nn_search_rbg_kernel(const typename T::elem* __restrict__ tgt,
                                                           const typename T::elem* __restrict__ piv,
                                                           const float* __restrict__ inv_norm,
                                                           int32_t* __restrict__ idx_out,
                                                           NnPartial* __restrict__ part_out, int64_t n_tgt, int S,
                                                           int kf0, int kf1, int tiles_per_split, NnChunks ch) {
    typedef typename T::elem E;
    typedef typename T::vec8 vec8;
    constexpr int D = 16 * DK;
    constexpr int TMR = 32;                  // pivots per tile
    constexpr int RS = D;                    // LDS row stride (elements): the DMA image is dense
    static_assert((D / 8) % 8 == 0, "the swizzle permutes inside groups of 8 pieces");
    constexpr int NPIECE = TMR * D * 2 / 1024;   // 1 KB DMA pieces per tile (20 at D = 320)
    static_assert(NPIECE % 4 == 0, "pieces split evenly over the 4 waves");
    constexpr int A_ELEMS = TMR * RS;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    auto sA = [&](int b) { return reinterpret_cast<E*>(smem) + b * A_ELEMS; };
    float* sInv = reinterpret_cast<float*>(smem + 2 * A_ELEMS * sizeof(E));  // [2][TMR]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hi = lane >> 5;
    const int l31 = lane & 31;
    const int p = blockIdx.y;
    const int chunk = blockIdx.x / ch.ppc;       // see nn_search_kernel
    if (p == 1 && chunk == 0 && ch.first_single) return;
    const int kf = (p == 0 ? kf0 : kf1) + chunk;
    const E* pv = piv + (int64_t)kf * S * D;
    const float* inv = inv_norm + (int64_t)kf * S;
    const int64_t t_end = (chunk + 1) * ch.nS;
    int64_t t_row[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
        t_row[tt] = chunk * ch.nS + (int64_t)(blockIdx.x - chunk * ch.ppc) * (128 * TT) + (wave * TT + tt) * 32 + l31;

    const int n_mt_all = (S + TMR - 1) / TMR;
    const int mt0 = blockIdx.z * tiles_per_split;
    const int n_mt = min(tiles_per_split, n_mt_all - mt0);

    vec8 fb[TT][DK];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const E* tp = tgt + (t_row[tt] < t_end ? t_row[tt] : t_end - 1) * D + 8 * hi;
#pragma unroll
        for (int t = 0; t < DK; ++t) fb[tt][t] = __builtin_bit_cast(vec8, ld16(tp + 16 * t));
    }

    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // this lane's slot of DMA piece j = wave + 4 * i: byte o of the image -> row o / (2 D), position (o % (2 D)) / 16
    // (S % 32 == 0 in this variant: every tile is full, no row clamp)
    int a_off[NPIECE / 4];
#pragma unroll
    for (int i = 0; i < NPIECE / 4; ++i) {
        const int o = (wave_u + 4 * i) * 1024 + lane * 16;
        const int r = o / (2 * D), pos = (o - r * 2 * D) >> 4;
        a_off[i] = r * D + ((pos ^ ((r >> 1) & 7)) << 3);
    }
    float rinv = 0.f;
    auto stage_load = [&](int mt) {
        unsigned char* a = reinterpret_cast<unsigned char*>(sA(mt & 1));
        const E* tile = pv + (int64_t)(mt0 + mt) * TMR * D;
#pragma unroll
        for (int i = 0; i < NPIECE / 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr)(tile + a_off[i]), (lds_ptr)(a + (wave_u + 4 * i) * 1024), 16, 0, 0);
        if (tid < TMR) {
            int row = (mt0 + mt) * TMR + tid;
            rinv = inv[row < S ? row : S - 1];
        }
    };
    auto stage_write = [&](int mt) {
        if (tid < TMR) sInv[(mt & 1) * TMR + tid] = rinv;
    };

    float best_v[TT];
    int best_i[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) best_v[tt] = -INFINITY, best_i[tt] = 0;
    stage_load(0);
    stage_write(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int mt = 0; mt < n_mt; ++mt) {
        const bool has_next = mt + 1 < n_mt;
        if (has_next) stage_load(mt + 1);
        f32x16 acc[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;
        const E* arow = sA(mt & 1) + l31 * RS;
        const int swz = (l31 >> 1) & 7;
#pragma unroll
        for (int t = 0; t < DK; ++t) {
            const vec8 fa = __builtin_bit_cast(vec8, ld16(arow + (((2 * t + hi) ^ swz) << 3)));
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) acc[tt] = T::mfma32(fa, fb[tt][t], acc[tt]);
        }
        const float* si = sInv + (mt & 1) * TMR;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = cd_row(r, hi);
            const float w = si[rl];
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const float sc = acc[tt][r] * w;
                if (sc > best_v[tt]) {
                    best_v[tt] = sc;
                    best_i[tt] = (mt0 + mt) * TMR + rl;
                }
            }
        }
        if (has_next) stage_write(mt + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const float ov = __shfl_xor(best_v[tt], 32);
        const int oi = __shfl_xor(best_i[tt], 32);
        float bv = best_v[tt];
        int bi = best_i[tt];
        if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
        }
        bi = bi < S ? bi : S - 1;
        if (hi == 0 && t_row[tt] < t_end) {
            if (part_out)
                part_out[((int64_t)blockIdx.z * gridDim.y + p) * n_tgt + t_row[tt]] = NnPartial{bv, bi};
            else
                idx_out[(int64_t)p * n_tgt + t_row[tt]] = bi;
        }
    }
}

// ---------------------------------------------------------------------------
// The two-target-tile kernel above with SHORT MFMAs (round 6, last session): v_mfma_f32_16x16x32 instead of 32x32x16.  Same
// workgroup (4 waves x 64 targets), same LDS image of a pivot tile (32 x D, dense, DMA-staged, pieces XOR-swizzled with
// (row >> 1) & 7 -- conflict-free for this read pattern too), same bytes through the LDS per flop, same matrix-pipe clocks:
// a wave's 64 targets are four 16-target B-fragment sets (160 VGPRs), a pivot tile two 16-row A fragments per 32-wide k-step,
// each feeding four MFMAs.  Why: on these power-limited boxes the short shape sustains ~16 % more in an MFMA-bound loop
// (tools/ubench/nn_loop_proxy.hip: 1 810 against 1 555 TF/s in the stripped loop of this kernel; the bare MFMAs 2 213 against
// 1 894, profiles/r02_mfma_shapes.txt) -- less accumulator traffic per flop.
// TJ = 16-target sub-tiles per wave: 4 (the two-target-tile form: 256 targets per workgroup) or 2 (the one-tile form: 128).
// Arithmetic: a (target, pivot) dot product is summed 32 features per MFMA in ascending order -- NOT bit-identical to the
// 16-wide steps of the other search kernels (fp32 rounding of the partial sums); exact duplicates still tie exactly (the same
// instruction sequence on the same data) and the first index wins as everywhere (ascending rows, strict '>').
template <typename T, int DK, int TJ = 4>
__global__ __launch_bounds__(256, TJ == 4 ? 2 : 3) void nn_search_rbs_kernel(const typename T::elem* __restrict__ tgt,
                                                               const typename T::elem* __restrict__ piv,
                                                               const float* __restrict__ inv_norm,
                                                               int32_t* __restrict__ idx_out,
                                                               NnPartial* __restrict__ part_out, int64_t n_tgt, int S,
                                                               int kf0, int kf1, int tiles_per_split, NnChunks ch) {
    typedef typename T::elem E;
    typedef typename T::vec8 vec8;
    constexpr int D = 16 * DK;
    constexpr int KS32 = D / 32;             // 32-wide k-steps
    static_assert(D % 32 == 0 && (D / 8) % 8 == 0, "32-wide k-steps; the swizzle permutes inside groups of 8 pieces");
    constexpr int TMR = 32;                  // pivots per tile
    constexpr int NPIECE = TMR * D * 2 / 1024;
    static_assert(NPIECE % 4 == 0, "pieces split evenly over the 4 waves");
    constexpr int A_ELEMS = TMR * D;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    auto sA = [&](int b) { return reinterpret_cast<E*>(smem) + b * A_ELEMS; };
    float* sInv = reinterpret_cast<float*>(smem + 2 * A_ELEMS * sizeof(E));  // [2][TMR]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = lane >> 4;      // 16-lane row: the k-block of the A / B fragments, the row group of C
    const int n16 = lane & 15;    // A: pivot row of the 16-row sub-tile; B / C: target of the 16-target sub-tile
    const int p = blockIdx.y;
    const int chunk = blockIdx.x / ch.ppc;       // see nn_search_kernel
    if (p == 1 && chunk == 0 && ch.first_single) return;
    const int kf = (p == 0 ? kf0 : kf1) + chunk;
    const E* pv = piv + (int64_t)kf * S * D;
    const float* inv = inv_norm + (int64_t)kf * S;
    const int64_t t_end = (chunk + 1) * ch.nS;
    int64_t t_row[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j)
        t_row[j] = chunk * ch.nS + (int64_t)(blockIdx.x - chunk * ch.ppc) * (64 * TJ) + wave * (16 * TJ) + 16 * j + n16;

    const int n_mt_all = (S + TMR - 1) / TMR;
    const int mt0 = blockIdx.z * tiles_per_split;
    const int n_mt = min(tiles_per_split, n_mt_all - mt0);

    vec8 fb[TJ][KS32];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const E* tp = tgt + (t_row[j] < t_end ? t_row[j] : t_end - 1) * D + 8 * g;
#pragma unroll
        for (int t = 0; t < KS32; ++t) fb[j][t] = __builtin_bit_cast(vec8, ld16(tp + 32 * t));
    }

    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int a_off[NPIECE / 4];
#pragma unroll
    for (int i = 0; i < NPIECE / 4; ++i) {
        const int o = (wave_u + 4 * i) * 1024 + lane * 16;
        const int r = o / (2 * D), pos = (o - r * 2 * D) >> 4;
        a_off[i] = r * D + ((pos ^ ((r >> 1) & 7)) << 3);
    }
    float rinv = 0.f;
    auto stage_load = [&](int mt) {
        unsigned char* a = reinterpret_cast<unsigned char*>(sA(mt & 1));
        const E* tile = pv + (int64_t)(mt0 + mt) * TMR * D;
#pragma unroll
        for (int i = 0; i < NPIECE / 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr)(tile + a_off[i]), (lds_ptr)(a + (wave_u + 4 * i) * 1024), 16, 0, 0);
        if (tid < TMR) {
            int row = (mt0 + mt) * TMR + tid;
            rinv = inv[row < S ? row : S - 1];
        }
    };
    auto stage_write = [&](int mt) {
        if (tid < TMR) sInv[(mt & 1) * TMR + tid] = rinv;
    };

    float best_v[TJ];
    int best_i[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) best_v[j] = -INFINITY, best_i[j] = 0;
    stage_load(0);
    stage_write(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int mt = 0; mt < n_mt; ++mt) {
        const bool has_next = mt + 1 < n_mt;
        if (has_next) stage_load(mt + 1);
        f32x4 acc[2][TJ];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const E* a0 = sA(mt & 1);
#pragma unroll
        for (int t = 0; t < KS32; ++t) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = 16 * i + n16;   // pivot row of the tile; its 16-B piece 4 t + g, swizzled as stored
                const vec8 fa = __builtin_bit_cast(vec8, ld16(a0 + r * D + (((4 * t + g) ^ ((r >> 1) & 7)) << 3)));
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = T::mfma16(fa, fb[j][t], acc[i][j]);
            }
        }
        const float* si = sInv + (mt & 1) * TMR;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = 16 * i + 4 * g + r;   // ascending within the lane: first index wins with the strict '>'
                const float w = si[rl];
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const float sc = acc[i][j][r] * w;
                    if (sc > best_v[j]) {
                        best_v[j] = sc;
                        best_i[j] = (mt0 + mt) * TMR + rl;
                    }
                }
            }
        if (has_next) stage_write(mt + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // the four 16-lane rows hold disjoint pivot rows of the same target: merge, lower index on equal scores
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        float bv = best_v[j];
        int bi = best_i[j];
#pragma unroll
        for (int o_ = 16; o_ <= 32; o_ <<= 1) {
            const float ov = __shfl_xor(bv, o_);
            const int oi = __shfl_xor(bi, o_);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        bi = bi < S ? bi : S - 1;
        if (g == 0 && t_row[j] < t_end) {
            if (part_out)
                part_out[((int64_t)blockIdx.z * gridDim.y + p) * n_tgt + t_row[j]] = NnPartial{bv, bi};
            else
                idx_out[(int64_t)p * n_tgt + t_row[j]] = bi;
        }
    }
}

// merge the per-split candidates: ascending split order == ascending pivot index
__global__ __launch_bounds__(256) void nn_finalize_kernel(const NnPartial* __restrict__ part,
                                                          int32_t* __restrict__ idx_out, int64_t total, int splits) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    NnPartial b = part[g];
    for (int s = 1; s < splits; ++s) {
        const NnPartial c = part[(int64_t)s * total + g];
        if (c.v > b.v || (c.v == b.v && c.i < b.i)) b = c;
    }
    idx_out[g] = b.i;
}

// Launch plan shared by the launchers and tf_nn_search_workspace_bytes.
enum NnKernel {
    NN_RB,      // register-B kernel (D == 320): 128-target panels, 32-pivot tiles
    NN_RB2,     // the same with two target tiles per wave (256-target panels): chip-filling grids, S % 32 == 0
    NN_WIDE,    // generic kernel, 128-target panels, 64-wide D chunks
    NN_BK64,    // generic kernel, 64-target panels, 64-wide D chunks
    NN_BK128,   // few workgroups and a long contraction: 128-wide D chunks
    NN_DEEP,    // fewer still, D >= 1024: 64-pivot tiles and 256-wide D chunks (a quarter of the barrier intervals)
    NN_GLDS     // D % 64 == 0, D >= 512, chip-filling grids: 256 x 256 workgroup tiles staged by LDS-DMA
};
struct NnPlan {
    int kern;
    int64_t panels;
    int splits, tiles_per_split;
};

// Grid target of the pivot-range split: workgroups the launch should have at least.  1024 = 4 per CU for one chunk.
// A multi-chunk launch is C times larger; it keeps splitting up to TF_NN_MIN_WGS (environment, read once; default
// 4096) so that the last round of workgroups is a small fraction of the launch.
static int nn_min_wgs(int C) {
    static const int env = [] {
        const char* e = getenv("TF_NN_MIN_WGS");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 4096;
    }();
    return C > 1 ? env : 1024;
}

// splits of the pivot range so that the grid has >= nn_min_wgs workgroups while every split keeps >= 1 tile.
// n_tgt = targets of ONE chunk, C = chunks in the launch (grid.x = C * panels).
static NnPlan nn_plan(int64_t n_tgt, int S, int D, int P, int C = 1) {
    NnPlan pl;
    auto shape = [&](int tn, int tm) {
        pl.panels = (n_tgt + tn - 1) / tn;
        const int n_tiles = (S + tm - 1) / tm;
        int splits = 1;
        // a multi-chunk launch keeps >= 128 pivots per split: below that the per-workgroup fixed cost (target
        // fragments, partial results, their merge) outweighs the finer tail (cfg1 level 0: 41.8 us at 128 pivots /
        // split, 72 us at 32)
        const int min_tiles = C > 1 ? (128 + tm - 1) / tm : 1;
        while (pl.panels * C * P * splits < nn_min_wgs(C) && splits * 2 <= n_tiles && n_tiles / (splits * 2) >= min_tiles)
            splits *= 2;
        pl.tiles_per_split = (n_tiles + splits - 1) / splits;
        pl.splits = (n_tiles + pl.tiles_per_split - 1) / pl.tiles_per_split;
        return pl.panels * C * P * pl.splits;   // workgroups of the launch
    };
#ifndef TF_TUNE_NN_NO_GLDS   // A/B switch of tools/build_variants.sh
    // LDS-DMA kernel (256-target panels, 256-pivot tiles, one 8-wave workgroup per CU): where the launch still has
    // >= TF_NN_GLDS_MIN_WGS workgroups after splitting the pivot range down to one tile per workgroup
    static const int glds_min_d = [] { const char* e = getenv("TF_NN_GLDS_MIN_D"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
    // D = 320 keeps the register-B kernel at the large levels (1.00 vs 1.12 ms at cfg2 level 0); with <= 1024 pivots per
    // keyframe (BASELINE config 1, level 0) its 32-pivot tiles are mostly prologue: 44 -> 33 us (profiles/r05_nn_glds_ab.txt)
    if (D % 64 == 0 && (D >= glds_min_d || (D == 320 && S <= 1024)) && S % 4 == 0) {
        static const int min_wgs = [] { const char* e = getenv("TF_NN_GLDS_MIN_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 160; }();
        static const int rounds = [] { const char* e = getenv("TF_NN_GLDS_ROUNDS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 3; }();
        const int64_t panels = (n_tgt + 255) / 256;
        const int n_tiles = (S + 255) / 256;
        if (panels * C * P * n_tiles >= min_wgs) {
            pl.kern = NN_GLDS;
            pl.panels = panels;
            // one workgroup per CU: aim at >= 3 rounds of workgroups (tail), never below one tile per split
            int splits = 1;
            while (panels * C * P * splits < rounds * 256 && splits * 2 <= n_tiles) splits *= 2;
            pl.tiles_per_split = (n_tiles + splits - 1) / splits;
            pl.splits = (n_tiles + pl.tiles_per_split - 1) / pl.tiles_per_split;
            return pl;
        }
    }
#endif
    if (D == 320) {
#ifndef TF_TUNE_NN_NO_RB2
        // two target tiles per wave where the launch still has >= 2 rounds of its 512 resident workgroups (2 per CU) AND a
        // workgroup streams >= 24 pivot tiles: its 160 VGPRs of target fragments are a fixed cost per workgroup (one chunk of
        // cfg5, 32 tiles per workgroup: 196 -> 163 us; one chunk of cfg2, 16 tiles: 123 -> 125 us, profiles/r06_nn_rb2_ab.txt)
        static const int rb2_min = [] { const char* e = getenv("TF_NN_RB2_MIN_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
        static const bool dma_ok = [] { const char* e = getenv("TF_NN_RB_GLDS"); return !e || atoi(e) != 0; }();
        if (dma_ok && S % 32 == 0 && shape(256, 32) >= rb2_min && pl.tiles_per_split >= 24) {
            pl.kern = NN_RB2;
            return pl;
        }
#endif
        pl.kern = NN_RB;
        shape(128, 32);
        return pl;
    }
    // 128-target panels halve the pivot re-reads; they pay as soon as the grid still fills the GPU after
    // splitting the pivot range (measured at cfg2 level 1, 5120 targets x 2 keyframes: 34.6 vs 39.8 us)
    if (((n_tgt + 127) / 128) * P * C >= 64) {
        pl.kern = NN_WIDE;
        shape(128, 128);
        return pl;
    }
    // few workgroups and a long contraction: the loop is a chain of load -> LDS -> barrier -> MFMA steps whose
    // latency nothing hides -> wider D chunks (fewer steps)
#ifndef TF_TUNE_NN_NO_DEEP   // A/B switch of tools/build_variants.sh
    if (D >= 1024 && shape(64, 64) <= 512) {
        pl.kern = NN_DEEP;
        return pl;
    }
#endif
    pl.kern = (shape(64, 128) <= 512 && D >= 512) ? NN_BK128 : NN_BK64;
    return pl;
}

static int finalize(const NnPartial* part, int32_t* idx, int64_t total, int splits, hipStream_t st) {
    hipLaunchKernelGGL(nn_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part, idx, total,
                       splits);
    TF_LAUNCH_CHECK("tf_nn_search(finalize)");
    return 0;
}

// n_tgt = targets per chunk; C chunks per launch (C = 1, first_single = 0: the plain single-chunk search).
// Partial results / indices are laid out over all C * n_tgt targets.
template <typename T, int WN, int BK, int TM>
int launch_nn(const void* tgt, const void* piv, const float* inv_norm, int32_t* idx, NnPartial* ws, int64_t n_tgt,
              int S, int D, int P, int kf0, int kf1, hipStream_t st, bool fin, int C, int first_single) {
    constexpr int TN = 64 * WN;
    constexpr size_t lds = 2 * (TM + TN) * BK * 2 + 2 * TM * 4 + 2 * TN * 8;
    static_assert(lds <= 160 * 1024, "LDS");
    const NnPlan pl = nn_plan(n_tgt, S, D, P, C);
    const int splits = pl.splits, tps = pl.tiles_per_split;
    dim3 grid((unsigned)(pl.panels * C), (unsigned)P, (unsigned)splits);
    auto kern = nn_search_kernel<T, WN, BK, TM>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const NnChunks ch{n_tgt, (int)pl.panels, first_single};
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, reinterpret_cast<const typename T::elem*>(tgt),
                       reinterpret_cast<const typename T::elem*>(piv), inv_norm, idx,
                       (splits > 1 || !fin) ? ws : nullptr, n_tgt * C, S, D, kf0, kf1, tps, ch);
    TF_LAUNCH_CHECK("tf_nn_search");
    return (fin && splits > 1) ? finalize(ws, idx, n_tgt * C * P, splits, st) : 0;
}

template <typename T>
int launch_nn_glds(const void* tgt, const void* piv, const float* inv_norm, int32_t* idx, NnPartial* ws, int64_t n_tgt,
                   int S, int D, int P, int kf0, int kf1, hipStream_t st, bool fin, int C, int first_single) {
    constexpr size_t lds = 2 * (256 + 256) * 64 * 2 + 2 * 256 * 4 + 4 * 256 * 8;
    static_assert(lds <= 160 * 1024, "LDS");
    const NnPlan pl = nn_plan(n_tgt, S, D, P, C);
    const int splits = pl.splits, tps = pl.tiles_per_split;
    dim3 grid((unsigned)(pl.panels * C), (unsigned)P, (unsigned)splits);
#ifndef TF_TUNE_NN_NO_GLDS_SH
    auto kern = nn_search_glds_kernel<T, true>;    // short MFMAs (round 6, last session)
#else
    auto kern = nn_search_glds_kernel<T, false>;
#endif
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const NnChunks ch{n_tgt, (int)pl.panels, first_single};
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, reinterpret_cast<const typename T::elem*>(tgt),
                       reinterpret_cast<const typename T::elem*>(piv), inv_norm, idx,
                       (splits > 1 || !fin) ? ws : nullptr, n_tgt * C, S, D, kf0, kf1, tps, ch);
    TF_LAUNCH_CHECK("tf_nn_search");
    return (fin && splits > 1) ? finalize(ws, idx, n_tgt * C * P, splits, st) : 0;
}

template <typename T, int DK>
int launch_nn_rb(const void* tgt, const void* piv, const float* inv_norm, int32_t* idx, NnPartial* ws, int64_t n_tgt,
                 int S, int P, int kf0, int kf1, hipStream_t st, bool fin, int C, int first_single) {
    constexpr int D = 16 * DK;
    const size_t lds = 2 * 32 * (D + 8) * 2 + 2 * 32 * 4;
    const NnPlan pl = nn_plan(n_tgt, S, D, P, C);
    const int splits = pl.splits, tps = pl.tiles_per_split;
    dim3 grid((unsigned)(pl.panels * C), (unsigned)P, (unsigned)splits);
    const NnChunks ch{n_tgt, (int)pl.panels, first_single};
    if (pl.kern == NN_RB2) {   // 256-target panels, two target tiles per wave (the plan has checked S % 32 == 0)
        const size_t lds_g = 2 * 32 * D * 2 + 2 * 32 * 4;
#ifndef TF_TUNE_NN_NO_RBS
        // round 6, last session: the same kernel with short MFMAs (16x16x32), see nn_search_rbs_kernel
        hipLaunchKernelGGL((nn_search_rbs_kernel<T, DK>), grid, dim3(256), lds_g, st,
                           reinterpret_cast<const typename T::elem*>(tgt), reinterpret_cast<const typename T::elem*>(piv),
                           inv_norm, idx, (splits > 1 || !fin) ? ws : nullptr, n_tgt * C, S, kf0, kf1, tps, ch);
        TF_LAUNCH_CHECK("tf_nn_search");
        return (fin && splits > 1) ? finalize(ws, idx, n_tgt * C * P, splits, st) : 0;
#endif
        hipLaunchKernelGGL((nn_search_rbg_kernel<T, DK, 2>), grid, dim3(256), lds_g, st,
                           reinterpret_cast<const typename T::elem*>(tgt), reinterpret_cast<const typename T::elem*>(piv),
                           inv_norm, idx, (splits > 1 || !fin) ? ws : nullptr, n_tgt * C, S, kf0, kf1, tps, ch);
        TF_LAUNCH_CHECK("tf_nn_search");
        return (fin && splits > 1) ? finalize(ws, idx, n_tgt * C * P, splits, st) : 0;
    }
    // pivot tiles by LDS-DMA where every tile is full (S % 32 == 0): +1.4 % at cfg2 / cfg4 level 0
    // (profiles/r05_nn_glds_ab.txt: 1028 -> 1014 us, 8826 -> 8746 us per block); TF_NN_RB_GLDS=0: the register-staged kernel
    static const bool dma = [] { const char* e = getenv("TF_NN_RB_GLDS"); return !e || atoi(e) != 0; }();
    if (dma && S % 32 == 0) {
        const size_t lds_g = 2 * 32 * D * 2 + 2 * 32 * 4;
#ifndef TF_TUNE_NN_NO_RBS
        hipLaunchKernelGGL((nn_search_rbs_kernel<T, DK, 2>), grid, dim3(256), lds_g, st,
                           reinterpret_cast<const typename T::elem*>(tgt), reinterpret_cast<const typename T::elem*>(piv),
                           inv_norm, idx, (splits > 1 || !fin) ? ws : nullptr, n_tgt * C, S, kf0, kf1, tps, ch);
        TF_LAUNCH_CHECK("tf_nn_search");
        return (fin && splits > 1) ? finalize(ws, idx, n_tgt * C * P, splits, st) : 0;
#endif
        hipLaunchKernelGGL((nn_search_rbg_kernel<T, DK>), grid, dim3(256), lds_g, st,
                           reinterpret_cast<const typename T::elem*>(tgt), reinterpret_cast<const typename T::elem*>(piv),
                           inv_norm, idx, (splits > 1 || !fin) ? ws : nullptr, n_tgt * C, S, kf0, kf1, tps, ch);
        TF_LAUNCH_CHECK("tf_nn_search");
        return (fin && splits > 1) ? finalize(ws, idx, n_tgt * C * P, splits, st) : 0;
    }
    hipLaunchKernelGGL((nn_search_rb_kernel<T, DK>), grid, dim3(256), lds, st,
                       reinterpret_cast<const typename T::elem*>(tgt), reinterpret_cast<const typename T::elem*>(piv),
                       inv_norm, idx, (splits > 1 || !fin) ? ws : nullptr, n_tgt * C, S, kf0, kf1, tps, ch);
    TF_LAUNCH_CHECK("tf_nn_search");
    return (fin && splits > 1) ? finalize(ws, idx, n_tgt * C * P, splits, st) : 0;
}

template <typename T>
int dispatch_nn(const void* tgt, const void* piv, const float* inv_norm, int32_t* idx, NnPartial* ws, int64_t n_tgt,
                int S, int D, int P, int kf0, int kf1, hipStream_t st, bool fin, int C = 1, int first_single = 0) {
    switch (nn_plan(n_tgt, S, D, P, C).kern) {
        case NN_RB:
        case NN_RB2:
            return launch_nn_rb<T, 20>(tgt, piv, inv_norm, idx, ws, n_tgt, S, P, kf0, kf1, st, fin, C, first_single);
        case NN_GLDS:
            return launch_nn_glds<T>(tgt, piv, inv_norm, idx, ws, n_tgt, S, D, P, kf0, kf1, st, fin, C, first_single);
        case NN_WIDE:
            return launch_nn<T, 2, 64, 128>(tgt, piv, inv_norm, idx, ws, n_tgt, S, D, P, kf0, kf1, st, fin, C, first_single);
        case NN_DEEP:
            return launch_nn<T, 1, 256, 64>(tgt, piv, inv_norm, idx, ws, n_tgt, S, D, P, kf0, kf1, st, fin, C, first_single);
        case NN_BK128:
            return launch_nn<T, 1, 128, 128>(tgt, piv, inv_norm, idx, ws, n_tgt, S, D, P, kf0, kf1, st, fin, C, first_single);
        default:
            return launch_nn<T, 1, 64, 128>(tgt, piv, inv_norm, idx, ws, n_tgt, S, D, P, kf0, kf1, st, fin, C, first_single);
    }
}

}  // namespace

extern "C" int tf_pivot_inv_norm(const void* piv, float* inv_norm, int64_t rows, int D, int dtype, void* stream) {
    TF_ARG(piv && inv_norm, TF_ERR_NULL, "tf_pivot_inv_norm: null pointer");
    TF_ARG(dtype == TF_BF16 || dtype == TF_F16, TF_ERR_DTYPE, "tf_pivot_inv_norm: dtype %d (bf16/f16 only)", dtype);
    TF_ARG(rows > 0 && D > 0 && D % 8 == 0, TF_ERR_SHAPE, "tf_pivot_inv_norm: rows=%lld D=%d (D %% 8 == 0)",
           (long long)rows, D);
    TF_ARG(tf_aligned16(piv), TF_ERR_ALIGN, "tf_pivot_inv_norm: piv not 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
    if (dtype == TF_BF16)
        hipLaunchKernelGGL(pivot_inv_norm_kernel<BF16>, dim3(grid), dim3(256), 0, st,
                           reinterpret_cast<const __bf16*>(piv), inv_norm, rows, D);
    else
        hipLaunchKernelGGL(pivot_inv_norm_kernel<F16>, dim3(grid), dim3(256), 0, st,
                           reinterpret_cast<const _Float16*>(piv), inv_norm, rows, D);
    TF_LAUNCH_CHECK("tf_pivot_inv_norm");
    return 0;
}

extern "C" size_t tf_nn_search_workspace_bytes(int64_t n_tgt, int S, int D, int P) {
    if (n_tgt <= 0 || S <= 0 || D <= 0 || P <= 0) return 0;
    const NnPlan pl = nn_plan(n_tgt, S, D, P);
    const size_t bytes = pl.splits > 1 ? (size_t)pl.splits * P * n_tgt * sizeof(NnPartial) : 0;
    return bytes < 256 ? 256 : bytes;   // never 0: the caller always passes a valid pointer
}

extern "C" int tf_nn_search(const void* tgt, const void* piv, const float* inv_norm, int32_t* idx, int64_t n_tgt,
                            int S, int D, int P, int kf0, int kf1, int dtype, void* ws, size_t ws_bytes,
                            void* stream) {
    TF_ARG(tgt && piv && inv_norm && idx && ws, TF_ERR_NULL, "tf_nn_search: null pointer");
    TF_ARG(dtype == TF_BF16 || dtype == TF_F16, TF_ERR_DTYPE, "tf_nn_search: dtype %d (bf16/f16 only)", dtype);
    TF_ARG(n_tgt > 0 && S > 0 && D > 0 && D % 8 == 0 && (P == 1 || P == 2) && kf0 >= 0 && (P == 1 || kf1 >= 0),
           TF_ERR_SHAPE, "tf_nn_search: n_tgt=%lld S=%d D=%d P=%d kf=(%d,%d)", (long long)n_tgt, S, D, P, kf0, kf1);
    TF_ARG(tf_aligned16(tgt) && tf_aligned16(piv) && tf_aligned16(ws) && tf_aligned16(inv_norm), TF_ERR_ALIGN,
           "tf_nn_search: inputs not 16-byte aligned (inv_norm included: the LDS-DMA kernel fetches it in 16-byte pieces)");
    TF_ARG(ws_bytes >= tf_nn_search_workspace_bytes(n_tgt, S, D, P), TF_ERR_WORKSPACE,
           "tf_nn_search: workspace %zu < %zu bytes", ws_bytes, tf_nn_search_workspace_bytes(n_tgt, S, D, P));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    NnPartial* part = reinterpret_cast<NnPartial*>(ws);
    return dtype == TF_BF16 ? dispatch_nn<BF16>(tgt, piv, inv_norm, idx, part, n_tgt, S, D, P, kf0, kf1, st, true)
                            : dispatch_nn<F16>(tgt, piv, inv_norm, idx, part, n_tgt, S, D, P, kf0, kf1, st, true);
}

int tf_nn_search_partials(const void* tgt, const void* piv, const float* inv_norm, NnPartial* part, int64_t n_tgt,
                          int S, int D, int P, int kf0, int kf1, int dtype, hipStream_t st, int* splits, int C,
                          int first_single) {
    *splits = nn_plan(n_tgt, S, D, P, C).splits;
    return dtype == TF_BF16
               ? dispatch_nn<BF16>(tgt, piv, inv_norm, nullptr, part, n_tgt, S, D, P, kf0, kf1, st, false, C, first_single)
               : dispatch_nn<F16>(tgt, piv, inv_norm, nullptr, part, n_tgt, S, D, P, kf0, kf1, st, false, C, first_single);
}

size_t tf_nn_partials_bytes(int64_t n_tgt, int S, int D, int P, int C) {
    return (size_t)nn_plan(n_tgt, S, D, P, C).splits * P * n_tgt * C * sizeof(NnPartial);
}
