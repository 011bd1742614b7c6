// One rank's pivotal pass of a transformer block in a frame-sharded multi-GPU run, issued by ONE host call.
//
// The reference is single-process; tokenflow_amd/sharded.py describes the partitioning
// (tokenflow_utils.py:133-138: every keyframe's queries read the keys / values of ALL K keyframes; 331-333: chunk c
// reads keyframes c and c-1) and is the Python form of the same sequence.  This file is that sequence as native
// code: the per-block work of a rank at 8 GPUs is a dozen launches of 5..500 us with three exchanges between them,
// and a Python host spends 200-300 us per block issuing them (profiles/r03_rank_step_v2.txt) -- more than the GPU
// needs for the block at three of the four UNet levels.  Here the host cost is one foreign call.
//
// Streams: the caller's stream carries the compute AND every collective of the pivotal pass's communicator (one
// communicator, one stream: no reliance on cross-stream ordering inside RCCL; a hand-over between two streams on the
// critical path also costs ~10 us of idle device each way, profiles/r04_rank_timeline_v1.txt).  Two side streams:
// an auxiliary COMPUTE stream runs the source branch of the local frames beside the first exchange and the bank
// attention where those are separate launches (level 0; forked behind the pack -- in FRONT of the exchange -- and joined
// in front of the second exchange by events, nothing waits on it before the join), and the neighbour halo has a stream of its own (with its own communicator when the host gives one:
// collectives of ONE RCCL communicator execute in issue order, and a 10 MB halo message in front of the next block's
// all-to-all would sit on the critical path); it has the rest of the pass to arrive.  All ordering is by events; no
// host synchronisation anywhere.
#include <stdlib.h>

#include <new>

#include "attn_fused.h"
#include "tf_common.h"

struct tf_comm;   // csrc/comm.hip

namespace {

constexpr int RING = 64;

int hip_fail(const char* what, hipError_t e) {
    tf_set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
}
#define TF_HIP(call, what)                                   \
    do {                                                     \
        const hipError_t e_ = (call);                        \
        if (e_ != hipSuccess) return hip_fail(what, e_);     \
    } while (0)

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

struct tf_rank {
    tf_comm* comm;
    tf_comm* halo_comm;
    int K, world, rank, Kl, kf0;
    int counts[TF_MAX_WORLD];
    hipStream_t hs = nullptr;   // neighbour halo
    hipStream_t as = nullptr;   // auxiliary COMPUTE stream: the source branch beside the exchange and the bank launch
    hipEvent_t ring[RING];
    int ring_i = 0;
    hipEvent_t halo_done[TF_RANK_SLOTS];
    bool halo_set[TF_RANK_SLOTS];

    hipEvent_t next() {
        hipEvent_t e = ring[ring_i];
        ring_i = (ring_i + 1) % RING;
        return e;
    }
};

namespace {

// `to` continues after everything enqueued on `from` so far.  Events are reused: a wait captures the record that is
// current when it is issued (HIP semantics), and 64 events outlast any hand-over of a block.
int order(tf_rank* rk, hipStream_t from, hipStream_t to, const char* what) {
    hipEvent_t e = rk->next();
    TF_HIP(hipEventRecord(e, from), what);
    TF_HIP(hipStreamWaitEvent(to, e, 0), what);
    return 0;
}

struct Layout {   // exchange buffers of one call inside the caller's workspace
    size_t send, recv, send2, recv2, ws_bank, ws_src, total;
    size_t ws_bank_bytes, ws_src_bytes;
};

Layout layout(const tf_rank* rk, int S, int H, int Dh, int dtype, int mode) {
    const size_t eb = 2;
    const int W = rk->world, Kl = rk->Kl, K = rk->K;
    const size_t D = (size_t)H * Dh;
    Layout L{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += up256(bytes);
        return at;
    };
    if (mode == TF_RANK_HEADS) {
        const size_t hd = D / W;
        L.send = take((size_t)W * Kl * 6 * S * hd * eb);
        L.recv = take((size_t)K * 6 * S * hd * eb);
        L.send2 = take((size_t)K * 2 * S * hd * eb);
        L.recv2 = take((size_t)W * Kl * 2 * S * hd * eb);
        L.ws_bank_bytes = tf_ext_attn_workspace_bytes(K, S, H / W, Dh, dtype);
        L.ws_src_bytes = tf_ext_attn_workspace_bytes(Kl, S, H, Dh, dtype);
        L.ws_bank = take(L.ws_bank_bytes);
        L.ws_src = take(L.ws_src_bytes);
    } else {
        L.send = take((size_t)Kl * 6 * S * D * eb);
        L.recv = take((size_t)K * 6 * S * D * eb);
        L.ws_bank_bytes = tf_ext_attn_workspace_bytes(K, S, H, Dh, dtype);
        L.ws_bank = take(L.ws_bank_bytes);
    }
    L.total = off;
    return L;
}

}  // namespace

extern "C" int tf_rank_create(tf_comm* comm, tf_comm* halo_comm, int K, tf_rank** out) {
    TF_ARG(out, TF_ERR_NULL, "tf_rank_create: null pointer");
    const int world = comm ? tf_comm_world(comm) : 1, rank = comm ? tf_comm_rank(comm) : 0;
    TF_ARG(world >= 1 && world <= TF_MAX_WORLD && K >= world, TF_ERR_SHAPE,
           "tf_rank_create: %d keyframes over %d ranks (every rank owns at least one; at most %d ranks)", K, world,
           TF_MAX_WORLD);
    TF_ARG(!halo_comm || (tf_comm_world(halo_comm) == world && tf_comm_rank(halo_comm) == rank), TF_ERR_SHAPE,
           "tf_rank_create: the halo communicator must have the same rank and world");
    tf_rank* rk = new (std::nothrow) tf_rank();
    TF_ARG(rk, TF_ERR_NULL, "tf_rank_create: out of memory");
    rk->comm = comm;
    rk->halo_comm = halo_comm ? halo_comm : comm;
    rk->K = K, rk->world = world, rk->rank = rank;
    int off = 0;
    for (int r = 0; r < world; ++r) {   // contiguous runs, the first K % W ranks hold one more (sharded.py)
        rk->counts[r] = K / world + (r < K % world ? 1 : 0);
        if (r == rank) rk->kf0 = off;
        off += rk->counts[r];
    }
    rk->Kl = rk->counts[rank];
    for (int i = 0; i < TF_RANK_SLOTS; ++i) rk->halo_set[i] = false, rk->halo_done[i] = nullptr;
    for (int i = 0; i < RING; ++i) rk->ring[i] = nullptr;
    hipError_t e = hipSuccess;
    for (int i = 0; i < RING && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&rk->ring[i], hipEventDisableTiming);
    for (int i = 0; i < TF_RANK_SLOTS && e == hipSuccess; ++i)
        e = hipEventCreateWithFlags(&rk->halo_done[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&rk->hs, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&rk->as, hipStreamNonBlocking);
    if (e != hipSuccess) {
        tf_rank_destroy(rk);
        return hip_fail("tf_rank_create", e);
    }
    *out = rk;
    return 0;
}

extern "C" int tf_rank_destroy(tf_rank* rk) {
    if (!rk) return 0;
    for (hipStream_t s : {rk->hs, rk->as})
        if (s) {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    for (int i = 0; i < RING; ++i)
        if (rk->ring[i]) (void)hipEventDestroy(rk->ring[i]);
    for (int i = 0; i < TF_RANK_SLOTS; ++i)
        if (rk->halo_done[i]) (void)hipEventDestroy(rk->halo_done[i]);
    delete rk;
    return 0;
}

extern "C" int tf_rank_local_keyframes(const tf_rank* rk) { return rk ? rk->Kl : 0; }
extern "C" int tf_rank_first_keyframe(const tf_rank* rk) { return rk ? rk->kf0 : 0; }

extern "C" size_t tf_rank_pivotal_workspace_bytes(const tf_rank* rk, int S, int H, int Dh, int dtype) {
    if (!rk || S <= 0 || H <= 0 || Dh <= 0 || dtype == TF_F32) return 0;
    if (rk->world == 1) return up256(tf_ext_attn_workspace_bytes(rk->K, S, H, Dh, dtype));
    const size_t bank = layout(rk, S, H, Dh, dtype, TF_RANK_BANK).total;
    const size_t heads = H % rk->world == 0 ? layout(rk, S, H, Dh, dtype, TF_RANK_HEADS).total : 0;
    return bank > heads ? bank : heads;
}

extern "C" int tf_rank_pivotal(tf_rank* rk, const void* q, const void* k, const void* v, const int64_t* st_in,
                               void* piv_ext, float* inv_ext, void* kfo_ext, int S, int H, int Dh, float scale,
                               int flags, int dtype, int mode, int slot, void* ws, size_t ws_bytes, void* stream) {
    const bool no_halo = (mode & TF_RANK_NO_HALO) != 0;   // attention only: kfo_ext is a plain [3, Kl, S, H*Dh] output
    const bool want_inv = (mode & TF_RANK_INV_NORM) != 0; // the call computes the local pivots' inverse norms (in its pack launch)
    mode &= ~(TF_RANK_NO_HALO | TF_RANK_INV_NORM);
    TF_ARG(rk && q && k && v && st_in && kfo_ext && ws && (no_halo || (piv_ext && inv_ext)), TF_ERR_NULL,
           "tf_rank_pivotal: null pointer");
    TF_ARG(dtype == TF_BF16 || dtype == TF_F16, TF_ERR_DTYPE, "tf_rank_pivotal: dtype %d (bf16/f16 only)", dtype);
    TF_ARG(mode == TF_RANK_HEADS || mode == TF_RANK_BANK, TF_ERR_SHAPE, "tf_rank_pivotal: mode %d", mode);
    TF_ARG(slot >= 0 && slot < TF_RANK_SLOTS, TF_ERR_SHAPE, "tf_rank_pivotal: slot %d outside [0, %d)", slot, TF_RANK_SLOTS);
    TF_ARG(!(want_inv && no_halo), TF_ERR_SHAPE, "tf_rank_pivotal: TF_RANK_INV_NORM needs the propagation state (no TF_RANK_NO_HALO)");
    TF_ARG(!(flags & (TF_ATTN_BANK_ONLY | TF_ATTN_SOURCE_ONLY)), TF_ERR_SHAPE,
           "tf_rank_pivotal: the part flags are the executor's own");
    TF_ARG(ws_bytes >= tf_rank_pivotal_workspace_bytes(rk, S, H, Dh, dtype), TF_ERR_WORKSPACE,
           "tf_rank_pivotal: workspace %zu < %zu bytes", ws_bytes, tf_rank_pivotal_workspace_bytes(rk, S, H, Dh, dtype));
    const int W = rk->world, Kl = rk->Kl, K = rk->K, o = (W > 1 && !no_halo) ? 1 : 0;
    const int64_t D = (int64_t)H * Dh, SD = (int64_t)S * D;
    const int64_t q_bs = st_in[0], q_fs = st_in[1], k_bs = st_in[2], k_fs = st_in[3], v_bs = st_in[4], v_fs = st_in[5],
                  ld_q = st_in[6], ld = st_in[7];
    typedef unsigned short E;   // any 16-bit element: only pointer arithmetic happens here
    const E* qe = static_cast<const E*>(q);
    const E* ke = static_cast<const E*>(k);
    const E* ve = static_cast<const E*>(v);
    E* kfo = static_cast<E*>(kfo_ext);
    E* piv = static_cast<E*>(piv_ext);
    unsigned char* wsb = static_cast<unsigned char*>(ws);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t o_bs = (int64_t)(Kl + o) * SD;   // branch stride of the halo-extended attention output
    E* out_loc = kfo + (int64_t)o * SD;            // its local slots
    const int inject = flags & TF_ATTN_INJECT;
    if (!no_halo) rk->halo_set[slot] = false;

    // the local keyframes' pivots and inverse norms (slots o.. of the halo-extended state)
    const E* piv_loc = want_inv ? piv + (int64_t)o * SD : nullptr;
    float* inv_loc = want_inv ? inv_ext + (int64_t)o * S : nullptr;
    if (W == 1) {
        if (want_inv)
            if (const int rc = tf_pivot_inv_norm(piv_loc, inv_loc, (int64_t)Kl * S, (int)D, dtype, stream)) return rc;
        const int64_t strides[9] = {q_bs, q_fs, k_bs, k_fs, v_bs, v_fs, o_bs, SD, ld_q};
        return tf_ext_attn_fwd_strided(q, k, v, out_loc, K, K, 0, S, H, Dh, ld, strides, scale, flags, dtype, ws, ws_bytes,
                                       stream);
    }
    int64_t cnt[TF_MAX_WORLD], own[TF_MAX_WORLD];
    for (int p = 0; p < W; ++p) cnt[p] = rk->counts[p], own[p] = Kl;

    if (mode == TF_RANK_HEADS) {
        TF_ARG(H % W == 0, TF_ERR_SHAPE, "tf_rank_pivotal: %d heads do not divide over %d ranks (use TF_RANK_BANK)", H, W);
        TF_ARG(ld_q == ld, TF_ERR_SHAPE, "tf_rank_pivotal: the head re-sharding packs q, k, v with one token stride");
        const Layout L = layout(rk, S, H, Dh, dtype, TF_RANK_HEADS);
        const int Hl = H / W;
        const int64_t hd = D / W, Shd = (int64_t)S * hd;
        E* send = reinterpret_cast<E*>(wsb + L.send);
        E* recv = reinterpret_cast<E*>(wsb + L.recv);
        E* send2 = reinterpret_cast<E*>(wsb + L.send2);
        E* recv2 = reinterpret_cast<E*>(wsb + L.recv2);
        // ---- pack: head group w of every slab the bank branches read, frame-major (slab order [q.., k.., v..])
        const void* slabs[6];
        int64_t fss[6];
        int ns;
        if (inject) {   // source q, k (what uncond and cond use, tokenflow_utils.py:124-130) and the two value banks
            ns = 4;
            slabs[0] = qe, slabs[1] = ke, slabs[2] = ve + v_bs, slabs[3] = ve + 2 * v_bs;
            fss[0] = q_fs, fss[1] = k_fs, fss[2] = v_fs, fss[3] = v_fs;
        } else {
            ns = 6;
            slabs[0] = qe + q_bs, slabs[1] = qe + 2 * q_bs, slabs[2] = ke + k_bs, slabs[3] = ke + 2 * k_bs;
            slabs[4] = ve + v_bs, slabs[5] = ve + 2 * v_bs;
            fss[0] = fss[1] = q_fs, fss[2] = fss[3] = k_fs, fss[4] = fss[5] = v_fs;
        }
        if (const int rc = tf_head_pack_norm(slabs, fss, ns, send, W, Kl, S, (int)hd, ld, 2, piv_loc, inv_loc,
                                             (int64_t)Kl * S, (int)D, dtype, stream))
            return rc;
        // ---- the two tensor sets of the attention: the bank branches on this rank's head group over all K frames, read
        //      from `recv` and written to `send2` in place (both laid out for the collectives), and the source branch of
        //      the local frames on the local projections.
        const int64_t fs_r = ns * Shd;          // frame stride of recv [K][ns][S][hd]
        // base such that branch b sits at base + b * Shd (the strided entry point's convention): a bank-only call
        // never touches branch 0, so the base may lie one slab in front of the buffer -- formed as an integer
        auto slab = [&](const E* buf, int64_t first_slab, int first_branch) {
            return reinterpret_cast<const E*>(reinterpret_cast<uintptr_t>(buf) +
                                              (uintptr_t)((first_slab - first_branch) * Shd * (int64_t)sizeof(E)));
        };
        const E *qb, *kb, *vb;
        if (inject)    // slabs [q0, k0, v1, v2]
            qb = slab(recv, 0, 0), kb = slab(recv, 1, 0), vb = slab(recv, 2, 1);
        else           // slabs [q1, q2, k1, k2, v1, v2]
            qb = slab(recv, 0, 1), kb = slab(recv, 2, 1), vb = slab(recv, 4, 1);
        E* ob = const_cast<E*>(slab(send2, 0, 1));                           // send2 [K][uncond|cond][S][hd]
        TfAttnSet sets[2] = {};
        sets[0].q = qb, sets[0].k = kb, sets[0].v = vb, sets[0].out = ob;
        sets[0].q_bs = Shd, sets[0].q_fs = fs_r, sets[0].ld_q = hd, sets[0].k_bs = Shd, sets[0].k_fs = fs_r;
        sets[0].v_bs = Shd, sets[0].v_fs = fs_r, sets[0].ld = hd, sets[0].o_bs = Shd, sets[0].o_fs = 2 * Shd;
        sets[0].H = Hl, sets[0].Kq = K, sets[0].q_frame0 = 0, sets[0].Kb = K, sets[0].b0 = 1, sets[0].nb = 2;
        sets[1].q = q, sets[1].k = k, sets[1].v = v, sets[1].out = out_loc;
        sets[1].q_bs = q_bs, sets[1].q_fs = q_fs, sets[1].ld_q = ld_q, sets[1].k_bs = k_bs, sets[1].k_fs = k_fs;
        sets[1].v_bs = v_bs, sets[1].v_fs = v_fs, sets[1].ld = ld, sets[1].o_bs = o_bs, sets[1].o_fs = SD;
        sets[1].H = H, sets[1].Kq = Kl, sets[1].q_frame0 = 0, sets[1].Kb = Kl, sets[1].b0 = 0, sets[1].nb = 1;
        // small problems (the coarse levels, a rank's share of the middle ones): ONE launch for both sets behind the
        // exchange -- no V^T pre-passes, no split + merge pair, no separate source launch (csrc/ext_attn_fused.hip)
        // The plan (which kernel, which key split: it decides the arithmetic) must be the SAME on every rank of the
        // block: derived from rank-independent quantities -- the largest local run ceil(K / W) stands for Kl (with an
        // uneven K the ranks' own Kl differ by one).
        TfAttnSet plan_sets[2] = {sets[0], sets[1]};
        plan_sets[1].Kq = plan_sets[1].Kb = (K + W - 1) / W;
        const TfFusedPlan plan = tf_attn_fused_plan(plan_sets, 2, S, Dh, dtype, flags);
        // Everything on the caller's stream, collectives included: every collective of the communicator is issued on
        // ONE stream (no reliance on RCCL's ordering of one communicator across streams), and a hand-over between two
        // streams costs ~10 us of idle device each way (profiles/r04_rank_timeline_v1.txt: four of them per block were
        // 45 us of a 120 us block at the coarse levels).
        // Where the source branch of the local frames is a launch of its own (level 0: the fused plan does not apply), it
        // runs on the auxiliary COMPUTE stream, forked HERE -- behind the pack, in front of the first exchange: it reads
        // only the caller's q / k / v, so it runs under the exchange's wire time and then beside the bank launch, which
        // absorbs what is left of it (a rank's own frames are too few workgroups to fill the chip: cfg2 level 0 at W = 8,
        // 128 workgroups, 61 us alone; 471 -> 440 us per level-0 block with the wire taken out,
        // profiles/r05_rank_step_srcaux_ab.txt).  Joined in front of the second exchange.  The collectives stay on the
        // caller's stream.  TOKENFLOW_RANK_SRC_AUX=0: in line on the caller's stream, behind the exchange.
        static const bool src_aux = [] { const char* e = getenv("TOKENFLOW_RANK_SRC_AUX"); return !e || atoi(e) != 0; }();
        const int64_t src_strides[9] = {q_bs, q_fs, k_bs, k_fs, v_bs, v_fs, o_bs, SD, ld_q};
        auto source_branch = [&](hipStream_t on) {
            return tf_ext_attn_fwd_strided(q, k, v, out_loc, Kl, Kl, 0, S, H, Dh, ld, src_strides, scale,
                                           flags | TF_ATTN_SOURCE_ONLY, dtype, wsb + L.ws_src, L.ws_src_bytes, on);
        };
        const bool fork = !plan.use && src_aux;
        if (fork) {
            if (const int rc = order(rk, st, rk->as, "tf_rank_pivotal")) return rc;
            if (const int rc = source_branch(rk->as)) return rc;
        }
        if (const int rc = tf_all_to_all_rows(rk->comm, send, recv, own, cnt, ns * Shd, dtype, st)) return rc;
        if (plan.use) {
            // small problems (the coarse levels, a rank's share of the middle ones): ONE launch for both sets behind
            // the exchange -- no V^T pre-passes, no split + merge pair, no separate source launch
            if (const int rc = tf_attn_fused_launch(sets, 2, S, Dh, scale, flags, dtype, plan, st)) return rc;
        } else {
            if (!fork)
                if (const int rc = source_branch(st)) return rc;
            const int64_t strides[9] = {Shd, fs_r, Shd, fs_r, Shd, fs_r, Shd, 2 * Shd, hd};
            if (const int rc = tf_ext_attn_fwd_strided(qb, kb, vb, ob, K, K, 0, S, Hl, Dh, hd, strides, scale,
                                                       flags | TF_ATTN_BANK_ONLY, dtype,
                                                       wsb + L.ws_bank, L.ws_bank_bytes, stream))
                return rc;
            if (fork)
                if (const int rc = order(rk, rk->as, st, "tf_rank_pivotal")) return rc;
        }
        // ---- outputs back to the frame owners: on the caller's stream (nothing can run beside this exchange: the
        //      unpack and the next block need its result)
        if (const int rc = tf_all_to_all_rows(rk->comm, send2, recv2, cnt, own, 2 * Shd, dtype, st)) return rc;
        void* dsts[2] = {out_loc + o_bs, out_loc + 2 * o_bs};
        const int64_t dfs[2] = {SD, SD};
        if (const int rc = tf_head_unpack(recv2, dsts, dfs, 2, W, Kl, S, (int)hd, D, 2, stream)) return rc;
    } else {
        // ---- ONE collective: the slabs the attention reads across frames, gathered into [K][slabs][S][D]
        const Layout L = layout(rk, S, H, Dh, dtype, TF_RANK_BANK);
        E* send = reinterpret_cast<E*>(wsb + L.send);
        E* recv = reinterpret_cast<E*>(wsb + L.recv);
        const void* slabs[6];
        int64_t fss[6];
        int ns;
        if (inject) {
            ns = 4;
            slabs[0] = ke, slabs[1] = ve, slabs[2] = ve + v_bs, slabs[3] = ve + 2 * v_bs;
            fss[0] = k_fs, fss[1] = fss[2] = fss[3] = v_fs;
        } else {
            ns = 6;
            slabs[0] = ke, slabs[1] = ke + k_bs, slabs[2] = ke + 2 * k_bs;
            slabs[3] = ve, slabs[4] = ve + v_bs, slabs[5] = ve + 2 * v_bs;
            fss[0] = fss[1] = fss[2] = k_fs, fss[3] = fss[4] = fss[5] = v_fs;
        }
        if (const int rc = tf_head_pack_norm(slabs, fss, ns, send, 1, Kl, S, (int)D, ld, 2, piv_loc, inv_loc,
                                             (int64_t)Kl * S, (int)D, dtype, stream))
            return rc;
        // the gather on the caller's stream: the attention needs it at once (no stream hand-over, see above)
        if (const int rc = tf_allgather_rows(rk->comm, send, recv, cnt, ns * SD, dtype, st)) return rc;
        const int64_t fs_r = ns * SD;
        const E* kb = recv;
        const E* vb = recv + (inject ? 1 : 3) * SD;
        const int64_t strides[9] = {q_bs, q_fs, SD, fs_r, SD, fs_r, o_bs, SD, ld_q};
        if (const int rc = tf_ext_attn_fwd_strided(q, kb, vb, out_loc, K, Kl, rk->kf0, S, H, Dh, D, strides, scale, flags,
                                                   dtype, wsb + L.ws_bank, L.ws_bank_bytes, stream))
            return rc;
    }

    // ---- neighbour halo: the last local keyframe's pivots, inverse norms and attention output -> slot 0 of rank r+1
    const int to = rk->rank + 1 < W ? rk->rank + 1 : -1, from = rk->rank > 0 ? rk->rank - 1 : -1;
    if (!no_halo && (to >= 0 || from >= 0)) {
        if (const int rc = order(rk, st, rk->hs, "tf_rank_pivotal")) return rc;
        // ONE grouped exchange for the five messages (the entry point moves bytes: the 16-bit rows are counted as
        // SD/2 four-byte elements next to the fp32 inverse norms; S*D is even, D being a multiple of 8)
        const void* snd[5] = {piv + (int64_t)Kl * SD, kfo + (int64_t)Kl * SD, kfo + o_bs + (int64_t)Kl * SD,
                              kfo + 2 * o_bs + (int64_t)Kl * SD, inv_ext + (int64_t)Kl * S};
        void* rcv[5] = {piv, kfo, kfo + o_bs, kfo + 2 * o_bs, inv_ext};
        const int64_t n32[5] = {SD / 2, SD / 2, SD / 2, SD / 2, S};
        if (const int rc = tf_sendrecv_pivot(rk->halo_comm, snd, n32, 5, to, rcv, n32, 5, from, TF_F32, rk->hs)) return rc;
        TF_HIP(hipEventRecord(rk->halo_done[slot], rk->hs), "tf_rank_pivotal");
        rk->halo_set[slot] = true;
    }
    return 0;
}

extern "C" int tf_rank_halo_wait(tf_rank* rk, int slot, void* stream) {
    TF_ARG(rk, TF_ERR_NULL, "tf_rank_halo_wait: null pointer");
    TF_ARG(slot >= 0 && slot < TF_RANK_SLOTS, TF_ERR_SHAPE, "tf_rank_halo_wait: slot %d outside [0, %d)", slot,
           TF_RANK_SLOTS);
    if (rk->halo_set[slot])
        TF_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), rk->halo_done[slot], 0), "tf_rank_halo_wait");
    return 0;
}
