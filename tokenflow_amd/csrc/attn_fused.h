// Internal interface of the fused small-problem attention kernel (csrc/ext_attn_fused.hip).
// Used by tf_ext_attn_fwd[_strided] (csrc/ext_attn.hip) and by the rank executor (csrc/rank_exec.hip).
#pragma once
#include "tf_common.h"

// One tensor set of a fused launch.  A launch carries one or two sets (a sharded rank: the bank branches on the
// buffer its all-to-all delivered AND the source branch of its own frames on its local projections).
// Branch b of q / k / v / out lives at base + b * branch stride (the convention of tf_ext_attn_fwd_strided);
// branch 0 is the source branch (a frame attends to its own S keys), branches 1 and 2 attend to all Kb frames.
struct TfAttnSet {
    const void* q;
    const void* k;
    const void* v;
    void* out;
    int64_t q_bs, q_fs, ld_q;          // q[b*q_bs + f*q_fs + s*ld_q + h*Dh + c]          (elements)
    int64_t k_bs, k_fs, v_bs, v_fs, ld;   // k / v[b*bs + f*fs + s*ld + h*Dh + c]
    int64_t o_bs, o_fs;                // out[b*o_bs + f*o_fs + s*(H*Dh) + h*Dh + c]
    int H;          // heads of this set
    int Kq;         // query frames
    int q_frame0;   // bank index of query frame 0 (the source branch of frame f reads keys of bank frame q_frame0 + f)
    int Kb;         // bank frames
    int b0, nb;     // branches [b0, b0 + nb)
};

struct TfFusedPlan {
    int use;    // take the fused kernel
    int qw;     // query waves per workgroup (32 queries each; they share every staged key tile); 1 = the wave-private form
    int kw;     // key groups per workgroup (the in-workgroup split of the key sequence, merged through LDS)
    int qb;     // 32-query blocks per wave (wave-private form: 2 = every K / V^T fragment feeds two MFMAs)
    int prec;   // P carried as hi + lo bf16 (two P.V MFMAs): removes the rounding of P from the result
};

// Shape- and grid-based decision; `flags` = the `inject` bit mask of tf_ext_attn_fwd (hints included).
TfFusedPlan tf_attn_fused_plan(const TfAttnSet* sets, int n_sets, int S, int Dh, int dtype, int flags);

// One launch over every (set, branch, frame, head, query tile) problem.  No workspace, no pre-pass, no merge launch.
int tf_attn_fused_launch(const TfAttnSet* sets, int n_sets, int S, int Dh, float scale, int flags, int dtype,
                         const TfFusedPlan& plan, hipStream_t st);
