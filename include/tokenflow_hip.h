/*
 * tokenflow_hip.h  --  C ABI of libtokenflow_hip.so (MI355X / gfx950 only).
 *
 * The drop-in boundary for TokenFlow's per-step hot path.  The reference has no
 * native code and no FFI: its "operator interface" for this path is a set of
 * torch-op sequences inside Python hook closures.  Each entry point below
 * replaces one such sequence; the file:line after "replaces" is into
 * omerbt/TokenFlow (mounted at /root/reference in the build container).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / STL types.
 *   - every function returns int: 0 = ok, >0 = hipError_t of a failed launch,
 *     <0 = argument error (TF_ERR_*); never throws, never allocates, never
 *     synchronises, keeps no global mutable state (re-entrant per stream).
 *   - pointers are DEVICE pointers (tensor.data_ptr()); the stream is a
 *     hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all launches are asynchronous on that stream.
 *   - tensors are dense row-major unless a leading dimension is given.
 *   - branch layout everywhere is the reference's [source | uncond | cond]
 *     (tokenflow_utils.py:117,312  `n_frames = batch_size // 3`).
 */
#ifndef TOKENFLOW_HIP_H
#define TOKENFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TF_ABI_VERSION 7

/* Every entry point below is exported with default visibility; the library itself is built with -fvisibility=hidden, so
 * its exported symbols are exactly the declarations of this header (checked by tests/test_hooks_cpu.py). */
#define TF_API __attribute__((visibility("default")))

/* element types */
#define TF_BF16 0
#define TF_F16 1
#define TF_F32 2

/* tf_ext_attn_fwd flags (the `inject` argument is a bit mask) */
#define TF_ATTN_INJECT 1       /* q/k injection: uncond and cond use the source branch's q and k */
#define TF_ATTN_EXACT_SCALE 2  /* scale the scores in fp32: the default since ABI 2, the bit is accepted and ignored */
#define TF_ATTN_BANK_ONLY 4    /* compute only the uncond and cond branches (those that read the K-frame bank) */
#define TF_ATTN_SOURCE_ONLY 8  /* compute only the source branch (own-frame keys) */
#define TF_ATTN_NO_SPLIT 16    /* never split a bank problem over workgroups (one pass, bit-stable across grid sizes) */
#define TF_ATTN_FOLD_SCALE 64  /* Dh = 40 only: fold scale*log2e into q, rounded to the input dtype (faster, less exact) */
#define TF_ATTN_OUT_F32 32     /* `out` is float: the normalised fp32 accumulator, without the rounding to the 16-bit I/O type */
#define TF_ATTN_NO_FUSED 128   /* never take the fused small-problem kernel (below): the streaming kernels at every size */
#define TF_ATTN_FUSED (1 << 17) /* take the fused small-problem kernel at any size it is built for */
/* tuning hints of the fused small-problem kernel (0 = automatic; A/B measurements, tools/attn_microbench.py) */
#define TF_ATTN_HINT_QW(code) (((code) & 7) << 8)    /* code 1, 2, 3 = 1 (the wave-private form), 2, 4 query waves per workgroup */
#define TF_ATTN_HINT_KW(code) (((code) & 7) << 11)   /* code 1, 2, 3, 4 = 1, 2, 4, 8 key groups per workgroup */
#define TF_ATTN_HINT_QB2 (1 << 14)                   /* wave-private form, head dims <= 80: two 32-query blocks per wave */
#define TF_ATTN_PRECISE_P (1 << 15)                  /* fused kernel, bf16: carry P as hi + lo whatever the size */
#define TF_ATTN_NO_PRECISE_P (1 << 16)               /* fused kernel: P in one 16-bit value whatever the size */
#define TF_ATTN_HINT_MIX (1 << 18)                   /* Dh = 40 streaming kernel: the mixed-MFMA-shape form whatever the launch size (it is the
                                                        default for launches of >= 1024 workgroups outside the bit-stable mode); tests, measurements */

/* argument errors */
#define TF_ERR_NULL (-1)
#define TF_ERR_DTYPE (-2)
#define TF_ERR_SHAPE (-3)
#define TF_ERR_ALIGN (-4)
#define TF_ERR_WORKSPACE (-5)
#define TF_ERR_COMM (-6)      /* RCCL not loadable, or a collective failed (tf_last_error has RCCL's message) */

TF_API int tf_abi_version(void);

/* Thread-local description of the last non-zero return value on this thread. */
TF_API const char* tf_last_error(void);

/* ------------------------------------------------------------------------
 * Extended attention  --  replaces the body of sa_forward.forward between the
 * q/k/v projections and to_out: tokenflow_utils.py:124-197 (PnP variant) and
 * 234-279 (SDEdit variant; call with inject = 0).
 *
 *   k, v    : [3, K, S, H*Dh]   the key/value bank of all K keyframes (token stride = ld
 *                               elements, ld >= H*Dh)
 *   q       : [3, Kq, S, H*Dh]  queries of keyframes q_frame0 .. q_frame0+Kq-1 (same ld).
 *                               Single GPU: Kq = K, q_frame0 = 0 (q is the reference's q).
 *                               Frame-sharded multi-GPU: a rank passes its own keyframes' q
 *                               and the all-gathered bank.
 *   out     : [3, Kq, S, H*Dh]  dense, same dtype (float with TF_ATTN_OUT_F32)
 *   source branch: frame f attends to its own S keys (lines 173,177);
 *   uncond / cond: frame f attends to all K*S keys of its branch (133-138,
 *   174-179) -- the bank is read in place, never replicated.
 *   inject & TF_ATTN_INJECT: uncond and cond use the SOURCE branch's q and k (124-130),
 *   by pointer aliasing; q and k are not modified.
 *   The scores are scaled in fp32 after the QK^T product, as the reference does (`* self.scale`, 173-175).
 *   inject & TF_ATTN_FOLD_SCALE (opt-in, Dh = 40 only): scale*log2(e) is folded into q, rounded to the input dtype
 *   once (relative error <= 2^-9 per element in bf16): several % faster, inside the parity bound on unit-variance
 *   logits but 3-12x outside it on peaked ones (logit std 4-16; profiles/r02_fold_accuracy.txt) -- a speed knob
 *   for callers who accept that, never the default.
 *   scale = attn.scale (Dh^-0.5).  Dh in {40, 64, 80, 160}; dtype bf16 or f16;
 *   any S >= 1 (latent grids of odd resolutions: 9x5 = 45 tokens at the mid block of 576x320);
 *   ld a multiple of 8.  fp32 softmax / accumulation, online softmax over
 *   64-key tiles, P rounded to the input dtype before P.V (as the reference's
 *   autocast path does).
 *
 *   inject & TF_ATTN_BANK_ONLY / TF_ATTN_SOURCE_ONLY: compute only the uncond + cond branches / only the
 *   source branch (the head-sharded multi-GPU path runs them on different tensors); slabs of q, k, v, out
 *   that the selected part does not need are never touched.
 *   Small grids (a sharded rank, the coarse levels) split every bank problem into runs of bank frames over
 *   extra workgroups and merge the partial results (fp32) in a second launch; TF_ATTN_NO_SPLIT keeps the
 *   one-pass form, whose arithmetic per (query, head) does not depend on the grid.
 *
 *   inject & TF_ATTN_OUT_F32: `out` is float [3, Kq, S, H*Dh]; the softmax-normalised fp32 accumulator is stored
 *   as is.  Removes the output rounding (2^-9 relative for bf16) from the result: the mode in which the
 *   "< 1e-3 per token" target of BASELINE.json holds for outputs of any magnitude.
 *
 *   Small problems -- S <= 256 always; S <= 1024 on small grids (a sharded rank, BASELINE config 1) unless
 *   TF_ATTN_NO_SPLIT -- run in ONE fused launch (csrc/ext_attn_fused.hip): V is transposed inside the kernel
 *   (ds_read_b64_tr_b16, no pre-pass), the key sequence is split over the wave groups of a workgroup and merged
 *   through LDS (no partials, no merge launch), and for bf16 at S <= 256 P is carried as hi + lo bf16 so that the
 *   rounding of P (2^-9; the reference's fp16 autocast path rounds to 2^-11 at this point) drops out of the result.
 *   Which calls take it is a function of the SHAPE alone under TF_ATTN_NO_SPLIT, so one-pass results stay
 *   bit-identical between a sharded rank and the single GPU; TF_ATTN_NO_FUSED keeps the streaming kernels.
 *
 *   ws: scratch for the transposed V bank (+ key norms, + split-form partials); size from
 *   tf_ext_attn_workspace_bytes.
 * ------------------------------------------------------------------------ */
TF_API size_t tf_ext_attn_workspace_bytes(int K, int S, int H, int Dh, int dtype);

TF_API int tf_ext_attn_fwd(const void* q, const void* k, const void* v, void* out,
                    int K, int Kq, int q_frame0, int S, int H, int Dh, int64_t ld, float scale,
                    int inject, int dtype, void* ws, size_t ws_bytes, void* stream);

/* The same with explicit branch / frame strides (elements), for callers whose q, k, v arrive in the layout a
 * collective delivers them and whose output feeds the next collective (tokenflow_amd/sharded.py: the received
 * all-to-all buffer is [frame][slab][S][H*Dh], the returned one [frame][branch][S][H*Dh]):
 *   strides = { q_branch, q_frame, k_branch, k_frame, v_branch, v_frame, out_branch, out_frame, q_token }   (9 values)
 *   element (b, f, s, c) of k / v is k[b*k_branch + f*k_frame + s*ld + c], of q  q[b*q_branch + f*q_frame + s*q_token + c]
 *   (q has its own token stride since ABI 3: a rank's queries may be a column slab of its fused projection output
 *   while the bank arrives from an all-gather as dense slabs); out has token stride H*Dh.
 * Branch b of a tensor is addressed as base + b*branch_stride even when a call never touches branch 0 (bank-only
 * calls): pass base = (first touched slab) - b*branch_stride.  tf_ext_attn_fwd is this function with dense strides. */
TF_API int tf_ext_attn_fwd_strided(const void* q, const void* k, const void* v, void* out,
                            int K, int Kq, int q_frame0, int S, int H, int Dh, int64_t ld, const int64_t* strides,
                            float scale, int inject, int dtype, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Frames <-> heads re-sharding of the multi-GPU pivotal pass (no counterpart in the single-process reference;
 * tokenflow_amd/sharded.py).  Rank r sends head group w of its Kl keyframes' slabs to rank w:
 *   tf_head_pack:    send[w][f][i][s][0..hd)  = slab_i[f][s][w*hd .. (w+1)*hd)     i < ns <= 6 slabs, each a
 *                    [Kl, S, ld] tensor with its own frame stride (elements); one launch for all slabs.
 *   tf_head_unpack:  dst_b[f][s][w*hd ..)     = recv[w][f][b][s][0..hd)            b < nb <= 6 destinations.
 * elem_bytes 2 or 4; hd*elem_bytes and ld*elem_bytes multiples of 16.
 * ------------------------------------------------------------------------ */
TF_API int tf_head_pack(const void* const* slabs, const int64_t* frame_strides, int ns, void* send, int W, int Kl, int S,
                 int hd, int64_t ld, int elem_bytes, void* stream);

TF_API int tf_head_unpack(const void* recv, void* const* dsts, const int64_t* frame_strides, int nb, int W, int Kl, int S,
                   int hd, int64_t ld, int elem_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Nearest-neighbour token search  --  replaces batch_cosine_sim + chunk + argmax:
 * util.py:61-69 and tokenflow_utils.py:335-343.
 *
 * tf_pivot_inv_norm: once per pivotal pass and block, inv_norm[r] = 1/||piv[r]||_2
 *   for the rows of the source-branch pivots `pivot_hidden_states[0]` viewed as
 *   [rows = K*S, D] (util.py:67).
 *
 * tf_nn_search: for every target row t (norm_hidden_states[0] as [n_tgt = n*S, D],
 *   tokenflow_utils.py:335) and every selected keyframe p < P (kf[p] indexes the
 *   K keyframes; the reference order is [i, i-1], lines 331-333):
 *       idx[p, t] = argmax_j  <tgt[t], piv[kf[p], j]> * inv_norm[kf[p]*S + j]
 *   first maximal j wins (torch.argmax).  The 1/||tgt[t]|| factor of util.py:66
 *   is a positive per-row constant and cannot change the argmax, so targets are
 *   never normalised.  idx is int32 [P, n_tgt].  D multiple of 8; dtype bf16/f16.
 *   tgt, piv, ws AND inv_norm 16-byte aligned (TF_ERR_ALIGN otherwise; the same holds for the tf_nn_gather_blend*
 *   forms): the LDS-DMA search kernel fetches the inverse norms of a pivot tile in 16-byte pieces.
 *   ws: scratch for per-split candidates when the pivot range is split over workgroups
 *   (size from tf_nn_search_workspace_bytes, >= 256 bytes).
 * ------------------------------------------------------------------------ */
TF_API int tf_pivot_inv_norm(const void* piv, float* inv_norm, int64_t rows, int D, int dtype,
                      void* stream);

TF_API size_t tf_nn_search_workspace_bytes(int64_t n_tgt, int S, int D, int P);

TF_API int tf_nn_search(const void* tgt, const void* piv, const float* inv_norm, int32_t* idx,
                 int64_t n_tgt, int S, int D, int P, int kf0, int kf1, int dtype,
                 void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Gather + blend + residual  --  replaces tokenflow_utils.py:362-397
 * (propagation branch): advanced-index of the cached keyframe outputs, the
 * int64 [3, n*S, D] index expansion + gather (372-373 / 390), the materialised
 * fp32 weight tensor (375-385), the blend (388) and the residual add (396-397).
 *
 *   kf_out : [3, K, S, D]      cached attn1 output of the pivotal pass (`in_dtype`)
 *   idx    : int32 [P, n*S]    from tf_nn_search (same indices for all 3 branches, 344-348)
 *   w      : float [n]         w1 per frame of the chunk (only read when P == 2)
 *   resid  : [3, n, S, D] or NULL (`res_dtype`)   hidden_states added at 396-397
 *   out    : [3, n, S, D]      (`out_dtype`)
 *   P == 2:  out = (w*a1 + (1-w)*a2) + resid   evaluated in fp32 with the
 *            reference's operation order and no fused multiply-add, so an fp32
 *            `out` is bit-identical to the reference's.
 *   P == 1:  out = a1 + resid  (fp32 add, rounded to out_dtype).
 * ------------------------------------------------------------------------ */
TF_API int tf_gather_blend(const void* kf_out, const int32_t* idx, const float* w, const void* resid,
                    void* out, int K, int n, int S, int D, int P, int kf0, int kf1,
                    int in_dtype, int res_dtype, int out_dtype, void* stream);

/* ------------------------------------------------------------------------
 * Propagation branch in one call  --  tokenflow_utils.py:329-397 for one chunk:
 * tf_nn_search followed by tf_gather_blend, without the index tensor in between.
 * The search leaves its per-split candidates in `ws` and the gather merges them
 * itself (same order, same tie rule), which saves the finalize launch -- about
 * 4 us per call, i.e. what a launch costs on this GPU even when it does nothing.
 * Results are bit-identical to the two separate calls.  n_tgt = n*S; arguments as
 * in the two functions above; `search_dtype` is the dtype of tgt and piv.
 * ------------------------------------------------------------------------ */
TF_API size_t tf_nn_gather_blend_workspace_bytes(int64_t n_tgt, int S, int D, int P);

TF_API int tf_nn_gather_blend(const void* tgt, const void* piv, const float* inv_norm,
                       const void* kf_out, const float* w, const void* resid, void* out,
                       int K, int n, int S, int D, int P, int kf0, int kf1,
                       int search_dtype, int in_dtype, int res_dtype, int out_dtype,
                       void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Propagation branch for a RUN of consecutive chunks in one call  --  tokenflow_utils.py:329-397 executed for
 * C chunks at once.  The reference runs one UNet pass per chunk of n = batch_size frames
 * (run_tokenflow_pnp.py:228-231) because a pass over all frames does not fit its GPUs; with 288 GB a caller can
 * carry every chunk in one pass (or, as bench.py and the frame-sharded path do, simply owns all chunks' tensors),
 * and the C searches + gathers become two launches whose grids are C times larger (no short tail round per chunk).
 *
 *   tgt    : [C*n*S, D]        norm_hidden_states[0], chunk-major (chunk j = rows j*n*S .. (j+1)*n*S-1)
 *   resid, out : [3, C*n, S, D]
 *   chunk j matches keyframe slots slot0 + j and slot0 + j - 1 of piv / inv_norm / kf_out ([.., K, ..]; the reference
 *   order [i, i-1], 331-333).  first_single != 0: chunk 0 of the call is chunk 0 of the video and matches slot0
 *   alone (line 390); its rows are rounded to `single_dtype` -- the dtype the reference's pass produces for that
 *   chunk, torch promotion of the cached output and hidden_states -- before being stored as `out_dtype`.
 *   Results are bit-identical to C calls of tf_nn_gather_blend.  C = 1 with first_single is tf_nn_gather_blend(P = 1).
 *   w : float [n], as tf_gather_blend.
 * ------------------------------------------------------------------------ */
TF_API size_t tf_nn_gather_blend_chunks_workspace_bytes(int64_t n_tgt_chunk, int S, int D, int C);

TF_API int tf_nn_gather_blend_chunks(const void* tgt, const void* piv, const float* inv_norm, const void* kf_out,
                              const float* w, const void* resid, void* out, int K, int n, int C, int S, int D,
                              int slot0, int first_single, int search_dtype, int in_dtype, int res_dtype,
                              int out_dtype, int single_dtype, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * The two calls above with the block's NEXT LayerNorm fused behind them  --  tokenflow_utils.py:329-397 followed by
 * norm2 (399-403) or, without cross-attention, norm3 (414): `out` is written exactly as by tf_nn_gather_blend[_chunks]
 * and norm_out = LayerNorm(out) row by row exactly as tf_layer_norm(out) would produce it (bit-identical on both),
 * but a row never leaves the registers in between: the propagation leaves the residual stream in fp32 (the
 * reference's promotion, 385-397) and a separate norm re-reads 4 bytes per element to emit 2.
 * Covers the hook path's types: in_dtype = res_dtype = norm_dtype = the 16-bit model type, resid != NULL,
 * out_dtype = TF_F32 when two keyframes are blended (always for the chunks form), the model type for P = 1;
 * D <= 1536 (TF_ERR_DTYPE otherwise: issue the two calls).  gamma, beta: [D] of w_dtype or NULL.  Workspace as the
 * unfused calls.
 * ------------------------------------------------------------------------ */
TF_API int tf_nn_gather_blend_norm(const void* tgt, const void* piv, const float* inv_norm, const void* kf_out,
                            const float* w, const void* resid, void* out, int K, int n, int S, int D, int P, int kf0,
                            int kf1, int search_dtype, int in_dtype, int res_dtype, int out_dtype, const void* gamma,
                            const void* beta, float eps, int w_dtype, void* norm_out, int norm_dtype, void* ws,
                            size_t ws_bytes, void* stream);

TF_API int tf_nn_gather_blend_chunks_norm(const void* tgt, const void* piv, const float* inv_norm, const void* kf_out,
                                   const float* w, const void* resid, void* out, int K, int n, int C, int S, int D,
                                   int slot0, int first_single, int search_dtype, int in_dtype, int res_dtype,
                                   int out_dtype, int single_dtype, const void* gamma, const void* beta, float eps,
                                   int w_dtype, void* norm_out, int norm_dtype, void* ws, size_t ws_bytes,
                                   void* stream);

/* ------------------------------------------------------------------------
 * Row LayerNorm producer  --  the `norm1` call of TokenFlowBlock.forward
 * (tokenflow_utils.py:313-323; also norm2 / norm3 of the same forward, 399-417)
 * when the block runs in 16 bit:  out[r] = (x[r] - mean) / sqrt(var + eps) * gamma + beta
 * with fp32 statistics (biased variance, as torch) and ONE rounding to out_dtype,
 * instead of torch's autocast sequence cast-up / fp32 norm / cast-down.
 *   x      : [rows, D]  (in_dtype)      D multiple of 8, D <= 2048
 *   gamma, beta : [D] (w_dtype) or NULL (= 1 / 0)
 *   out    : [rows, D]  (out_dtype)
 *   inv_norm : float [rows] or NULL:  1 / ||out[r]||_2 of the ROUNDED output row -- what
 *              tf_pivot_inv_norm would compute from the stored pivots (util.py:67).
 * ------------------------------------------------------------------------ */
TF_API int tf_layer_norm(const void* x, const void* gamma, const void* beta, void* out, float* inv_norm,
                  int64_t rows, int D, float eps, int in_dtype, int w_dtype, int out_dtype,
                  void* stream);

/* Residual add + LayerNorm  --  `hidden_states = attn_output + hidden_states` followed by the block's next norm
 * (tokenflow_utils.py:396-403 and 409-414), one pass instead of an add kernel plus a norm:
 *   sum_out[r] = round_to(sum_dtype, a[r] + b[r])        (what torch's `a + b` stores: fp32 add, promoted dtype)
 *   out[r]     = LayerNorm(sum_out[r])                   (as tf_layer_norm, on the ROUNDED sum)
 * a, b, sum_out: [rows, D] of a_dtype / b_dtype / sum_dtype.  Results are bit-identical to the add followed by
 * tf_layer_norm. */
TF_API int tf_add_layer_norm(const void* a, const void* b, void* sum_out, const void* gamma, const void* beta, void* out,
                      int64_t rows, int D, float eps, int a_dtype, int b_dtype, int sum_dtype, int w_dtype,
                      int out_dtype, void* stream);

/* ------------------------------------------------------------------------
 * DDIM latent update  --  the step BEFORE the hot path (row f4): replaces preprocess.py:224-225 (ddim_inversion)
 * and 259-260 (ddim_sample), six elementwise torch ops per UNet call in the loops that write / check the latents
 * directory:
 *     out = mu_b * ((x - sigma_a * eps) / mu_a) + sigma_b * eps          (n elements; out may alias x)
 * inversion at step i:  a = timestep i-1 (mu_prev, sigma_prev), b = timestep i;  sampling: a = t, b = the next one.
 * Same operation order and the same per-op rounding to `dtype` as the reference's sequence (fp32 scalars, no FMA,
 * IEEE division): bit-identical to it in f32 / f16 / bf16.
 * ------------------------------------------------------------------------ */
TF_API int tf_ddim_step(const void* x, const void* eps, void* out, int64_t n, float mu_a, float sigma_a, float mu_b,
                 float sigma_b, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * PnP feature injection  --  replaces tokenflow_utils.py:87-91:
 *   x viewed as [3, elems_per_branch]:  x[1] = x[0];  x[2] = x[0]   (in place)
 * elem_bytes = bytes per element; elems_per_branch*elem_bytes multiple of 16.
 * ------------------------------------------------------------------------ */
TF_API int tf_inject_copy(void* x, int64_t elems_per_branch, int elem_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Multi-GPU exchange steps over RCCL (one process per GPU).  The reference is single-process (no
 * parallelism of any kind), so these replace nothing in it; they are the C-ABI form of the two exchange steps of
 * tokenflow_amd/sharded.py for hosts that do not go through torch.distributed (the Python host does, and never calls
 * these).  Frames are sharded over the ranks:
 *   pivotal pass  (tokenflow_utils.py:133-138: every keyframe's queries read the keys/values of ALL K keyframes)
 *       tf_allgather_kv      each rank contributes its keyframes' slab, all ranks receive the bank (equal slabs), or
 *       tf_allgather_rows    the same for runs of different lengths (K % W != 0): rank p contributes rows[p] rows of
 *                            row_elems elements, every rank receives all of them in rank order, straight into place, or
 *       tf_all_to_all_rows   frames <-> heads re-sharding: rows [send_rows[p]] to peer p, [recv_rows[p]] from peer p
 *                            (row = row_elems elements; send / recv buffers are the concatenation in peer order);
 *   propagation   (331-333: chunk c reads keyframes c and c-1)
 *       tf_sendrecv_pivot    the last local keyframe's pivot features / inverse norms / attention output go to
 *                            send_peer (rank + 1) while the left neighbour's arrive from recv_peer (rank - 1);
 *                            a peer of -1 skips that direction (first / last rank).
 * Exceptions to the conventions at the top, all of them RCCL's: tf_comm_init / tf_comm_destroy allocate and free the
 * communicator handle and may synchronise; RCCL (librccl.so.1) is dlopen-ed on the first tf_comm_* call -- inside a
 * PyTorch process that is the copy torch has loaded -- and TF_ERR_COMM is returned if it is absent.  The collectives
 * themselves are asynchronous on `stream` like every other entry point.  tf_comm_init binds the calling thread's
 * current device (hipSetDevice first).  The unique id (128 bytes) is created on one rank by tf_comm_unique_id and
 * handed to the others by the host (file, socket, MPI ...).
 * ------------------------------------------------------------------------ */
typedef struct tf_comm tf_comm;
TF_API int tf_comm_available(void);   /* 0 if RCCL can be loaded and has every entry point used here; starts no
                                         bootstrap listener (a pre-flight probe: tf_comm_unique_id does start one) */
TF_API int tf_comm_unique_id(void* id_out_128_bytes);
TF_API int tf_comm_init(const void* unique_id_128_bytes, int rank, int world, tf_comm** comm_out);

/* Two more transports behind the same exchange entry points (no RCCL needed for either):
 *   tf_comm_init_hooks     a host-provided transport -- the exchanges are handed to the function table (sizes in
 *                          BYTES, device pointers, the stream the caller's work is ordered on; a function returns 0 or
 *                          an error code that surfaces as TF_ERR_COMM).  For hosts with their own fabric (MPI ...);
 *                          the multi-process tests of this repository carry it over gloo to run W ranks on one GPU.
 *   tf_comm_init_loopback  the wire-less stand-in: every exchange becomes device-to-device copies, on the stream, of
 *                          the sizes a real rank of a `world`-GPU run would receive, out of this rank's own buffers.
 *                          A rank's complete launch / stream / buffer sequence on ONE GPU: for timing the host and GPU
 *                          side of a rank (tools/rank_step_microbench.py); the received DATA is meaningless for
 *                          world > 1.
 * The peer arithmetic (rank, world, row counts) is the callers' in all three. */
#define TF_MAX_WORLD 64
typedef struct tf_comm_hooks {
    int (*all_to_all_rows)(void* user, const void* send, void* recv, const int64_t* send_rows, const int64_t* recv_rows,
                           int64_t row_bytes, void* stream);
    int (*allgather_rows)(void* user, const void* local, void* bank, const int64_t* rows, int64_t row_bytes,
                          void* stream);
    int (*sendrecv)(void* user, const void* const* send, const int64_t* send_bytes, int n_send, int send_peer,
                    void* const* recv, const int64_t* recv_bytes, int n_recv, int recv_peer, void* stream);
    void* user;
} tf_comm_hooks;
TF_API int tf_comm_init_hooks(const tf_comm_hooks* hooks, int rank, int world, tf_comm** comm_out);
TF_API int tf_comm_init_loopback(int rank, int world, tf_comm** comm_out);
/* loopback only: enabled = 0 makes every exchange a no-op (nothing moves, nothing is enqueued) -- the rank's launch sequence
 * with the stand-in copies taken out of the timing as well; buffers that an exchange would have filled keep their contents */
TF_API int tf_comm_loopback_copies(tf_comm* comm, int enabled);
/* loopback only (ABI 7): a wire MODEL.  Behind its copies (if enabled) every exchange enqueues, on the stream it was issued
 * on, a one-wave kernel that holds the stream for  latency_us + bytes_on_the_busiest_link / (gbps_per_link GB/s):  the
 * schedule's overlap of exchanges with compute is then EXECUTED, not estimated (tools/rank_step_microbench.py
 * --wire-model).  Links are full duplex and point to point (xGMI): an all-to-all or all-gather puts the rows for / from
 * peer p on the link to p, the busiest link carries max_p max(sent_p, received_p) bytes; a neighbour exchange sends on
 * one link and receives on another.  latency_us <= 0 and gbps_per_link <= 0 switch the model off (the default). */
TF_API int tf_comm_loopback_wire(tf_comm* comm, double latency_us, double gbps_per_link);
TF_API int tf_comm_destroy(tf_comm* comm);
TF_API int tf_comm_rank(const tf_comm* comm);
TF_API int tf_comm_world(const tf_comm* comm);
TF_API int tf_allgather_kv(tf_comm* comm, const void* local, void* bank, int64_t elems_per_rank, int dtype, void* stream);
TF_API int tf_allgather_rows(tf_comm* comm, const void* local, void* bank, const int64_t* rows, int64_t row_elems, int dtype,
                      void* stream);
TF_API int tf_all_to_all_rows(tf_comm* comm, const void* send, void* recv, const int64_t* send_rows, const int64_t* recv_rows,
                       int64_t row_elems, int dtype, void* stream);
TF_API int tf_sendrecv_pivot(tf_comm* comm, const void* const* send, const int64_t* send_elems, int n_send, int send_peer,
                      void* const* recv, const int64_t* recv_elems, int n_recv, int recv_peer, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * One rank's pivotal pass of a block, issued by ONE host call  --  the native form of tokenflow_amd/sharded.py's
 * `pivotal_block` (no counterpart in the single-process reference; what is computed is tokenflow_utils.py:124-197
 * for the rank's keyframes against the bank of all K).  Rank r of W owns a contiguous run of the K keyframes (the
 * first K % W ranks one more); per block it
 *   TF_RANK_HEADS  packs head group w of its keyframes' q / k / v for every rank w (tf_head_pack), exchanges them
 *                  (tf_all_to_all_rows), computes the source branch of its own frames and
 *                  the uncond / cond branches of ITS head group over all K frames in place on the received buffer
 *                  (tf_ext_attn_fwd_strided, TF_ATTN_BANK_ONLY), sends the outputs back and unpacks them
 *                  (tf_head_unpack); needs H % W == 0 and one token stride for q, k, v;
 *   TF_RANK_BANK   gathers the K/V slabs of all ranks (tf_allgather_rows) and computes its own keyframes' queries
 *                  against the gathered bank: one collective, 4x the bytes; any head count;
 * then sends its last keyframe's pivot features, inverse norms and attention output to rank r+1 (tf_sendrecv_pivot on
 * the halo communicator and a stream of its own: chunk c of the propagation reads keyframes c and c-1, 331-333).
 *
 *   q, k, v   : the rank's local [3, Kl, S, H*Dh] projections; strides (elements) =
 *               { q_branch, q_frame, k_branch, k_frame, v_branch, v_frame, q_token, kv_token }
 *   piv_ext   : [Kl+o, S, H*Dh] 16-bit, inv_ext : float [Kl+o, S], kfo_ext : [3, Kl+o, S, H*Dh] 16-bit, o = 1 for
 *               W > 1 else 0: the per-block state of the propagation with the halo slot in front.  The caller has
 *               written the local keyframes' pivots / inverse norms into slots o..; the call writes the attention
 *               output into slots o.. of kfo_ext and the neighbour's last keyframe arrives in slot 0 of all three.
 *   flags     : TF_ATTN_INJECT / TF_ATTN_NO_SPLIT / TF_ATTN_FOLD_SCALE as for tf_ext_attn_fwd
 *   slot      : names this block's halo exchange, 0 <= slot < TF_RANK_SLOTS: tf_rank_halo_wait(rk, slot, stream) orders
 *               `stream` behind it (call it before the propagation reads slot 0; no host blocking)
 *   ws        : tf_rank_pivotal_workspace_bytes; holds the exchange buffers, so ONE workspace serves consecutive
 *               blocks of one stream (each use is complete before the next block touches it), not concurrent ones.
 * Every collective of `comm` is issued on the caller's stream (one communicator, one stream).
 * tf_rank_create makes two streams (auxiliary compute, halo) and the events on the CURRENT device; `comm` may be NULL (one rank: plain
 * tf_ext_attn_fwd into kfo_ext), `halo_comm` NULL = comm (a second communicator lets the halo of block b travel
 * beside the exchanges of block b+1: collectives of one RCCL communicator execute in issue order).
 * ------------------------------------------------------------------------ */
#define TF_RANK_HEADS 0
#define TF_RANK_BANK 1
#define TF_RANK_NO_HALO 16   /* or-ed into `mode`: the attention alone -- kfo_ext is a plain [3, Kl, S, H*Dh] output, piv_ext /
                                inv_ext are not read (NULL allowed), no neighbour exchange (hosts whose cached attention
                                output is not this one: the hook path caches it after the to_out projection) */
#define TF_RANK_INV_NORM 32  /* or-ed into `mode`: inv_ext's local slots o.. are OUTPUTS -- the call computes 1 / ||row|| of the
                                local pivots in piv_ext (tf_pivot_inv_norm's arithmetic, bit for bit) inside its pack launch,
                                instead of the caller in a launch of its own */
#define TF_RANK_SLOTS 64
typedef struct tf_rank tf_rank;
TF_API int tf_rank_create(tf_comm* comm, tf_comm* halo_comm, int K, tf_rank** rank_out);
TF_API int tf_rank_destroy(tf_rank* rk);
TF_API int tf_rank_local_keyframes(const tf_rank* rk);
TF_API int tf_rank_first_keyframe(const tf_rank* rk);
TF_API size_t tf_rank_pivotal_workspace_bytes(const tf_rank* rk, int S, int H, int Dh, int dtype);
TF_API int tf_rank_pivotal(tf_rank* rk, const void* q, const void* k, const void* v, const int64_t* strides, void* piv_ext,
                    float* inv_ext, void* kfo_ext, int S, int H, int Dh, float scale, int flags, int dtype, int mode,
                    int slot, void* ws, size_t ws_bytes, void* stream);
TF_API int tf_rank_halo_wait(tf_rank* rk, int slot, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TOKENFLOW_HIP_H */
