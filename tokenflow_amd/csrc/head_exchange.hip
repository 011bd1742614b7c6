// Pack / unpack kernels of the frames <-> heads re-sharding of the multi-GPU pivotal pass (tokenflow_amd/sharded.py).
// The reference is single-process; these have no counterpart there.  They exist so that a rank's q / k / v slabs go
// into the all-to-all send buffer, and the returned attention outputs into the [3,Kl,S,D] result, in ONE launch each
// (the torch formulation was ~10 strided copy kernels per block -- at 8 ranks more host time than the rank's GPU work).
// Pure data movement, HBM-bound: one thread = one 16-byte piece, grid-stride.
#include "tf_common.h"

namespace {

constexpr int MAX_SLABS = 6;

struct SlabPtrs {
    const unsigned char* src[MAX_SLABS];   // pack: slab i = a [Kl, S, ld] tensor (frame stride fs[i]); unpack: dst[i]
    int64_t fs[MAX_SLABS];                  // frame strides in BYTES
};

// send[w][f][i][s][hd] = slab_i[f][s][w*hd ...]
__global__ __launch_bounds__(256) void head_pack_kernel(SlabPtrs sl, unsigned char* __restrict__ send, int ns, int W,
                                                        int Kl, int S, int hd_pieces, int64_t ld_bytes) {
    const int64_t total = (int64_t)W * Kl * ns * S * hd_pieces;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        int64_t t = g;
        const int pc = (int)(t % hd_pieces);
        t /= hd_pieces;
        const int s = (int)(t % S);
        t /= S;
        const int i = (int)(t % ns);
        t /= ns;
        const int f = (int)(t % Kl);
        const int w = (int)(t / Kl);
        const unsigned char* src = sl.src[i] + f * sl.fs[i] + (int64_t)s * ld_bytes + ((int64_t)w * hd_pieces + pc) * 16;
        st16(send + g * 16, ld16(src));
    }
}

// The same launch with extra workgroups behind the packing ones that compute inv_norm[r] = 1 / ||piv[r]|| of the rank's
// pivot rows (tf_pivot_inv_norm's arithmetic, one wave per row): at the coarse levels of a sharded rank a launch costs
// more than either piece of work.
template <typename T>
__global__ __launch_bounds__(256) void head_pack_norm_kernel(SlabPtrs sl, unsigned char* __restrict__ send, int ns, int W,
                                                             int Kl, int S, int hd_pieces, int64_t ld_bytes, int nb_pack,
                                                             const typename T::elem* __restrict__ piv,
                                                             float* __restrict__ inv_norm, int64_t rows, int D) {
    if ((int)blockIdx.x < nb_pack) {
        const int64_t total = (int64_t)W * Kl * ns * S * hd_pieces;
        for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)nb_pack * 256) {
            int64_t t = g;
            const int pc = (int)(t % hd_pieces);
            t /= hd_pieces;
            const int s = (int)(t % S);
            t /= S;
            const int i = (int)(t % ns);
            t /= ns;
            const int f = (int)(t % Kl);
            const int w = (int)(t / Kl);
            const unsigned char* src = sl.src[i] + f * sl.fs[i] + (int64_t)s * ld_bytes + ((int64_t)w * hd_pieces + pc) * 16;
            st16(send + g * 16, ld16(src));
        }
    } else {
        const int lane = threadIdx.x & 63;
        const int64_t nwaves = (int64_t)(gridDim.x - nb_pack) * 4;
        for (int64_t r = (int64_t)(blockIdx.x - nb_pack) * 4 + (threadIdx.x >> 6); r < rows; r += nwaves) {
            const float inv = tf_row_inv_norm<T>(piv + r * D, D, lane);
            if (lane == 0) inv_norm[r] = inv;
        }
    }
}

// dst_b[f][s][w*hd ...] = recv[w][f][b][s][hd]
__global__ __launch_bounds__(256) void head_unpack_kernel(const unsigned char* __restrict__ recv, SlabPtrs sl, int nb,
                                                          int W, int Kl, int S, int hd_pieces, int64_t ld_bytes) {
    const int64_t total = (int64_t)W * Kl * nb * S * hd_pieces;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        int64_t t = g;
        const int pc = (int)(t % hd_pieces);
        t /= hd_pieces;
        const int s = (int)(t % S);
        t /= S;
        const int b = (int)(t % nb);
        t /= nb;
        const int f = (int)(t % Kl);
        const int w = (int)(t / Kl);
        unsigned char* dst = const_cast<unsigned char*>(sl.src[b]) + f * sl.fs[b] + (int64_t)s * ld_bytes +
                             ((int64_t)w * hd_pieces + pc) * 16;
        st16(dst, ld16(recv + g * 16));
    }
}

int check(const char* name, const void* const* slabs, const int64_t* fs, int n, const void* buf, int W, int Kl, int S,
          int hd, int64_t ld, int elem_bytes) {
    TF_ARG(slabs && fs && buf, TF_ERR_NULL, "%s: null pointer", name);
    TF_ARG(n >= 1 && n <= MAX_SLABS && W >= 1 && Kl >= 1 && S >= 1 && hd >= 1 && (elem_bytes == 2 || elem_bytes == 4) &&
               ((int64_t)hd * elem_bytes) % 16 == 0 && ld >= (int64_t)W * hd && ((int64_t)ld * elem_bytes) % 16 == 0,
           TF_ERR_SHAPE, "%s: n=%d W=%d Kl=%d S=%d hd=%d ld=%lld elem_bytes=%d (hd*elem_bytes and ld*elem_bytes multiples of 16)",
           name, n, W, Kl, S, hd, (long long)ld, elem_bytes);
    TF_ARG(tf_aligned16(buf), TF_ERR_ALIGN, "%s: buffer not 16-byte aligned", name);
    for (int i = 0; i < n; ++i) {
        TF_ARG(slabs[i], TF_ERR_NULL, "%s: slab %d is null", name, i);
        TF_ARG(tf_aligned16(slabs[i]) && (fs[i] * elem_bytes) % 16 == 0, TF_ERR_ALIGN, "%s: slab %d not 16-byte aligned", name, i);
    }
    return 0;
}

}  // namespace

int tf_head_pack_norm(const void* const* slabs, const int64_t* frame_strides, int ns, void* send, int W, int Kl, int S,
                      int hd, int64_t ld, int elem_bytes, const void* piv, float* inv_norm, int64_t rows, int D, int dtype,
                      void* stream) {
    if (!piv) return tf_head_pack(slabs, frame_strides, ns, send, W, Kl, S, hd, ld, elem_bytes, stream);
    if (const int rc = check("tf_head_pack", slabs, frame_strides, ns, send, W, Kl, S, hd, ld, elem_bytes)) return rc;
    TF_ARG(inv_norm && rows > 0 && D > 0 && D % 8 == 0 && (dtype == TF_BF16 || dtype == TF_F16) && tf_aligned16(piv),
           TF_ERR_SHAPE, "tf_head_pack(+inverse norms): rows=%lld D=%d dtype=%d", (long long)rows, D, dtype);
    SlabPtrs sl{};
    for (int i = 0; i < ns; ++i) {
        sl.src[i] = static_cast<const unsigned char*>(slabs[i]);
        sl.fs[i] = frame_strides[i] * elem_bytes;
    }
    const int hd_pieces = hd * elem_bytes / 16;
    const int64_t total = (int64_t)W * Kl * ns * S * hd_pieces;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    int64_t nblocks = (rows + 3) / 4;
    if (nblocks > 4096) nblocks = 4096;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == TF_BF16)
        hipLaunchKernelGGL(head_pack_norm_kernel<BF16>, dim3((unsigned)(blocks + nblocks)), dim3(256), 0, st, sl,
                           static_cast<unsigned char*>(send), ns, W, Kl, S, hd_pieces, ld * elem_bytes, (int)blocks,
                           static_cast<const __bf16*>(piv), inv_norm, rows, D);
    else
        hipLaunchKernelGGL(head_pack_norm_kernel<F16>, dim3((unsigned)(blocks + nblocks)), dim3(256), 0, st, sl,
                           static_cast<unsigned char*>(send), ns, W, Kl, S, hd_pieces, ld * elem_bytes, (int)blocks,
                           static_cast<const _Float16*>(piv), inv_norm, rows, D);
    TF_LAUNCH_CHECK("tf_head_pack");
    return 0;
}

extern "C" int tf_head_pack(const void* const* slabs, const int64_t* frame_strides, int ns, void* send, int W, int Kl,
                            int S, int hd, int64_t ld, int elem_bytes, void* stream) {
    if (const int rc = check("tf_head_pack", slabs, frame_strides, ns, send, W, Kl, S, hd, ld, elem_bytes)) return rc;
    SlabPtrs sl{};
    for (int i = 0; i < ns; ++i) {
        sl.src[i] = static_cast<const unsigned char*>(slabs[i]);
        sl.fs[i] = frame_strides[i] * elem_bytes;
    }
    const int hd_pieces = hd * elem_bytes / 16;
    const int64_t total = (int64_t)W * Kl * ns * S * hd_pieces;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(head_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), sl,
                       static_cast<unsigned char*>(send), ns, W, Kl, S, hd_pieces, ld * elem_bytes);
    TF_LAUNCH_CHECK("tf_head_pack");
    return 0;
}

extern "C" int tf_head_unpack(const void* recv, void* const* dsts, const int64_t* frame_strides, int nb, int W, int Kl,
                              int S, int hd, int64_t ld, int elem_bytes, void* stream) {
    if (const int rc = check("tf_head_unpack", const_cast<const void* const*>(dsts), frame_strides, nb, recv, W, Kl, S, hd,
                             ld, elem_bytes))
        return rc;
    SlabPtrs sl{};
    for (int i = 0; i < nb; ++i) {
        sl.src[i] = static_cast<const unsigned char*>(dsts[i]);
        sl.fs[i] = frame_strides[i] * elem_bytes;
    }
    const int hd_pieces = hd * elem_bytes / 16;
    const int64_t total = (int64_t)W * Kl * nb * S * hd_pieces;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(head_unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       static_cast<const unsigned char*>(recv), sl, nb, W, Kl, S, hd_pieces, ld * elem_bytes);
    TF_LAUNCH_CHECK("tf_head_unpack");
    return 0;
}
