"""Oracle-backed stand-in for `tokenflow_amd.ops` (TEST INFRASTRUCTURE).

Lets the host logic of tokenflow_amd/hooks.py and tokenflow_amd/sharded.py run on CPU
tensors in `-m "not gpu"` tests: tests monkeypatch `hooks.ops` / `sharded.ops` with an
instance of this class.  The product never imports it.

round16=True mimics the precision contract of the HIP ops (inputs rounded to bf16, fp32
arithmetic, attention output rounded to bf16) so a CPU run can be compared with a GPU run.
"""
import torch

from oracle import tokenflow_oracle as orc


class FakeOps:
    def __init__(self, round16: bool = False):
        self.round16 = round16
        self.calls = []

    def _r(self, t):
        return t.to(torch.bfloat16).float() if self.round16 else t.float()

    def compute_dtype(self, t):
        return t.dtype

    def ext_attn(self, q, k, v, heads, scale, inject, out=None, q_frame0=0, fold_scale=None, part="all",
                 out_dtype=None, no_split=None):
        self.calls.append(("ext_attn", tuple(q.shape), bool(inject)) + ((part,) if part != "all" else ()))
        K, Kq = k.shape[0] // 3, q.shape[0] // 3
        qf, kf, vf = self._r(q).clone(), self._r(k).clone(), self._r(v).clone()
        if part != "all":     # slabs the call may not read: poison them (the HIP kernels never touch them)
            unread = [0] if part == "bank" else [1, 2]
            vf.view(3, K, -1)[unread] = float("nan")
            qk_unread = [1, 2] if (part == "source" or inject) else [0]
            qf.view(3, Kq, -1)[qk_unread] = float("nan")
            kf.view(3, K, -1)[qk_unread] = float("nan")
        if Kq != K:      # queries of a frame subset: embed at their global positions, slice the result
            full = torch.zeros(3, K, *q.shape[1:])
            full[:, q_frame0:q_frame0 + Kq] = qf.view(3, Kq, *q.shape[1:])
            qf = full.view(3 * K, *q.shape[1:])
        if part == "all":
            o = orc.ext_attn_core(qf, kf, vf, heads, scale, inject)
        else:            # run the oracle with the unread slabs replaced by readable zeros, poison its unused outputs
            qz, kz, vz = (torch.nan_to_num(t, nan=0.0) for t in (qf, kf, vf))
            for t, z in ((qf, qz), (kf, kz), (vf, vz)):
                assert torch.isnan(t).view(3, -1).all(1).tolist() == torch.isnan(t).view(3, -1).any(1).tolist()
            o = orc.ext_attn_core(qz, kz, vz, heads, scale, inject).clone()
            o.view(3, -1)[[0] if part == "bank" else [1, 2]] = float("nan")
        if Kq != K:
            o = o.view(3, K, *q.shape[1:])[:, q_frame0:q_frame0 + Kq].reshape(3 * Kq, *q.shape[1:])
        o = o.float() if out_dtype == torch.float32 else self._r(o).to(q.dtype)
        if out is None:
            return o
        computed = {"all": [0, 1, 2], "bank": [1, 2], "source": [0]}[part]
        out.view(3, -1)[computed] = o.view(3, -1)[computed]
        return out

    def ext_attn_views(self, q, k, v, out, heads, scale, inject, part="all", branch0=(0, 0, 0, 0), q_frame0=0,
                       fold_scale=None, no_split=None, stream=None):
        """Strided 4-D views [branches b0.., frames, S, D]: materialise dense [3F,S,D] tensors (branches a call
        may not read stay NaN), run `ext_attn`, scatter the computed branches into the `out` view."""
        K, Kq, S, D = k.shape[1], q.shape[1], k.shape[2], k.shape[3]

        def dense(t, b0, F):
            full = torch.full((3, F, S, D), float("nan"), dtype=t.dtype)
            full[b0:b0 + t.shape[0]] = t
            return full.view(3 * F, S, D)
        qd, kd, vd = dense(q, branch0[0], Kq), dense(k, branch0[1], K), dense(v, branch0[2], K)
        res = torch.full((3, Kq, S, D), float("nan"), dtype=out.dtype)
        # ext_attn poisons what it may not read; NaN slabs here play the same role for what was never passed
        qd2, kd2, vd2 = (torch.nan_to_num(t, nan=0.0) for t in (qd, kd, vd))
        self.ext_attn(qd2, kd2, vd2, heads, scale, inject, out=res.view(3 * Kq, S, D), q_frame0=q_frame0, part=part,
                      out_dtype=out.dtype)
        computed = {"all": [0, 1, 2], "bank": [1, 2], "source": [0]}[part]
        need_q = [0] if (inject or part == "source") else ([1, 2] if part == "bank" else [0, 1, 2])
        if inject and part == "all":
            need_q = [0]
        need_v = computed
        for b in need_q:
            assert not torch.isnan(qd.view(3, -1)[b]).any() and not torch.isnan(kd.view(3, -1)[b]).any(), "unread q/k slab"
        for b in need_v:
            assert not torch.isnan(vd.view(3, -1)[b]).any(), "unread v slab"
        for b in computed:
            out[b - branch0[3]] = res[b]
        return out

    def head_pack(self, slabs, W, out=None):
        Kl, S, D = slabs[0].shape
        hd = D // W
        st = torch.stack([t.reshape(Kl, S, W, hd) for t in slabs], dim=1)     # [Kl, ns, S, W, hd]
        res = st.permute(3, 0, 1, 2, 4).contiguous()                           # [W, Kl, ns, S, hd]
        return res if out is None else out.copy_(res)

    def head_unpack(self, recv, dsts):
        W, Kl, nb, S, hd = recv.shape
        for b, d in enumerate(dsts):
            d.view(Kl, S, W, hd).copy_(recv[:, :, b].permute(1, 2, 0, 3))

    def pivot_inv_norm(self, piv, out=None):
        res = 1.0 / self._r(piv).norm(dim=-1)
        return res if out is None else out.copy_(res)

    def nn_search(self, tgt, piv, inv_norm, kf_ids):
        self.calls.append(("nn_search", tuple(tgt.shape), tuple(kf_ids)))
        sim = orc.batch_cosine_sim(self._r(tgt), self._r(piv[list(kf_ids)]).reshape(-1, piv.shape[-1]))
        return torch.stack([c.argmax(-1) for c in sim.chunk(len(kf_ids), dim=1)]).to(torch.int32)

    def gather_blend(self, kf_out, idx, w, kf_ids, n, residual, out_dtype):
        self.calls.append(("gather_blend", tuple(kf_out.shape), tuple(kf_ids)))
        BK, S, D = kf_out.shape
        sel = kf_out.view(3, BK // 3, S, D)
        a1 = sel[:, kf_ids[0]][:, idx[0].long()].float()
        if len(kf_ids) == 2:
            a2 = sel[:, kf_ids[1]][:, idx[1].long()].float()
            w1 = w.view(1, n, 1, 1)
            o = (w1 * a1.view(3, n, S, D) + (1 - w1) * a2.view(3, n, S, D)).reshape(3 * n, S, D)
        else:
            o = a1.reshape(3 * n, S, D)
        if residual is not None:
            o = o + residual.float()
        return o.to(out_dtype)

    def layer_norm(self, x, weight, bias, eps, out_dtype, want_inv_norm=False):
        y = torch.nn.functional.layer_norm(x.float(), (x.shape[-1],), None if weight is None else weight.float(),
                                           None if bias is None else bias.float(), eps).to(out_dtype)
        return y, (1.0 / y.float().norm(dim=-1) if want_inv_norm else None)

    def add_layer_norm(self, a, b, weight, bias, eps, out_dtype):
        total = a + b
        return total, self.layer_norm(total, weight, bias, eps, out_dtype)[0]

    def norm_fusable(self, kf_out, residual, out_dtype, P, norm_dtype):
        return residual is not None

    def _with_norm(self, out, norm):
        if norm is None:
            return out
        weight, bias, eps, ndt = norm
        return out, self.layer_norm(out, weight, bias, eps, ndt)[0]

    def propagate(self, tgt, piv, inv_norm, kf_ids, kf_out, w, n, residual, out_dtype, norm=None):
        out = self.gather_blend(kf_out, self.nn_search(tgt, piv, inv_norm, kf_ids), w, kf_ids, n, residual, out_dtype)
        return self._with_norm(out, norm)

    def propagate_chunks(self, tgt, piv, inv_norm, kf_out, w, n, n_chunks, slot0, first_single, residual, out_dtype,
                         norm=None):
        """C chunks in one call = C calls of `propagate`, outputs interleaved back to [3, C*n, S, D]; the
        one-keyframe chunk is rounded to the dtype its own call would have produced."""
        S, D = piv.shape[1:]
        res = residual.view(3, n_chunks, n, S, D) if residual is not None else None
        outs = []
        for j in range(n_chunks):
            single = first_single and j == 0
            ids = [slot0 + j] if single else [slot0 + j, slot0 + j - 1]
            r = res[:, j].reshape(3 * n, S, D) if res is not None else None
            dt = out_dtype
            if single:
                dt = kf_out.dtype if r is None else torch.promote_types(kf_out.dtype, r.dtype)
            o = self.propagate(tgt[j * n * S:(j + 1) * n * S], piv, inv_norm, ids, kf_out, None if single else w,
                               n, r, dt)
            outs.append(o.to(out_dtype).view(3, n, S, D))
        return self._with_norm(torch.stack(outs, dim=1).reshape(3 * n_chunks * n, S, D), norm)

    def ddim_step(self, x, eps, mu_a, sigma_a, mu_b, sigma_b, out=None):
        res = orc.ddim_step(x, eps, mu_a, sigma_a, mu_b, sigma_b)
        return res if out is None else out.copy_(res)

    def inject_copy_(self, x):
        self.calls.append(("inject_copy_", tuple(x.shape)))
        return orc.conv_inject_(x)
