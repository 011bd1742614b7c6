"""Harness for the DRIVER <-> HOOK seam (test infrastructure).

The drop-in contract is that the reference's drivers run unmodified over this repository's `tokenflow_utils`.  The
drivers themselves cannot run here or on the GPU box (no diffusers, no weights, no video decoder), but the part of
them that touches the hooks can: `TokenFlow.init_method`, `TokenFlow.denoise_step` and
`TokenFlow.batched_denoise_step` (/root/reference/run_tokenflow_pnp.py:195-239 and
run_tokenflow_sdedit.py:154-193) only need a `self` with `unet`, `scheduler`, `config`, `latents_path`,
`text_embeds`, `pnp_guidance_embeds`, `sd_version`.  This module provides that `self` (`SeamPipe`, on the runnable
stand-in UNet of tests/fake_diffusers.py), a tracer for the hook calls the driver makes, recorders for what the UNet
and its 16 transformer blocks return, and a restatement of the three methods for machines without /root/reference.

Where /root/reference is mounted the methods are cut out of the reference's syntax tree and executed UNCHANGED
(`oracle/driver_cut.py`); `oracle/make_golden.py` runs them over the verbatim reference hooks on CPU fp32 and
writes tests/golden/driver.pt.  On the GPU box the restatement below drives the hooks;
tests/test_driver_seam.py::test_restated_driver_equals_the_verbatim_cut proves (in the build container) that the
restatement and the cut issue the same hook calls, draw the same `pivotal_idx` and return the same bits.
"""
import os

import torch
import torch.nn as nn

from oracle.golden_util import digest
from tests import fake_diffusers as fd

HOOK_NAMES = ("load_source_latents_t", "register_time", "register_pivotal", "register_batch_idx",
              "register_extended_attention_pnp", "register_extended_attention", "register_conv_injection",
              "set_tokenflow")

CFG = dict(F=8, batch_size=2, H=32, W=32, dims=(80, 160, 320), heads=2, cross=32, seed=777, n_steps=3,
           guidance_scale=7.5, qk_injection_t=1, conv_injection_t=2, draw_seed=2024, block_stride=997)
# timesteps 951, 901, 851: q/k + feature injection / feature injection only / none (run_tokenflow_pnp.py:235-237)


class SeamPipe(nn.Module):
    """The `self` of the reference's TokenFlow methods (run_tokenflow_pnp.py:25-68): an nn.Module holding the UNet
    next to other sub-modules, the scheduler, the config dict and the embeddings."""

    def __init__(self, latents_path, device="cpu"):
        super().__init__()
        cfg = CFG
        torch.manual_seed(cfg["seed"])
        self.unet = fd.RunnableUNet(dims=cfg["dims"], heads=cfg["heads"], cross_dim=cfg["cross"]).eval()
        self.text_encoder = nn.Linear(4, 4)
        self.vae = nn.Conv2d(4, 4, 1)
        g = torch.Generator().manual_seed(cfg["seed"] + 1)
        self.text_embeds = torch.randn(2, 7, cfg["cross"], generator=g).to(device)
        self.pnp_guidance_embeds = torch.randn(1, 7, cfg["cross"], generator=g).to(device)
        self.scheduler = fd.FakeDDIMScheduler(cfg["n_steps"])
        self.config = {"batch_size": cfg["batch_size"], "guidance_scale": cfg["guidance_scale"],
                       "n_frames": cfg["F"], "device": device}
        self.sd_version = "1.5"
        self.latents_path = latents_path
        self.to(device)


def seam_latents(device="cpu"):
    """(x0 [F,4,H,W], {t: source latents [F,4,H,W]}): video-like -- every frame is one base latent plus a small
    perturbation, so that nearest-neighbour fields have clear winners (as between real frames)."""
    cfg = CFG
    g = torch.Generator().manual_seed(cfg["seed"] + 2)
    shape = (cfg["F"], 4, cfg["H"], cfg["W"])
    base = torch.randn(1, *shape[1:], generator=g)
    x0 = base + 0.1 * torch.randn(*shape, generator=g)
    src = {}
    for t in fd.FakeDDIMScheduler(cfg["n_steps"]).timesteps.tolist():
        src[t] = (base + 0.1 * torch.randn(*shape, generator=g)).to(device)
    return x0.to(device), src


def write_latents(path, src):
    """`noisy_latents_{t}.pt`, the directory format `load_source_latents_t` reads (tokenflow_utils.py:43-47).  Saved
    from the device the driver runs on, as preprocess.py saves them from the GPU."""
    os.makedirs(path, exist_ok=True)
    for t, v in src.items():
        torch.save(v, os.path.join(path, f"noisy_latents_{t}.pt"))


def traced(hook_module, log):
    """name -> wrapper of `hook_module.name` that appends (name, argument summary) to `log`."""
    def summary(name, a):
        if name == "load_source_latents_t":
            return (type(a[0]).__name__, int(a[0]))
        if name == "register_time":
            return (type(a[1]).__name__, a[1])
        if name in ("register_pivotal", "register_batch_idx"):
            return (type(a[1]).__name__, a[1])
        if name in ("register_extended_attention_pnp", "register_conv_injection"):
            return tuple(int(s) for s in a[1])
        return ()

    def wrap(name):
        fn = getattr(hook_module, name)

        def w(*a, **kw):
            log.append((name,) + summary(name, a))
            return fn(*a, **kw)
        return w
    return {n: wrap(n) for n in HOOK_NAMES}


def restated_driver(kind, ns):
    """The three driver methods in this repository's own words, for machines where /root/reference is absent;
    `ns` = the hook functions they call (by the names the reference imports, run_tokenflow_pnp.py:17-18).
    kind "pnp": run_tokenflow_pnp.py:195-239; kind "sdedit": run_tokenflow_sdedit.py:154-193."""

    @torch.no_grad()
    def denoise_step(self, x, t, indices):
        # :198-199 source latents of these frames in front of two copies of x (source | uncond | cond)
        unet_in = torch.cat([ns["load_source_latents_t"](t, self.latents_path)[indices], x, x])
        ns["register_time"](self, t.item())                                                     # :203
        n = len(indices)
        prompts = torch.cat([self.pnp_guidance_embeds.repeat(n, 1, 1),                          # :206-207
                             torch.repeat_interleave(self.text_embeds, n, dim=0)])
        eps = self.unet(unet_in, t, encoder_hidden_states=prompts)["sample"]                    # :210
        _, eps_u, eps_c = eps.chunk(3)                                                          # :213-214
        eps = eps_u + self.config["guidance_scale"] * (eps_c - eps_u)
        return self.scheduler.step(eps, t, x)["prev_sample"]                                    # :217

    @torch.autocast(dtype=torch.float16, device_type="cuda")                                    # :220
    def batched_denoise_step(self, x, t, indices):
        bs = self.config["batch_size"]
        # :224 one keyframe per chunk, drawn from the global CPU generator
        pivotal_idx = torch.randint(bs, (len(x) // bs,)) + torch.arange(0, len(x), bs)
        ns["register_pivotal"](self, True)                                                      # :226-228
        self.denoise_step(x[pivotal_idx], t, indices[pivotal_idx])
        ns["register_pivotal"](self, False)
        out = []
        for i, b in enumerate(range(0, len(x), bs)):                                            # :229-231
            ns["register_batch_idx"](self, i)
            out.append(self.denoise_step(x[b:b + bs], t, indices[b:b + bs]))
        return torch.cat(out)

    if kind == "pnp":
        def init_method(self, conv_injection_t, qk_injection_t):                                # :235-239
            ts = self.scheduler.timesteps
            self.qk_injection_timesteps = ts[:qk_injection_t] if qk_injection_t >= 0 else []
            self.conv_injection_timesteps = ts[:conv_injection_t] if conv_injection_t >= 0 else []
            ns["register_extended_attention_pnp"](self, self.qk_injection_timesteps)
            ns["register_conv_injection"](self, self.conv_injection_timesteps)
            ns["set_tokenflow"](self.unet)
    else:
        def init_method(self):                                                                  # sdedit :191-193
            ns["register_extended_attention"](self)
            ns["set_tokenflow"](self.unet)
    return dict(denoise_step=denoise_step, batched_denoise_step=batched_denoise_step, init_method=init_method)


def run_driver(kind, methods, latents_dir, device="cpu", model_autocast=True, keep_tensors=False):
    """Install the hooks through the driver's own `init_method`, then run the sampling loop of
    run_tokenflow_pnp.py:246-247 (`for t in scheduler.timesteps: x = batched_denoise_step(x, t, indices)`) for
    CFG['n_steps'] timesteps.  Returns a record: per step the `indices` argument of every `denoise_step` call (the
    first one is `pivotal_idx`), what the UNet returned per call, what each of the 16 transformer blocks returned
    per call, the latents after the step.
    model_autocast=False: the UNet's own layers run outside the driver's `torch.autocast` (fp32 model: the only
    16-bit roundings left are the kernels'); the driver code is untouched either way."""
    cfg = CFG
    x, src = seam_latents(device)
    write_latents(latents_dir, src)
    cls = type("TokenFlow", (SeamPipe,), dict(methods))
    pipe = cls(latents_dir, device)
    if kind == "pnp":
        pipe.init_method(conv_injection_t=cfg["conv_injection_t"], qk_injection_t=cfg["qk_injection_t"])
    else:
        pipe.init_method()
    if not model_autocast:
        inner = pipe.unet.forward

        def fp32_forward(*a, **kw):
            with torch.autocast("cuda", enabled=False):
                return inner(*a, **kw)
        pipe.unet.forward = fp32_forward
    keep = (lambda t, s: t.detach().float().cpu()) if keep_tensors else (lambda t, s: digest(t, s))
    calls = []                      # per denoise_step call: dict(indices, unet, blocks)
    blocks = [b for _, b in pipe.unet.transformer_blocks_in_order()]
    for blk in blocks:
        blk.register_forward_hook(lambda m, a, out: calls[-1]["blocks"].append(keep(out, cfg["block_stride"])))
    pipe.unet.register_forward_hook(lambda m, a, out: calls[-1].__setitem__("unet", keep(out["sample"], 7)))
    bound = pipe.denoise_step

    def spy_denoise_step(x_, t_, indices_):
        calls.append(dict(indices=indices_.clone(), blocks=[]))
        return bound(x_, t_, indices_)
    pipe.denoise_step = spy_denoise_step          # `self.denoise_step(...)` inside batched_denoise_step lands here
    torch.manual_seed(cfg["draw_seed"])           # seed_everything(config["seed"]) of the driver (util.py:99-103)
    indices = torch.arange(cfg["F"])
    steps = []
    for t in pipe.scheduler.timesteps:
        calls.clear()
        x = pipe.batched_denoise_step(x, t, indices)
        steps.append(dict(t=int(t), calls=list(calls), x=keep(x, 3)))
    return dict(steps=steps, weights_checksum=float(sum(p.detach().double().abs().sum() for p in pipe.parameters())))
