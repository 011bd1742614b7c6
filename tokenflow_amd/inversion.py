"""The step before the hot path (row f4 of SURVEY.md section 8): DDIM inversion / reconstruction loops and the
latents directory they write -- `preprocess.py:198-261, 305-314` of omerbt/TokenFlow -- with the latent update of
every UNet call as ONE HIP launch (`tf_ddim_step`) instead of six elementwise torch ops.

Same function names, argument meaning and file format as the reference's `Preprocess.ddim_inversion` /
`Preprocess.ddim_sample`; they take the reference's `self` as first argument (anything with `.scheduler`
[`timesteps`, `alphas_cumprod`, `final_alpha_cumprod`], `.unet` and `.sd_version`; `.depth_maps`,
`.controlnet_pred`, `.canny_cond` where those variants are used), so a maintainer binds them with
`Preprocess.ddim_inversion = tokenflow_amd.inversion.ddim_inversion`.  The UNet, VAE and text encoder are
diffusers' and out of scope; what is owned here is the update, the loop and the on-disk format
(`<save_path>/latents/noisy_latents_<t>.pt` = `torch.save` of the `[F,4,H/8,W/8]` tensor, which
`load_source_latents_t` of the hook layer reads back).
"""
import os
from pathlib import Path

import torch

from . import ops

__all__ = ["ddim_inversion", "ddim_sample", "latents_save_path"]


def latents_save_path(save_dir, sd_version, data_path, steps, n_frames) -> str:
    """preprocess.py:305-309: <save_dir>/sd_<version>/<video stem>/steps_<N>/nframes_<F>."""
    return os.path.join(save_dir, f"sd_{sd_version}", Path(data_path).stem, f"steps_{steps}", f"nframes_{n_frames}")


def _coeffs(scheduler, t, t_other):
    """(mu, sigma) pairs of timestep t and of the neighbouring one (preprocess.py:211-220 / 245-254), as Python
    floats of the fp32 values the reference computes."""
    a_t = scheduler.alphas_cumprod[t]
    a_o = scheduler.alphas_cumprod[t_other] if t_other is not None else scheduler.final_alpha_cumprod
    a_t, a_o = torch.as_tensor(a_t, dtype=torch.float32), torch.as_tensor(a_o, dtype=torch.float32)
    return (float(a_t ** 0.5), float((1 - a_t) ** 0.5)), (float(a_o ** 0.5), float((1 - a_o) ** 0.5))


def _eps(self, x_batch, t, cond, b, batch_size):
    """The noise prediction of one batch (preprocess.py:204-209, 222-223): plain UNet, depth-conditioned input, or
    the ControlNet path."""
    cond_batch = cond.repeat(x_batch.shape[0], 1, 1)
    model_input = x_batch
    if self.sd_version == "depth":
        depth_maps = torch.cat([self.depth_maps[b: b + batch_size]])
        model_input = torch.cat([x_batch, depth_maps], dim=1)
    if self.sd_version != "ControlNet":
        return self.unet(model_input, t, encoder_hidden_states=cond_batch).sample
    return self.controlnet_pred(x_batch, t, cond_batch, torch.cat([self.canny_cond[b: b + batch_size]]))


@torch.no_grad()
def ddim_inversion(self, cond, latent_frames, save_path, batch_size, save_latents=True, timesteps_to_save=None):
    """preprocess.py:198-230.  `latent_frames` is updated in place and saved after every timestep in
    `timesteps_to_save` (all by default), and once more at the end, as the reference does."""
    timesteps = reversed(self.scheduler.timesteps)
    timesteps_to_save = timesteps_to_save if timesteps_to_save is not None else timesteps
    for i, t in enumerate(timesteps):
        (mu, sigma), (mu_prev, sigma_prev) = _coeffs(self.scheduler, t, timesteps[i - 1] if i > 0 else None)
        for b in range(0, latent_frames.shape[0], batch_size):
            x_batch = latent_frames[b:b + batch_size]
            eps = _eps(self, x_batch, t, cond, b, batch_size)
            # pred_x0 = (x - sigma_prev*eps) / mu_prev;  x <- mu*pred_x0 + sigma*eps   (224-225), in place
            ops.ddim_step(x_batch, eps.to(x_batch.dtype).contiguous(), mu_prev, sigma_prev, mu, sigma, out=x_batch)
        if save_latents and t in timesteps_to_save:
            torch.save(latent_frames, os.path.join(save_path, "latents", f"noisy_latents_{t}.pt"))
    torch.save(latent_frames, os.path.join(save_path, "latents", f"noisy_latents_{t}.pt"))
    return latent_frames


@torch.no_grad()
def ddim_sample(self, x, cond, batch_size):
    """preprocess.py:232-261: the reconstruction loop (the inversion's sanity check)."""
    timesteps = self.scheduler.timesteps
    for i, t in enumerate(timesteps):
        (mu, sigma), (mu_prev, sigma_prev) = _coeffs(self.scheduler, t,
                                                      timesteps[i + 1] if i < len(timesteps) - 1 else None)
        for b in range(0, x.shape[0], batch_size):
            x_batch = x[b:b + batch_size]
            eps = _eps(self, x_batch, t, cond, b, batch_size)
            # pred_x0 = (x - sigma*eps) / mu;  x <- mu_prev*pred_x0 + sigma_prev*eps   (259-260), in place
            ops.ddim_step(x_batch, eps.to(x_batch.dtype).contiguous(), mu, sigma, mu_prev, sigma_prev, out=x_batch)
    return x
