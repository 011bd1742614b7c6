"""Seeded input generators shared by oracle/make_golden.py (which feeds them to
the verbatim reference) and by tests/ (which feed them to the oracle and to the
HIP path).  Test infrastructure.  Inputs are regenerated from seeds instead of
being stored; each fixture carries an input checksum so RNG drift is detected
instead of silently mis-compared."""
import zlib

import torch
import torch.nn as nn


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


def checksum(*tensors) -> float:
    return float(sum(t.detach().double().abs().sum() for t in tensors))


ATTN_CASES = {
    # name: (K, S, heads, d, schedule, t)
    "d40_noinj": (3, 48, 2, 40, [], 500),
    "d40_inj": (3, 48, 2, 40, [981, 961], 961),
    "d40_t1000": (2, 40, 2, 40, [], 1000),
    "d80_inj_tensor_sched": (2, 80, 2, 80, torch.tensor([981, 961]), 981),
    "d64_K13_perframe": (13, 16, 2, 64, [], 1),
    "d160_noinj": (2, 24, 1, 160, [981], 1),
    "d64_ragged": (2, 72, 2, 64, [5], 5),
}


def attn_inputs(name):
    """q, k, v [3K,S,D] fp32 holding bf16-representable values."""
    K, S, h, d, _, _ = ATTN_CASES[name]
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000 + 17 * K + S)
    D = h * d
    return tuple(bf16r(torch.randn(3 * K, S, D, generator=g)) for _ in range(3))


PROP_CASES = {
    # name: (K, n, S, D, dtype of kf_attn_output / residual stream)
    "n5_D320": (3, 5, 64, 320, torch.float32),
    "n4_D80_bf16": (4, 4, 48, 80, torch.bfloat16),
    "n2_D640": (2, 2, 40, 640, torch.float32),
    "n8_D64_videolike": (3, 8, 56, 64, torch.bfloat16),
}


def prop_inputs(name):
    """pivots [3,K,S,D] fp32, kf_attn_output [3K,S,D] (case dtype), and per chunk
    the hidden states [3n,S,D] (case dtype; norm1 = exact upcast to fp32)."""
    K, n, S, D, dt = PROP_CASES[name]
    g = torch.Generator().manual_seed(1000 + K * 7 + n * 3 + S + D)
    ln = nn.LayerNorm(D, elementwise_affine=False)
    piv = bf16r(ln(torch.randn(3, K, S, D, generator=g)))
    kf_out = bf16r(torch.randn(3 * K, S, D, generator=g)).to(dt)
    hidden = []
    for bi in range(K):
        if "videolike" in name:
            perm = torch.stack([torch.randperm(S, generator=g) for _ in range(n)])
            src = piv[0, bi][perm.reshape(-1)].reshape(n, S, D) + 0.1 * torch.randn(n, S, D, generator=g)
            tgt = torch.cat([src[None], torch.randn(2, n, S, D, generator=g)])
        else:
            tgt = ln(torch.randn(3, n, S, D, generator=g))
        hidden.append(bf16r(tgt).reshape(3 * n, S, D).to(dt))
    return piv, kf_out, hidden


BLOCKS_CFG = dict(dims=(80, 160, 320), heads=2, cross_dim=32, K=3, n=2, S=(48, 32, 16, 8),
                  seed=4321, schedule=[801, 781], conv_schedule=[801, 781, 761], n_chunks=3,
                  timesteps=(801, 761, 1))


def blocks_inputs(t):
    """Per timestep: encoder states and, in block execution order, the hidden
    inputs of the pivotal pass and of each chunk pass, plus the resnet inputs."""
    cfg = BLOCKS_CFG
    K, n = cfg["K"], cfg["n"]
    g = torch.Generator().manual_seed(cfg["seed"] + t)
    levels = [0, 0, 1, 1, 2, 2, 3, 2, 2, 2, 1, 1, 1, 0, 0, 0]

    def hid(b):
        return [torch.randn(3 * b, cfg["S"][l], cfg["dims"][min(l, 2)], generator=g) for l in levels]

    return dict(enc=torch.randn(3 * K, 7, cfg["cross_dim"], generator=g),
                enc_n=torch.randn(3 * n, 7, cfg["cross_dim"], generator=g),
                pivotal=hid(K), chunks=[hid(n) for _ in range(cfg["n_chunks"])],
                res_x=torch.randn(3 * n, cfg["dims"][2], 4, 4, generator=g),
                res_temb=torch.randn(3 * n, 16, generator=g))


ADAZERO_CFG = dict(dim=80, heads=2, cross_dim=32, K=3, n=2, S=24, seed=977, timestep=481.0)


def adazero_block():
    """A BasicTransformerBlock whose norm1 is an AdaLayerNormZero (use_ada_layer_norm_zero = True): the block
    type the reference gates with gate_msa in BOTH passes (tokenflow_utils.py:365-366)."""
    from tests import fake_diffusers as fd
    cfg = ADAZERO_CFG
    torch.manual_seed(cfg["seed"])
    blk = fd.BasicTransformerBlock(cfg["dim"], cfg["heads"], cross_dim=cfg["cross_dim"])
    blk.norm1 = fd.AdaLayerNormZero(cfg["dim"])
    blk.use_ada_layer_norm_zero = True
    return blk.eval()


def adazero_inputs():
    cfg = ADAZERO_CFG
    K, n, S, D = cfg["K"], cfg["n"], cfg["S"], cfg["dim"]
    g = torch.Generator().manual_seed(cfg["seed"] + 1)
    x_piv = torch.randn(3 * K, S, D, generator=g)
    chunks = []
    for c in range(K):       # video-like source branch: permuted keyframe tokens + noise (far from NN ties)
        perm = torch.randperm(S, generator=g)
        src = x_piv.view(3, K, S, D)[0, c][perm][None].repeat(n, 1, 1) + 0.05 * torch.randn(n, S, D, generator=g)
        chunks.append(torch.cat([src, torch.randn(2 * n, S, D, generator=g)]))
    return dict(pivotal=x_piv, chunks=chunks, enc=torch.randn(3 * K, 7, cfg["cross_dim"], generator=g),
                enc_n=torch.randn(3 * n, 7, cfg["cross_dim"], generator=g),
                timestep=torch.tensor([cfg["timestep"]]))


INVERSION_CFG = dict(F=6, H=8, W=8, steps=10, batch_size=4, seed=31)


class _Sample:
    def __init__(self, sample):
        self.sample = sample


class InversionModel:
    """The `self` that `Preprocess.ddim_inversion` / `ddim_sample` (preprocess.py:198-261) consume: a scheduler with
    `timesteps`, `alphas_cumprod`, `final_alpha_cumprod` (a 1000-step scaled-linear beta schedule, as SD's DDIM
    scheduler), a deterministic stand-in UNet and `sd_version`."""

    def __init__(self):
        cfg = INVERSION_CFG
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
        self.scheduler = type("Sched", (), {})()
        self.scheduler.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.scheduler.final_alpha_cumprod = torch.tensor(1.0)
        step = 1000 // cfg["steps"]
        self.scheduler.timesteps = (torch.arange(0, cfg["steps"]) * step).flip(0) + 1      # 901, 801, ..., 1
        self.sd_version = "1.5"
        g = torch.Generator().manual_seed(cfg["seed"])
        self.w = 0.3 * torch.randn(4, 4, generator=g)

    def unet(self, model_input, t, encoder_hidden_states=None):
        # a smooth, timestep- and prompt-dependent function of the latents (stands for the noise prediction)
        mixed = torch.einsum("oc,fchw->fohw", self.w.to(model_input.dtype), model_input)
        eps = torch.tanh(mixed + 0.1 * torch.roll(model_input, 1, dims=-1)) * (0.5 + float(t) / 2000.0)
        return _Sample(eps + encoder_hidden_states.mean().to(model_input.dtype) * 0.01)


def inversion_inputs(dtype=torch.float32):
    cfg = INVERSION_CFG
    g = torch.Generator().manual_seed(cfg["seed"] + 1)
    latents = torch.randn(cfg["F"], 4, cfg["H"], cfg["W"], generator=g).to(dtype)
    cond = torch.randn(1, 7, 16, generator=g).to(dtype)
    return latents, cond
