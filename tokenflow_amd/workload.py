"""Geometry and algorithmic-work accounting of the BASELINE.json configs.

A "step" of the hot path = for each of the 16 transformer blocks of the SD UNet
(tokenflow_utils.py:23-40): one extended attention over the 3K keyframe batch (pivotal
pass) + per frame chunk one NN search and one gather/blend/residual (propagation passes).
"""
from dataclasses import dataclass
from typing import List, Tuple

# UNet execution order of the 16 blocks: (level, is one of the 8 q/k-injected decoder blocks)
# down 2+2+2, mid 1, up_blocks[1] x3 (level 2), up_blocks[2] x3 (level 1), up_blocks[3] x3 (level 0);
# injected = up_blocks[1].attentions[1,2], up_blocks[2,3].attentions[0..2] (tokenflow_utils.py:208-214)
BLOCKS: List[Tuple[int, bool]] = (
    [(0, False)] * 2 + [(1, False)] * 2 + [(2, False)] * 2 + [(3, False)]
    + [(2, False), (2, True), (2, True)] + [(1, True)] * 3 + [(0, True)] * 3)


@dataclass(frozen=True)
class Config:
    name: str
    frames: int          # F
    chunk: int           # n = batch_size = frames per chunk; K = C = F / n keyframes
    levels: Tuple[Tuple[int, int, int], ...]   # per level (S, D, heads)
    pnp: bool = True     # False = SDEdit variant (no injection)

    @property
    def K(self):
        return self.frames // self.chunk


def _sd15(res):
    s = (res // 8) ** 2
    return ((s, 320, 8), (s // 4, 640, 8), (s // 16, 1280, 8), (s // 64, 1280, 8))


def _sd21(res):
    s = (res // 8) ** 2
    return ((s, 320, 5), (s // 4, 640, 10), (s // 16, 1280, 20), (s // 64, 1280, 20))


CONFIGS = {
    "cfg1": Config("8f 256x256 SD1.5 PnP, 4 keyframes", 8, 2, _sd15(256)),
    "cfg2": Config("40f 512x512 SD1.5 PnP, 8 keyframes", 40, 5, _sd15(512)),
    "cfg4": Config("80f 768x768 SD2.1 PnP, 10 keyframes", 80, 8, _sd21(768)),
    "cfg5": Config("200f 512x512 SD2.1 SDEdit, 25 keyframes", 200, 8, _sd21(512), pnp=False),
}


def attn_flops(K, S, D):
    """4*K*S*D*(S + 2*K*S): QK^T and PV at 2 flop/MAC; source attends S keys, uncond and cond K*S."""
    return 4.0 * K * S * D * (S + 2.0 * K * S)


def nn_flops(n, S, D, P):
    return 2.0 * n * S * S * D * P


def gather_bytes(n, S, D, P, in_bytes=2, res_bytes=2, out_bytes=4):
    """P source rows read + residual read + output write, per chunk and block (3 branches)."""
    rows = 3.0 * n * S * D
    return rows * (P * in_bytes + res_bytes + out_bytes) + P * n * S * 4


def step_work(cfg: Config):
    """Algorithmic totals per step: (attention flops, NN flops, gather bytes)."""
    K, n, C = cfg.K, cfg.chunk, cfg.K
    fa = fn = gb = 0.0
    for lvl, _ in BLOCKS:
        S, D, _h = cfg.levels[lvl]
        fa += attn_flops(K, S, D)
        fn += nn_flops(n, S, D, 1) + (C - 1) * nn_flops(n, S, D, 2)
        gb += gather_bytes(n, S, D, 1, out_bytes=2) + (C - 1) * gather_bytes(n, S, D, 2)
    return fa, fn, gb


MFMA_BF16_PEAK = 2.5e15      # dense, MI355X_MICROARCH.md
HBM_PEAK = 8.0e12
