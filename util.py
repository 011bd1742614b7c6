"""Drop-in replacement for omerbt/TokenFlow's `util.py` (the names the run scripts import:
run_tokenflow_pnp.py:17 `from util import save_video, seed_everything`).

Only `isinstance_str` and `batch_cosine_sim` belong to the hot path (they live in
tokenflow_amd/hooks.py).  The media helpers are out of scope: they
are thin wrappers that import torchvision / PIL lazily and raise ImportError where the
package is absent, so that importing this module never fails.
"""
import os
import random

import numpy as np
import torch

from tokenflow_amd.hooks import batch_cosine_sim, isinstance_str  # noqa: F401


def seed_everything(seed):
    """util.py:99-103."""
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)


def save_video(raw_frames, save_path, fps=10):
    """util.py:88-96: [F,3,H,W] floats in [0,1] -> H.264 file (needs torchvision)."""
    from torchvision.io import write_video
    frames = (raw_frames * 255).to(torch.uint8).cpu().permute(0, 2, 3, 1)
    write_video(save_path, frames, fps=fps, video_codec="libx264", options={"crf": "18", "preset": "slow"})


def load_imgs(data_path, n_frames, device="cuda", pil=False):
    """util.py:72-85: %05d.jpg / .png frames -> [F,3,H,W] tensor (needs PIL + torchvision)."""
    from PIL import Image
    import torchvision.transforms as T
    pils = []
    for i in range(n_frames):
        path = os.path.join(data_path, "%05d.jpg" % i)
        if not os.path.exists(path):
            path = os.path.join(data_path, "%05d.png" % i)
        pils.append(Image.open(path))
    imgs = torch.cat([T.ToTensor()(p).unsqueeze(0) for p in pils]).to(device)
    return (imgs, pils) if pil else imgs


def add_dict_to_yaml_file(file_path, key, value):
    """util.py:31-44."""
    import yaml
    data = {}
    if os.path.exists(file_path):
        with open(file_path, "r") as f:
            data = yaml.safe_load(f)
    data[key] = value
    with open(file_path, "w") as f:
        yaml.dump(data, f)
