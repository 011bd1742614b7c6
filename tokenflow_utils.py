"""Drop-in replacement for omerbt/TokenFlow's `tokenflow_utils.py`.

`run_tokenflow_pnp.py` / `run_tokenflow_sdedit.py` of the reference do
`from tokenflow_utils import *` (run_tokenflow_pnp.py:16): put this repository first on
PYTHONPATH and they pick up the MI355X implementation unchanged.
The implementation lives in tokenflow_amd/hooks.py.
"""
from tokenflow_amd.hooks import (  # noqa: F401
    batch_cosine_sim, isinstance_str, load_source_latents_t, make_tokenflow_attention_block,
    register_batch_idx, register_conv_injection, register_extended_attention,
    register_extended_attention_pnp, register_frame_shard, register_pivotal, register_time, set_tokenflow)
# register_frame_shard: multi-GPU extension (one process per GPU), not part of the reference's surface

# `from tokenflow_utils import *` in the reference also leaks these two names (its module does
# `import torch, os` at top level, tokenflow_utils.py:2-3); keep that surface identical.
import os  # noqa: F401,E402
import torch  # noqa: F401,E402
