"""tokenflow_amd -- MI355X-native implementation of TokenFlow's per-step hot path.

Only what the path needs lives here:
  csrc/       hand-written gfx950 HIP kernels + the C ABI (include/tokenflow_hip.h)
  _lib.py     ctypes loader of libtokenflow_hip.so (fails loudly when missing)
  ops.py      torch-tensor wrappers over the C ABI (device memory + stream plumbing only)
  hooks.py    the reference's hook API (register_* / set_tokenflow), same names and semantics
  sharded.py  frame-sharded multi-GPU step (torch.distributed = RCCL over xGMI)
  workload.py geometry / algorithmic-work formulas of the BASELINE configs

There is NO CPU or eager-PyTorch fallback anywhere in this package: an op raises if the
HIP library is missing or a tensor is not on a GPU.
"""
__version__ = "0.1.0"
