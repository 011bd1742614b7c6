"""ctypes binding of libtokenflow_hip.so (C ABI: include/tokenflow_hip.h)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TOKENFLOW_HIP_LIB overrides the library path (kernel A/B experiments, tools/attn_microbench.py)
LIB_PATH = os.environ.get("TOKENFLOW_HIP_LIB") or os.path.join(_HERE, "libtokenflow_hip.so")

TF_BF16, TF_F16, TF_F32 = 0, 1, 2
TF_ATTN_INJECT, TF_ATTN_EXACT_SCALE, TF_ATTN_BANK_ONLY, TF_ATTN_SOURCE_ONLY, TF_ATTN_NO_SPLIT = 1, 2, 4, 8, 16
TF_ATTN_OUT_F32 = 32
TF_ATTN_FOLD_SCALE = 64
TF_ATTN_NO_FUSED, TF_ATTN_FUSED = 128, 1 << 17
TF_ATTN_HINT_QB2, TF_ATTN_PRECISE_P, TF_ATTN_NO_PRECISE_P = 1 << 14, 1 << 15, 1 << 16
TF_ATTN_HINT_MIX = 1 << 18   # Dh = 40 streaming kernel: the mixed-MFMA-shape form whatever the launch size


def attn_hint(qw: int = 0, kw: int = 0, qb: int = 1) -> int:
    """TF_ATTN_HINT_QW / _KW bits of the fused small-problem kernel: qw in {0 (auto), 1 (the wave-private form), 2, 4}
    query waves per workgroup, kw in {0 (auto), 1, 2, 4, 8} key groups."""
    code = {0: 0, 1: 1, 2: 2, 4: 3, 8: 4}
    return (code[qw] << 8) | (code[kw] << 11) | (TF_ATTN_HINT_QB2 if qb == 2 else 0)


ABI_VERSION = 7
TF_RANK_HEADS, TF_RANK_BANK, TF_RANK_SLOTS, TF_RANK_NO_HALO, TF_RANK_INV_NORM = 0, 1, 64, 16, 32
TF_ERR_COMM = -6

_c = ctypes
_SIGNATURES = {
    "tf_abi_version": (_c.c_int, []),
    "tf_last_error": (_c.c_char_p, []),
    "tf_ext_attn_workspace_bytes": (_c.c_size_t, [_c.c_int] * 5),
    "tf_ext_attn_fwd": (_c.c_int, [_c.c_void_p] * 4 + [_c.c_int] * 6 + [_c.c_int64, _c.c_float, _c.c_int, _c.c_int,
                                   _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "tf_ext_attn_fwd_strided": (_c.c_int, [_c.c_void_p] * 4 + [_c.c_int] * 6 + [_c.c_int64, _c.c_void_p, _c.c_float,
                                           _c.c_int, _c.c_int, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "tf_head_pack": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p] + [_c.c_int] * 4 + [_c.c_int64, _c.c_int,
                                _c.c_void_p]),
    "tf_head_unpack": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p] + [_c.c_int] * 5 + [_c.c_int64, _c.c_int,
                                  _c.c_void_p]),
    "tf_pivot_inv_norm": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_int, _c.c_void_p]),
    "tf_nn_search_workspace_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int, _c.c_int, _c.c_int]),
    "tf_nn_search": (_c.c_int, [_c.c_void_p] * 4 + [_c.c_int64] + [_c.c_int] * 6 + [_c.c_void_p, _c.c_size_t,
                                _c.c_void_p]),
    "tf_gather_blend": (_c.c_int, [_c.c_void_p] * 5 + [_c.c_int] * 10 + [_c.c_void_p]),
    "tf_nn_gather_blend_workspace_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int, _c.c_int, _c.c_int]),
    "tf_nn_gather_blend": (_c.c_int, [_c.c_void_p] * 7 + [_c.c_int] * 11 + [_c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "tf_nn_gather_blend_chunks_workspace_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int, _c.c_int, _c.c_int]),
    "tf_nn_gather_blend_chunks": (_c.c_int, [_c.c_void_p] * 7 + [_c.c_int] * 12 + [_c.c_void_p, _c.c_size_t,
                                             _c.c_void_p]),
    "tf_nn_gather_blend_norm": (_c.c_int, [_c.c_void_p] * 7 + [_c.c_int] * 11 + [_c.c_void_p, _c.c_void_p, _c.c_float,
                                           _c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "tf_nn_gather_blend_chunks_norm": (_c.c_int, [_c.c_void_p] * 7 + [_c.c_int] * 12 + [_c.c_void_p, _c.c_void_p,
                                                  _c.c_float, _c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p,
                                                  _c.c_size_t, _c.c_void_p]),
    "tf_layer_norm": (_c.c_int, [_c.c_void_p] * 5 + [_c.c_int64, _c.c_int, _c.c_float] + [_c.c_int] * 3 + [_c.c_void_p]),
    "tf_add_layer_norm": (_c.c_int, [_c.c_void_p] * 6 + [_c.c_int64, _c.c_int, _c.c_float] + [_c.c_int] * 5 + [_c.c_void_p]),
    "tf_ddim_step": (_c.c_int, [_c.c_void_p] * 3 + [_c.c_int64] + [_c.c_float] * 4 + [_c.c_int, _c.c_void_p]),
    "tf_inject_copy": (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int, _c.c_void_p]),
    # multi-GPU exchange steps over RCCL (tokenflow_amd/comm.py; the sharded host path uses torch.distributed instead)
    "tf_comm_available": (_c.c_int, []),
    "tf_comm_unique_id": (_c.c_int, [_c.c_void_p]),
    "tf_comm_init": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p]),
    "tf_comm_destroy": (_c.c_int, [_c.c_void_p]),
    "tf_comm_rank": (_c.c_int, [_c.c_void_p]),
    "tf_comm_world": (_c.c_int, [_c.c_void_p]),
    "tf_allgather_kv": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_void_p]),
    "tf_allgather_rows": (_c.c_int, [_c.c_void_p] * 4 + [_c.c_int64, _c.c_int, _c.c_void_p]),
    "tf_all_to_all_rows": (_c.c_int, [_c.c_void_p] * 5 + [_c.c_int64, _c.c_int, _c.c_void_p]),
    "tf_sendrecv_pivot": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p,
                                     _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    "tf_comm_init_hooks": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p]),
    "tf_comm_init_loopback": (_c.c_int, [_c.c_int, _c.c_int, _c.c_void_p]),
    "tf_comm_loopback_copies": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "tf_comm_loopback_wire": (_c.c_int, [_c.c_void_p, _c.c_double, _c.c_double]),
    # one rank's pivotal pass of a block in one call (csrc/rank_exec.hip; tokenflow_amd/sharded.py NativeShard)
    "tf_rank_create": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p]),
    "tf_rank_destroy": (_c.c_int, [_c.c_void_p]),
    "tf_rank_local_keyframes": (_c.c_int, [_c.c_void_p]),
    "tf_rank_first_keyframe": (_c.c_int, [_c.c_void_p]),
    "tf_rank_pivotal_workspace_bytes": (_c.c_size_t, [_c.c_void_p] + [_c.c_int] * 4),
    "tf_rank_pivotal": (_c.c_int, [_c.c_void_p] * 8 + [_c.c_int] * 3 + [_c.c_float] + [_c.c_int] * 4 +
                        [_c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "tf_rank_halo_wait": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_void_p]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


class TokenflowHipError(RuntimeError):
    pass


def load():
    """Load the library (idempotent).  torch must be imported first so that the
    library's libamdhip64 dependency resolves to the runtime torch already mapped."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (maps libamdhip64 before dlopen)
    if not os.path.isfile(LIB_PATH):
        raise TokenflowHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C tokenflow_amd/csrc`.  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.tf_abi_version() != ABI_VERSION:
        raise TokenflowHipError(f"ABI version {lib.tf_abi_version()} != {ABI_VERSION}: rebuild the library")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().tf_last_error().decode(errors="replace")
        raise TokenflowHipError(f"{what} failed (rc={rc}): {msg}")
