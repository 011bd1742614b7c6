"""Import the VERBATIM reference hook layer from /root/reference (test infrastructure).

`tokenflow_utils.py` / `util.py` of the reference are pure Python on torch; the
only obstacle is util.py:7-16 importing torchvision / kornia / cv2, which are
absent here.  They are never used on the hot path, so nine stub entries in
`sys.modules` are enough.  The reference modules are loaded
under private names (`_ref_tokenflow_utils`, `_ref_util`) so they never shadow
this repo's drop-in modules of the same name.

Only usable where /root/reference exists (the build container); the GPU box
does not have it, which is why `make_golden.py` commits fixtures.
"""
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("TOKENFLOW_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "tokenflow_utils.py"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def load():
    """Returns (ref_tokenflow_utils, ref_util) modules."""
    if "_ref_tokenflow_utils" in sys.modules:
        return sys.modules["_ref_tokenflow_utils"], sys.modules["_ref_util"]
    if not available():
        raise FileNotFoundError(f"reference not mounted at {REF_ROOT}")
    sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only mount
    stubs = {
        "torchvision": _stub("torchvision"),
        "torchvision.transforms": _stub("torchvision.transforms"),
        "torchvision.io": _stub("torchvision.io", read_video=None, write_video=None),
        "kornia": _stub("kornia"),
        "kornia.geometry": _stub("kornia.geometry"),
        "kornia.geometry.transform": _stub("kornia.geometry.transform", remap=None),
        "kornia.utils": _stub("kornia.utils"),
        "kornia.utils.grid": _stub("kornia.utils.grid", create_meshgrid=None),
        "cv2": _stub("cv2"),
    }
    saved = {k: sys.modules.get(k) for k in list(stubs) + ["util", "tokenflow_utils"]}
    sys.modules.update(stubs)
    try:
        def _load(private, fname, public):
            spec = importlib.util.spec_from_file_location(private, os.path.join(REF_ROOT, fname))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[public] = mod          # `from util import ...` inside the reference
            spec.loader.exec_module(mod)
            sys.modules[private] = mod
            return mod
        ref_util = _load("_ref_util", "util.py", "util")
        ref_tfu = _load("_ref_tokenflow_utils", "tokenflow_utils.py", "tokenflow_utils")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ref_tfu, ref_util
