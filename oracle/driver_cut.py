"""The hook-facing methods of the VERBATIM reference drivers as plain functions (test infrastructure).

`run_tokenflow_pnp.py` / `run_tokenflow_sdedit.py` import diffusers, transformers and torchvision at module level and
cannot be imported here, so `TokenFlow.init_method`, `TokenFlow.denoise_step` and `TokenFlow.batched_denoise_step`
(run_tokenflow_pnp.py:195-239, run_tokenflow_sdedit.py:154-193) are cut out of the file's syntax tree and compiled
UNCHANGED -- decorators (`@torch.no_grad()`, `@torch.autocast(dtype=torch.float16, device_type='cuda')`) included --
in a namespace that holds what their bodies name: `torch` and the hook functions the driver imports from
`tokenflow_utils` (run_tokenflow_pnp.py:17-18).  Which `tokenflow_utils` that is -- the reference's own or this
repository's drop-in -- is the caller's choice: that is the seam under test.

Only usable where /root/reference is mounted (the build container).  Nothing of the reference is copied into the
repository: the source is read, compiled and executed where it lies.
"""
import ast
import os

import torch

from oracle import ref_loader

SCRIPTS = {"pnp": "run_tokenflow_pnp.py", "sdedit": "run_tokenflow_sdedit.py"}
METHODS = ("init_method", "denoise_step", "batched_denoise_step")


def load_reference_driver(kind, hook_ns):
    """{method name: function} of the reference's TokenFlow class, bodies unchanged; `hook_ns` maps the hook names
    the bodies call to the implementations to run them over."""
    script = SCRIPTS[kind]
    src = open(os.path.join(ref_loader.REF_ROOT, script)).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TokenFlow")
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in METHODS]
    assert sorted(f.name for f in fns) == sorted(METHODS), [f.name for f in fns]
    ns = {"torch": torch}
    ns.update(hook_ns)
    exec(compile(ast.Module(body=fns, type_ignores=[]), script, "exec"), ns)
    return {m: ns[m] for m in METHODS}
