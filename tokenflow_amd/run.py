"""Launcher that runs an UNMODIFIED reference driver on the MI355X hook layer:

    python -m tokenflow_amd.run /path/to/TokenFlow/run_tokenflow_pnp.py --config_path configs/config_pnp.yaml

Why a launcher: `python script.py` puts the SCRIPT'S directory at sys.path[0], ahead of PYTHONPATH, so inside the
reference checkout `from tokenflow_utils import *` / `from util import ...` (run_tokenflow_pnp.py:16-17) resolve to
the reference's own pure-torch modules whatever PYTHONPATH says -- silently, the run just uses the slow path.  Here
this repository's drop-in modules `tokenflow_utils` and `util` are imported FIRST (and verified to be ours), then
the script runs as `__main__` with its own directory next on the path, exactly as `python script.py` would see it
(working directory untouched: the reference reads configs/ and data/ relative to it).
"""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _import_dropins():
    if REPO in sys.path:
        sys.path.remove(REPO)
    sys.path.insert(0, REPO)
    for name in ("tokenflow_utils", "util"):
        stale = sys.modules.get(name)
        if stale is not None and os.path.dirname(os.path.abspath(getattr(stale, "__file__", ""))) != REPO:
            del sys.modules[name]
    import tokenflow_utils
    import util
    for mod in (tokenflow_utils, util):
        if os.path.dirname(os.path.abspath(mod.__file__)) != REPO:
            raise ImportError(f"{mod.__name__} resolved to {mod.__file__}, not to {REPO}: the HIP hook layer would "
                              f"not be used")
    return tokenflow_utils, util


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 2
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        print(f"tokenflow_amd.run: no such script: {script}", file=sys.stderr)
        return 2
    _import_dropins()                       # cached in sys.modules: the script's imports get these
    script_dir = os.path.dirname(script)
    if script_dir in sys.path:
        sys.path.remove(script_dir)
    sys.path.insert(1, script_dir)          # the script's other local imports (preprocess, ...) still resolve
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
