"""Run the REFERENCE's own unit tests for the hot path against THIS package (build container only: needs /root/reference).

The test files are executed where they lie under /root/reference/tests -- nothing is copied.  `connectomics.*` imports are
redirected to `pytorch_connectomics_amd.*` by a meta-path alias; modules of the reference this package has no counterpart for
(config schema, data pipeline, decoding) are not aliased, so a test file that needs them fails at import and is reported as
"not applicable" with the missing module named.  Device-only code paths (the sliding-window engine, the model forwards) need an
MI355X: on a CPU-only host those tests are reported as such by their RuntimeError, not hidden.

    python tools/run_reference_tests.py                     # the default list below
    python tools/run_reference_tests.py tests/unit/test_x.py
    PYTC_STANDIN_KERNELS=1 python tools/run_reference_tests.py ...   # device kernels replaced by the torch stand-ins of tests/

Prints one line per test file and a summary; `--junit DIR` keeps pytest's XML reports."""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.util
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
DEFAULT = ["tests/unit/test_architecture_registry.py", "tests/unit/test_prediction_transform.py", "tests/unit/test_registry_basic.py",
           "tests/unit/test_window_engine.py", "tests/unit/test_inference_stage.py", "tests/unit/test_lazy_inference.py",
           "tests/unit/test_chunked_inference.py", "tests/unit/test_inference_tta_affinity.py", "tests/unit/test_inference_tta_masking.py",
           "tests/unit/test_mednext_multi_head_wrapper.py", "tests/unit/test_mednext_features.py", "tests/test_rsunet.py",
           "tests/unit/test_precomputed_affinity_output.py"]

ALIAS_CONFTEST = '''
import importlib, importlib.abc, importlib.util, os, sys, types
sys.path.insert(0, {root!r})
# packages the image lacks, provided by what IS here: h5py by the in-repo libhdf5 shim, imageio's reader by Pillow
try:
    from pytorch_connectomics_amd.utils import h5lite as _h5
    if _h5.available():
        sys.modules.setdefault("h5py", _h5)
except Exception:
    pass
if "imageio" not in sys.modules:
    try:
        import numpy as _np
        from PIL import Image as _Image
        _io = types.ModuleType("imageio"); _io.v2 = types.ModuleType("imageio.v2")
        _io.imread = _io.v2.imread = lambda f: _np.asarray(_Image.open(f))
        _io.imwrite = _io.v2.imwrite = lambda f, a: _Image.fromarray(_np.asarray(a)).save(f)
        _io.volread = _io.v2.volread = lambda f: _np.stack([_np.asarray(p) for p in __import__("PIL.ImageSequence", fromlist=["x"]).Iterator(_Image.open(f))])
        sys.modules["imageio"], sys.modules["imageio.v2"] = _io, _io.v2
    except Exception:
        pass

REF_PKG = "/root/reference/connectomics"


class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """`connectomics[.x.y]` -> `pytorch_connectomics_amd[.x.y]` (the same module object under both names).  A name this package
    has no counterpart for (data pipeline, config schema, runtime helpers: out of the hot path) falls back to the REFERENCE's own
    file under that name -- the integration picture: the reference with this package swapped in at the seams.  Reference
    packages load as empty namespaces (their __init__ files pull in Lightning / MONAI / Hydra), leaf modules as they are."""
    def find_spec(self, name, path=None, target=None):
        if not (name == "connectomics" or name.startswith("connectomics.")):
            return None
        rest = name[len("connectomics"):]
        real = "pytorch_connectomics_amd" + rest
        try:
            if importlib.util.find_spec(real) is not None:
                return importlib.util.spec_from_loader(name, self, origin=real)
        except (ImportError, ValueError, AttributeError):
            pass
        rel = rest.lstrip(".").replace(".", "/")
        if os.path.isdir(os.path.join(REF_PKG, rel)):
            spec = importlib.util.spec_from_loader(name, self, origin="ref-package:" + rel, is_package=True)
            spec.submodule_search_locations = [os.path.join(REF_PKG, rel)]
            return spec
        leaf = os.path.join(REF_PKG, rel + ".py")
        if os.path.isfile(leaf):
            return importlib.util.spec_from_file_location(name, leaf)
        return None
    def create_module(self, spec):
        if spec.origin.startswith("ref-package:"):
            return None
        return importlib.import_module(spec.origin)
    def exec_module(self, module):
        pass

sys.meta_path.insert(0, _Alias())

if os.environ.get("PYTC_STANDIN_KERNELS") == "1":
    # the product has no CPU path; with this switch the reference's device-free tests run OUR orchestration over the torch stand-ins
    # of the kernels that this repository's own host tests use (tests/test_host_lazy_tta.py) -- test infrastructure, not product
    sys.path.insert(0, os.path.join({root!r}, "tests"))
    import torch
    import test_host_lazy_tta as _L
    import pytorch_connectomics_amd.inference.window as _w
    import pytorch_connectomics_amd.inference.lazy as _lz
    import pytorch_connectomics_amd.inference.tta as _t
    import pytorch_connectomics_amd.inference.tta_ensemble as _e

    for _m in (_w, _lz, _t, _e):
        _m.ops = _L._Ops
    _w.EagerSlidingWindowEngine._check_inputs = lambda self, inputs: torch.device("cpu")
    _init = _w.EagerSlidingWindowEngine.__init__
    def _one_stream(self, *a, **k):
        _init(self, *a, **k)
        self.pipeline_streams = 1
    _w.EagerSlidingWindowEngine.__init__ = _one_stream
'''


def run(rel: str, junit: Path | None) -> dict:
    import tempfile
    target = REF / rel
    with tempfile.TemporaryDirectory() as d:
        plug = Path(d) / "pytc_alias_plugin.py"
        plug.write_text(ALIAS_CONFTEST.format(root=str(ROOT)))
        cmd = [sys.executable, "-m", "pytest", str(target), "-q", "-p", "pytc_alias_plugin", "-p", "no:cacheprovider", "--rootdir", d,
               "-c", "/dev/null", "--no-header", "-rN", "--tb=line"]
        if junit is not None:
            junit.mkdir(parents=True, exist_ok=True)
            cmd += ["--junitxml", str(junit / (target.stem + ".xml"))]
        env = dict(**__import__("os").environ, PYTHONPATH=f"{d}:{ROOT}", PYTHONDONTWRITEBYTECODE="1")
        p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=d, timeout=900)
    tail = [line for line in p.stdout.strip().splitlines() if line.strip()]
    return {"file": rel, "rc": p.returncode, "summary": tail[-1] if tail else p.stderr.strip().splitlines()[-1:],
            "lines": tail}


def main(argv):
    junit = None
    if "--junit" in argv:
        i = argv.index("--junit")
        junit = Path(argv[i + 1])
        argv = argv[:i] + argv[i + 2:]
    verbose = "-v" in argv
    argv = [a for a in argv if a != "-v"]
    if not REF.exists():
        raise SystemExit("/root/reference is not available: this runner only works in the build container")
    for rel in argv or DEFAULT:
        rec = run(rel, junit)
        print(f"{rel}: rc={rec['rc']}  {rec['summary']}")
        if verbose or rec["rc"] not in (0,):
            for line in (rec["lines"][:-1] if verbose else rec["lines"][-12:-1]):
                print("    " + line[:260])


if __name__ == "__main__":
    main(sys.argv[1:])
