"""Run the REFERENCE's own unit tests for the hot path against THIS package (build container only: needs /root/reference).

The test files are executed where they lie under /root/reference/tests -- nothing is copied.  `connectomics.*` imports are
redirected to `pytorch_connectomics_amd.*` by a meta-path alias; modules of the reference this package has no counterpart for
(config schema, data pipeline, decoding) are not aliased, so a test file that needs them fails at import and is reported as
"not applicable" with the missing module named.  Device-only code paths (the sliding-window engine, the model forwards) need an
MI355X: on a CPU-only host those tests are reported as such by their RuntimeError, not hidden.

    python tools/run_reference_tests.py                     # the default list below
    python tools/run_reference_tests.py tests/unit/test_x.py

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
           "tests/unit/test_mednext_multi_head_wrapper.py", "tests/unit/test_mednext_features.py", "tests/test_rsunet.py"]

ALIAS_CONFTEST = '''
import importlib, importlib.abc, importlib.util, sys
sys.path.insert(0, {root!r})

class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """`connectomics[.x.y]` -> `pytorch_connectomics_amd[.x.y]` (the same module object under both names)."""
    def find_spec(self, name, path=None, target=None):
        if name == "connectomics" or name.startswith("connectomics."):
            real = "pytorch_connectomics_amd" + name[len("connectomics"):]
            try:
                if importlib.util.find_spec(real) is None:
                    return None
            except (ImportError, ValueError):
                return None
            return importlib.util.spec_from_loader(name, self, origin=real)
        return None
    def create_module(self, spec):
        return importlib.import_module(spec.origin)
    def exec_module(self, module):
        pass

sys.meta_path.insert(0, _Alias())
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
            for line in rec["lines"][-12:-1]:
                print("    " + line[:220])


if __name__ == "__main__":
    main(sys.argv[1:])
