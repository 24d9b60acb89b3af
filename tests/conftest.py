import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    # The parity tests run the CPU oracle at BASELINE sizes.  torch defaults to one thread per physical core (128 on the MI355X
    # host), where the oracle's many small convolutions spend their time in thread hand-off: measured there
    # (tools/r03_threads_probe.py, gpurun_out/r03_threads_probe.txt) one MedNeXt-S 112^3 forward takes 10.1 s at 128 threads,
    # 4.8 s at 32, 5.2 s at 16, 5.8 s at 64 -- the same optimum bench.py's cpu_baseline sweep finds.  Results do not depend on it.
    import torch
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
