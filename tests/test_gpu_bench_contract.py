"""bench.py's output contract, and its N > 1 control flow on a single-GPU box (two ranks share the GPU and talk over gloo:
PYTC_BENCH_SHARE_GPU=1, a test hook the driver never sets).  Guards the collectives of the multi-rank path -- barriers, MAX over
ranks, the slab halo exchange, DDP steps that every rank has to run -- which the builder cannot run on real multi-GPU hardware."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(stdout: str) -> dict:
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_every_contract_field():
    env = dict(os.environ, PYTC_BENCH_VOLUME="165x336x336")
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--train-steps", "2", "--no-extras",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True and d["dtype"] == "bf16"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    roof = d["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] == "GB/s" and 0 < roof["frac"] <= 1.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    # HBM bytes per launch of the dominant kernel: PMC counters collected live by child runs under rocprofv3 (or, without the
    # profiler, the committed passes); a fused mixer moves its algorithmic bytes, not more
    assert roof["traffic_source"].startswith(("live", "committed"))
    # `roofline` is the largest rocprof SYMBOL of the step, the largest per-shape label rides as `by_label`
    assert "_kernel" in roof["kernel"] and roof["kernel"] == roof["by_symbol"][0]["kernel"]
    lab = roof["by_label"]
    if roof["traffic_source"].startswith("live") and lab.get("traffic"):
        assert 0.7 * lab["algorithmic_bytes"] < lab["traffic"] < 1.3 * lab["algorithmic_bytes"]      # a fused mixer re-reads nothing
    ws = roof["whole_step"]
    assert ws["algorithmic_bytes_per_voxel"] == 1557.0
    assert abs(ws["frac"] - d["value"] * 1557.0 / 8e12) < 5e-4
    tr = d["train"]
    assert tr["ms_per_step"] > 0 and tr["roofline"]["frac"] > 0 and tr["roofline"]["by_symbol"]
    assert abs(tr["roofline"]["whole_step"]["frac"] - tr["value"] * 3 * 1557.0 / 8e12) < 5e-4


def test_two_ranks_sharing_the_gpu_run_weak_strong_and_ddp_legs():
    env = dict(os.environ, PYTC_BENCH_SHARE_GPU="1", PYTC_BENCH_VOLUME="165x336x336")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_port()), "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--train-steps", "2", "--train-batch", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["windows_per_step"] > 0
    assert d["strong_slab"]["scaling"] == "strong" and d["strong_slab"]["seconds"] > 0
    assert d["train"]["parallelism"] == "ddp2" and d["train"]["ms_per_step"] > 0
    assert d["cpu_baseline"] is None or d["cpu_baseline"]["cores"] >= 1      # rank 0 at N = 1 only
