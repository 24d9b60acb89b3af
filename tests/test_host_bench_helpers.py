"""CPU checks of bench.py's roofline bookkeeping: which rocprof kernel a bench label maps to, and the committed PMC passes."""
import importlib.util
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_counter_traffic_of_the_dominant_kernel():
    b = _bench()
    assert b._kernel_key("pw_mlp_fwd[32->64->32]") == "pw_mlp_kernel<1, 2,"
    assert b._kernel_key("pw_mlp_fwd[64->128->32]") == "pw_mlp_kernel<2, 2,"
    assert b._kernel_key("dwconv3d_fwd[C32_k3]") is None          # runs at several shapes under one name: no per-shape average
    t = b.pmc_traffic_bytes("pw_mlp_fwd[32->64->32]")              # committed rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE)
    alg = 3 * 32 * 2 * 8 * 112 ** 3                                # t + residual + y of a plain launch
    assert t is not None and 0.7 * alg < t < 1.05 * alg            # launch-weighted over the head / stem-residual variants
    assert b.pmc_traffic_bytes("dwconv3d_fwd[C32_k3]") is None
    assert b.live_pmc_traffic_bytes("dwconv3d_fwd[C32_k3]") is None          # nothing to collect: returns before any child run


def test_dominant_kernel_record():
    b = _bench()
    summ = {"a": {"ms": 2.0, "bytes": 8e9, "launches": 4}, "b": {"ms": 1.0, "bytes": 1e9, "launches": 10}}
    r = b.dominant(summ, n_steps=2)
    assert r["kernel"] == "a" and r["unit"] == "GB/s" and r["bound"] == "hbm"
    assert abs(r["achieved"] - 4000.0) < 1e-6 and abs(r["frac"] - 0.5) < 1e-6 and r["launch_us"] == 500.0
    assert r["algorithmic_bytes"] == 2_000_000_000 and abs(r["share_of_step"] - 0.667) < 1e-3 and r["traffic"] is None
