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
    k = b._kernel_key("pw_mlp_fwd[32->64->32]")          # streaming, DMA-prefetching, LDS-resident or chunk-streamed form of the shape
    assert k.search("void pytc::pw_mlp_kernel<1, 2, 4, 3, false, false, false, false>") and not k.search("pw_mlp_kernel<1, 4, 4,")
    assert k.search("void pytc::pw_mlp_dma_kernel<2, 0>(pytc::MlpParams, int)") and k.search("pw_mlp_dma_kernel<2, 2>") and not k.search("pw_mlp_dma_kernel<3, 0>")
    assert b._kernel_key("pw_mlp_fwd[128->256->64]").search("void pytc::pw_mlp_chunk_kernel<4, 4, 8, 4>(pytc::MlpChunkParams)")
    assert b._kernel_key("pw_mlp_dma_kernel<2, 1>").search("pw_mlp_dma_kernel<2, 1>(") and not b._kernel_key("pw_mlp_dma_kernel<2, 1>").search("pw_mlp_dma_kernel<2, 0>(")
    assert b._kernel_key("pw_mlp_chunk_kernel<8, 8>").search("pw_mlp_chunk_kernel<8, 8, 8, 2>")
    k = b._kernel_key("pw_mlp_fwd[64->128->32]")
    assert k.search("pytc::pw_mlp_lds_kernel<2, 2, 2, 16, 4>") and k.search("pw_mlp_kernel<2, 2, 4, 3,") and not k.search("pw_mlp_lds_kernel<2, 4,")
    assert b._kernel_key("pw_mlp_lds_kernel<2, 4>").search("void pytc::pw_mlp_lds_kernel<2, 4, 2, 12, 3>(pytc::MlpLdsParams)")
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


def test_kernel_families_and_mfma_bound_entries():
    """Round 3: the step's kernel FAMILIES get their own roofline entries (the z-march depthwise conv runs under one label per shape
    and is the largest rocprof symbol), dense convolutions are priced against the MFMA peak, not against HBM."""
    b = _bench()
    summ = {"dwconv3d_fwd[C32_k3]": {"ms": 3.0, "bytes": 9e9, "launches": 6, "flops": 0, "symbol": "dwconv3d_k3_march_kernel"},
            "dwconv3d_fwd[C64_k3]": {"ms": 1.0, "bytes": 1e9, "launches": 8, "flops": 0, "symbol": "dwconv3d_k3_march_kernel"},
            "dwconv3d_fwd[C256_k3]": {"ms": 0.5, "bytes": 1e8, "launches": 8, "flops": 0, "symbol": "dwconv3d_xblock_kernel"},
            "pw_mlp_fwd[32->64->32]": {"ms": 3.5, "bytes": 14e9, "launches": 8, "flops": 0, "symbol": "pw_mlp_kernel"}}
    assert b.dominant(summ, 2)["kernel"] == "pw_mlp_fwd[32->64->32]"
    rows = b.largest_symbols(summ, 2, top=2, traffic_of=lambda sym: 1_000_000_000 if "march" in sym else None)
    assert [r["kernel"] for r in rows] == ["dwconv3d_k3_march_kernel", "pw_mlp_kernel"]
    m = rows[0]
    assert m["launches_per_step"] == 7.0 and m["ms_per_step"] == 2.0 and m["bound"] == "hbm"
    assert abs(m["achieved"] - 10e9 / 4e-3 / 1e9) < 1e-6 and m["algorithmic_bytes"] == int(10e9 / 14)
    assert m["traffic"] == 1_000_000_000 and abs(m["traffic_over_algorithmic"] - 1.4) < 1e-3
    assert b._kernel_key("dwconv3d_k3_march_kernel") == "dwconv3d_k3_march_kernel" and b._kernel_key("pw_conv_fwd[32->64]") is None
    conv = {"conv3d_fwd[64->64,k333]": {"ms": 1.0, "bytes": 1e9, "launches": 10, "flops": 2e12, "symbol": "conv3d_fwd"}}
    r = b.dominant(conv, 1)
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["achieved"] - 2000.0) < 1e-6 and abs(r["frac"] - 0.8) < 1e-6
    # the committed counter passes, per kernel family
    t = b._traffic_from_table({"void pytc::dwconv3d_k3_march_kernel<x>": (10, 100.0, 50.0), "other": (5, 1.0, 1.0)}, "dwconv3d_k3_march_kernel")
    assert t == int((2 * 100.0 + 50.0) * 1024)


def test_training_labels_merge_by_device_kernel_before_counters_are_attached():
    """`train.roofline`: labels that run one device kernel template are merged, so the event time, the algorithmic bytes and the
    rocprofv3 counter average of that kernel describe the same launches."""
    b = _bench()
    summ = {"pw_conv_fwd[32->64]": {"ms": 3.0, "bytes": 9e9, "launches": 12, "flops": 0},
            "pw_conv_fwd[32->128]": {"ms": 1.5, "bytes": 5e9, "launches": 2, "flops": 0},
            "pw_wgrad_gn[32->64]": {"ms": 2.0, "bytes": 4e9, "launches": 10, "flops": 0},
            "norm_bwd[C32]": {"ms": 1.0, "bytes": 2e9, "launches": 10, "flops": 0}}
    merged, members = b.merge_by_kernel(summ, b._train_kernel_key)
    assert members == {"pw_fast_kernel<1,": ["pw_conv_fwd[32->64]", "pw_conv_fwd[32->128]"],
                       "pw_wgrad_mfma_kernelILi4ELi2E": ["pw_wgrad_gn[32->64]"]}
    assert merged["pw_fast_kernel<1,"] == {"launches": 14, "ms": 4.5, "bytes": 14e9, "flops": 0} and "norm_bwd[C32]" in merged
    table = {"void pytc::pw_fast_kernel<1, 4>(pytc::PwFastParams)": (7, 500000.0, 100000.0)}
    r = b.dominant(merged, 2, traffic_fn=lambda name: b._traffic_from_table(table, name if name in members else None))
    assert r["kernel"] == "pw_fast_kernel<1," and r["traffic"] == int((2 * 500000.0 + 100000.0) * 1024)
    assert r["algorithmic_bytes"] == int(14e9 / 14)


def test_headline_watchdog_prints_the_line_and_leaves(tmp_path):
    """bench.HeadlineWatchdog: a process whose secondary legs never return (here: a sleep standing in for a peer that died inside a
    collective) still ends with ONE JSON line and exit code 0 once the budget is over; a disarmed watchdog does nothing."""
    import json
    import subprocess
    import sys
    import time
    script = tmp_path / "hang.py"
    script.write_text(
        "import importlib.util, json, sys, time\n"
        f"spec = importlib.util.spec_from_file_location('bench_module', {str(ROOT / 'bench.py')!r})\n"
        "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "quiet = b.HeadlineWatchdog(0.2, lambda: json.dumps({'value': -1})).arm(); quiet.disarm()\n"
        "off = b.HeadlineWatchdog(0.0, lambda: json.dumps({'value': -2})).arm()\n"
        "time.sleep(0.6)\n"
        "b.HeadlineWatchdog(0.5, lambda: json.dumps({'value': 42.0, 'errors': {'watchdog': {'0': 'late'}}})).arm()\n"
        "time.sleep(60)\n"
        "print('never')\n")
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=50)
    assert time.perf_counter() - t0 < 40 and p.returncode == 0, p.stderr[-400:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"value": 42.0, "errors": {"watchdog": {"0": "late"}}}
    assert "never" not in p.stdout
