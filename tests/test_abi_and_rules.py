"""CPU checks: the C-ABI library loads and exports every symbol include/pytc_hip.h declares; the
product package never touches the oracle; the product refuses to run without a GPU."""
import ctypes
import os
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "pytc_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pytc_[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pytorch_connectomics_amd import _native
    lib = _native.lib()     # builds nothing; the .so must have been built by __graft_entry__.build()
    declared = _declared_symbols()
    assert declared, "no symbols parsed from the header"
    handle = ctypes.CDLL(str(_native.LIB_PATH))
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in pytc_hip.h but not exported"
    assert set(_native.exported_symbols()) <= set(declared)
    header = (ROOT / "include" / "pytc_hip.h").read_text()
    assert lib.pytc_abi_version() == _native.ABI_VERSION == int(re.search(r"#define PYTC_ABI_VERSION (\d+)", header).group(1))


def test_abi_pure_host_queries():
    from pytorch_connectomics_amd import _native as nat
    lib = nat.lib()
    assert lib.pytc_pw_packed_elems(64, 32, nat.BF16) == 64 * 32
    assert lib.pytc_pw_packed_elems(1, 32, nat.BF16) == 16 * 32          # rows padded to 16
    assert lib.pytc_pw_packed_elems(3, 5, nat.F32) == 16 * 16
    assert lib.pytc_pw_packed_elems(0, 5, nat.F32) == -1
    assert lib.pytc_dwconv3d_stat_slots(8, 112, 112, 112, 32, 3, 1, nat.BF16, 0) > 0
    assert lib.pytc_dwconv3d_stat_slots(8, 7, 7, 7, 512, 3, 1, nat.BF16, 0) > 0


def test_product_never_imports_oracle_or_reference():
    bad = []
    for p in (ROOT / "pytorch_connectomics_amd").rglob("*.py"):
        src = p.read_text()
        if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "/root/reference" in src:
            bad.append(str(p))
    for p in list((ROOT / "pytorch_connectomics_amd").rglob("*.hip")) + list((ROOT / "pytorch_connectomics_amd").rglob("*.h")):
        if "oracle" in p.read_text():
            bad.append(str(p))
    assert not bad, f"product files reference the oracle / reference tree: {bad}"
    # run-time files must not read /root/reference either
    for name in ("bench.py", "__graft_entry__.py"):
        assert "/root/reference" not in (ROOT / name).read_text()


def test_no_cpu_fallback():
    from pytorch_connectomics_amd import hip_ops as ops
    from pytorch_connectomics_amd.inference.window import EagerSlidingWindowEngine
    from pytorch_connectomics_amd.models.architectures.mednext import MedNeXt
    m = MedNeXt(1, 8, 1, exp_r=2, kernel_size=3, do_res=True, do_res_up_down=True, block_counts=[1] * 9).eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        with torch.no_grad():
            m(torch.zeros(1, 1, 16, 16, 16))
    eng = EagerSlidingWindowEngine(roi_size=(8, 8, 8), sw_batch_size=1, overlap=0.5, mode="bump",
                                   padding_mode="constant", cval=0.0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        eng(torch.zeros(1, 1, 16, 16, 16), lambda x: x)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.blend_finalize(torch.zeros(1, 4, 4, 4), torch.zeros(4, 4, 4))


def test_tuning_knobs_from_the_environment():
    """PYTC_TUNING="knob=value,..." reaches pytc_set_tuning when the library is loaded (A/B runs of unmodified commands); a
    malformed entry fails loudly instead of being ignored."""
    import subprocess
    import sys
    root = str(Path(__file__).resolve().parent.parent)
    code = "from pytorch_connectomics_amd import _native as nat; nat.lib(); print('loaded')"
    good = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True,
                          env={**os.environ, "PYTC_TUNING": "dwconv_mfma=0, mlp_lds_variant=3"})
    assert good.returncode == 0 and "loaded" in good.stdout, good.stderr[-500:]
    for bad_value in ("dwconv_mfma", "dwconv_mfma=fast", "=3"):
        bad = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True,
                             env={**os.environ, "PYTC_TUNING": bad_value})
        assert bad.returncode != 0 and "PYTC_TUNING" in bad.stderr, (bad_value, bad.stderr[-300:])


def test_depthwise_dispatch_table_and_batch_invariant_statistics_slots():
    """Which device kernel a depthwise launch of the MedNeXt-S 112^3 window takes at every level (pytc_dwconv3d_kernel_variant mirrors
    the dispatch), and that the number of statistics slots per sample never depends on the batch size: a window's GroupNorm partial
    sums -- and with them its bf16 prediction -- must not change with the batch it travels in (chunked == whole-volume exactness)."""
    from pytorch_connectomics_amd import _native as nat
    lib = nat.lib()
    kv, slots = lib.pytc_dwconv3d_kernel_variant, lib.pytc_dwconv3d_stat_slots
    B, F = nat.BF16, nat.F32
    # stride 1, K = 3: matrix-core z-march from 112^3 down to 14^3, x-block kernel at the 7^3 bottleneck; fp32: VALU z-march / gather
    for side, C in ((112, 32), (56, 64), (28, 128), (14, 256)):
        assert kv(8, side, side, side, C, 3, 1, B, 0) == 6, (side, C)
    assert kv(8, 7, 7, 7, 512, 3, 1, B, 0) == 2
    assert kv(8, 112, 112, 112, 32, 3, 1, F, 0) == 3 and kv(8, 14, 14, 14, 256, 3, 1, F, 0) in (1, 2)
    # down blocks: LDS z-march at C = 32 / 64, gather kernel deeper; up blocks: tile kernel at C = 64 / 128, cell kernel deeper
    assert kv(8, 112, 112, 112, 32, 3, 2, B, 0) == 8 and kv(8, 56, 56, 56, 64, 3, 2, B, 0) == 8 and kv(8, 28, 28, 28, 128, 3, 2, B, 0) == 1
    assert kv(8, 56, 56, 56, 64, 3, 2, B, 1) == 7 and kv(8, 28, 28, 28, 128, 3, 2, B, 1) == 7 and kv(8, 14, 14, 14, 256, 3, 2, B, 1) == 4
    # K = 5 / 7 and odd channel counts stay on the generic kernels
    assert kv(2, 32, 32, 32, 32, 5, 1, B, 0) in (1, 2) and kv(2, 32, 32, 32, 12, 3, 1, B, 0) in (0, 1)
    for args in ((112, 112, 112, 32, 3, 1, B, 0), (56, 56, 56, 64, 3, 1, B, 0), (14, 14, 14, 256, 3, 1, B, 0), (112, 112, 112, 32, 3, 2, B, 0),
                 (56, 56, 56, 64, 3, 2, B, 0), (56, 56, 56, 64, 3, 2, B, 1), (28, 28, 28, 128, 3, 2, B, 1), (33, 47, 20, 64, 3, 1, B, 0),
                 (9, 20, 31, 32, 3, 2, B, 0), (160, 160, 160, 32, 3, 1, B, 0)):
        counts = {slots(n, *args) for n in (1, 2, 8, 13)}
        assert len(counts) == 1 and counts.pop() > 0, args
