"""Round 4: the matrix-core depthwise conv (csrc/dwconv_mfma_kernels.hip, knob dwconv_mfma) against the VALU z-march (dwconv_mfma = 0)
and an fp32 torch reference: error, statistics, time per launch at the network's shapes.

    python tools/r04_dwmfma.py
"""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16


def timeit(fn, reps=20, warm=6):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def ref(x, taps, bias):
    C = x.shape[-1]
    w = taps.t().reshape(C, 1, 3, 3, 3)
    return F.conv3d(x.float().permute(0, 4, 1, 2, 3), w, bias, padding=1, groups=C).permute(0, 2, 3, 4, 1)


def check(N, D, H, W, C, scale=1.0):
    torch.manual_seed(D + C)
    x = (torch.randn(N, D, H, W, C, device=dev) * scale).to(bf)
    taps = torch.randn(27, C, device=dev) * 0.2
    bias = torch.randn(C, device=dev)
    want = ref(x, taps, bias)
    out = {}
    for mf in (0, 1):
        ops.set_tuning("dwconv_mfma", mf)
        for v in ((0, 1) if mf else (0,)):
            ops.set_tuning("dwconv_mfma_variant", v)
            y, st = ops.dwconv3d(x, taps, bias, K=3)
            err = (y.float() - want).abs()
            rel = float(err.max() / want.abs().max())
            s = st.sum(1)
            yb = y.float()
            s_err = float((s[:, 0] - yb.sum((1, 2, 3))).abs().max() / yb.sum((1, 2, 3)).abs().max())
            q_err = float((s[:, 1] - (yb * yb).sum((1, 2, 3))).abs().max() / (yb * yb).sum((1, 2, 3)).abs().max())
            ulp_ok = float((err <= want.abs() * 2.0 ** -8 + 1e-6 * scale).float().mean())
            out[(mf, v)] = (rel, float(err.mean() / want.abs().mean()), ulp_ok, s_err, q_err)
    ops.set_tuning("dwconv_mfma_variant", 0)
    ops.set_tuning("dwconv_mfma", 1)
    print(f"check {N}x{D}x{H}x{W}x{C} scale {scale:g}:")
    for k, v in out.items():
        print(f"   mfma={k[0]} variant={k[1]}: max rel {v[0]:.2e} mean rel {v[1]:.2e} within-1-bf16-ulp {v[2]:.6f} stats sum err {v[3]:.1e} sumsq err {v[4]:.1e}")
    return out


def bench(N, D, C):
    x = torch.randn(N, D, D, D, C, device=dev).to(bf)
    taps = torch.randn(27, C, device=dev) * 0.2
    bias = torch.randn(C, device=dev)
    y = torch.empty_like(x)
    nb = 2 * x.numel() * 2
    line = f"bench {N}x{D}^3x{C}:"
    ops.set_tuning("dwconv_mfma", 0)
    us = timeit(lambda: ops.dwconv3d(x, taps, bias, K=3, y=y))
    line += f"  z-march {us:7.1f} us ({nb / us / 1e3:5.0f} GB/s) |"
    ops.set_tuning("dwconv_mfma", 1)
    for v in (0, 1, 2, 3, 91, 93):
        ops.set_tuning("dwconv_mfma_variant", v if v < 90 else 0)
        ops.set_tuning("dwconv_mfma_probe", v - 90 if v >= 90 else 0)      # 91 / 93: the timing probes (wrong results)
        us = timeit(lambda: ops.dwconv3d(x, taps, bias, K=3, y=y))
        line += f"  v{v} {us:6.1f}"
    ops.set_tuning("dwconv_mfma_variant", 0)
    ops.set_tuning("dwconv_mfma_probe", 0)
    print(line, flush=True)


if __name__ == "__main__":
    check(1, 9, 20, 31, 32)
    check(2, 16, 32, 28, 64)
    check(1, 30, 17, 16, 32, scale=1e-4)
    check(1, 24, 24, 24, 128, scale=300.0)
    bench(8, 112, 32)
    bench(8, 56, 64)
    bench(8, 28, 128)
    bench(1, 112, 32)
    if len(sys.argv) > 1 and sys.argv[1] == "quant":
        for n in (4, 5, 8, 10, 13, 16):
            bench(n, 56, 64)
        for n in (7, 8, 9):
            bench(n, 112, 32)
