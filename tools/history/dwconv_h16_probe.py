"""A/B of the z-march depthwise conv with fp32 taps (v_pk_fma_f32) against the packed-f16 partial-sum form (`dwconv_march_h16`):
time at the network's level 0 / 1 / 2 shapes and the error of both against an fp64 reference on the same bf16 operands."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pytorch_connectomics_amd import _native as nat  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16


def knob(k, v):
    nat.check(nat.lib().pytc_set_tuning(k.encode(), int(v)), "set_tuning")


def timeit(fn, reps=10, warm=5):
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


def main():
    torch.manual_seed(0)
    # accuracy on a small volume against fp64 (same bf16 inputs, fp32 taps)
    N, D, C = 1, 24, 32
    x = (torch.randn(N, D, D, D, C, device=dev) * 2).to(bf)
    taps = torch.randn(27, C, device=dev) * 0.3
    b = torch.randn(C, device=dev) * 0.1
    w64 = taps.t().reshape(C, 1, 3, 3, 3).double().cpu()
    ref = F.conv3d(x.double().cpu().permute(0, 4, 1, 2, 3), w64, b.double().cpu(), padding=1, groups=C).permute(0, 2, 3, 4, 1)
    for h in (0, 1):
        knob("dwconv_march_h16", h)
        y, st = ops.dwconv3d(x, taps, b, K=3)
        err = (y.double().cpu() - ref).abs()
        print(f"h16={h}: max err {float(err.max()):.4e}  mean err {float(err.mean()):.4e}  (|ref| max {float(ref.abs().max()):.2f}, "
              f"mean {float(ref.abs().mean()):.3f}); bf16 rounding of the exact result alone: "
              f"{float((ref.float().to(bf).double() - ref).abs().mean()):.4e} mean", flush=True)
    for (N, D, C) in ((8, 112, 32), (8, 56, 64), (8, 28, 128)):
        x = torch.randn(N, D, D, D, C, device=dev).to(bf)
        taps = torch.randn(27, C, device=dev) * 0.3
        b = torch.randn(C, device=dev)
        res = torch.randn(N, D, D, D, C, device=dev).to(bf)
        for h in (0, 1):
            knob("dwconv_march_h16", h)
            us = timeit(lambda: ops.dwconv3d(x, taps, b, K=3))
            us_r = timeit(lambda: ops.dwconv3d_res(x, taps, res, K=3)) if ops.dwconv3d_res_supported(x, 3, 1) else float("nan")
            print(f"dwconv3d N{N} {D}^3 C{C} h16={h}: {us:8.1f} us = {4 * x.numel() / us / 1e6:5.2f} TB/s (r+w)   with residual: {us_r:8.1f} us", flush=True)
    knob("dwconv_march_h16", 1)


if __name__ == "__main__":
    main()
