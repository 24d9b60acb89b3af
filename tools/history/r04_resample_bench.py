"""Round 4: the resampling depthwise convs of the down / up blocks (stride-2 gather kernel, transposed cell kernel) against plain
fills / copies of the same byte counts -- how far from the memory roof they are.

    python tools/history/r04_resample_bench.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
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


for (N, D, C) in [(8, 56, 64), (8, 28, 128)]:
    x = torch.randn(N, D, D, D, C, device=dev).to(bf)
    taps = torch.randn(27, C, device=dev) * 0.2
    y = torch.empty(N, 2 * D, 2 * D, 2 * D, C, device=dev, dtype=bf)
    us = timeit(lambda: ops.dwconv3d(x, taps, None, K=3, transposed=True, y=y))
    ops.set_tuning('dwconvT_tile', 0)
    us_cell = timeit(lambda: ops.dwconv3d(x, taps, None, K=3, transposed=True, y=y))
    ops.set_tuning('dwconvT_tile', 1)
    us_ns = timeit(lambda: ops.dwconv3d(x, taps, None, K=3, transposed=True, y=y, stats=False))
    us_fill = timeit(lambda: y.zero_())
    us_st = timeit(lambda: ops.dwconv3d(x, taps, None, K=3, transposed=True, store=False)) if C in (64, 128) else 0.0
    print(f"dwconvT {N}x{D}^3x{C} -> {2 * D}^3: cell kernel {us_cell:7.1f} us, tile kernel {us:7.1f} us ({(x.numel() + y.numel()) * 2 / us / 1e3:5.0f} GB/s)  without statistics {us_ns:7.1f}  "
          f"statistics only {us_st:7.1f}  zero_() of the output {us_fill:7.1f} us ({y.numel() * 2 / us_fill / 1e3:5.0f} GB/s)", flush=True)
for (N, D, C) in [(8, 112, 32), (8, 56, 64)]:
    x = torch.randn(N, D, D, D, C, device=dev).to(bf)
    taps = torch.randn(27, C, device=dev) * 0.2
    y = torch.empty(N, D // 2, D // 2, D // 2, C, device=dev, dtype=bf)
    us = timeit(lambda: ops.dwconv3d(x, taps, None, K=3, stride=2, y=y))
    z = torch.empty_like(x)
    us_copy = timeit(lambda: z.copy_(x))
    us_sum = timeit(lambda: x.view(torch.int16).sum())
    print(f"dwconv s2 {N}x{D}^3x{C} -> {D // 2}^3: {us:7.1f} us ({(x.numel() + y.numel()) * 2 / us / 1e3:5.0f} GB/s)   copy_ of the input {us_copy:7.1f} us, "
          f"sum of the input {us_sum:7.1f} us ({x.numel() * 2 / us_sum / 1e3:5.0f} GB/s)", flush=True)
