"""Round 4: where the 10 us of groupnorm_fold_mlp_kernel go -- back-to-back launches (no events between them), slot count and widths varied."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=200, warm=20):
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


for C, chid, slots in [(32, 64, 1), (32, 64, 98), (32, 64, 1176), (64, 128, 1), (64, 128, 196), (128, 256, 1), (128, 256, 49), (128, 512, 49)]:
    N = 8
    stats = torch.rand(N, slots, 2, C, device=dev)
    stats[:, :, 1] += 1.0
    g, b = torch.rand(C, device=dev), torch.rand(C, device=dev)
    w2, b2 = torch.randn(chid, C, device=dev), torch.randn(chid, device=dev)
    us = timeit(lambda: ops.groupnorm_fold_mlp(stats, 1000.0, g, b, 1e-5, w2, b2))
    us_fin = timeit(lambda: ops.groupnorm_finalize(stats, 1000.0, g, b, 1e-5))
    print(f"C {C:4d} hid {chid:4d} slots {slots:5d}: fold {us:6.2f} us   finalize {us_fin:6.2f} us", flush=True)
x = torch.zeros(8, device=dev)
print(f"empty-ish torch kernel: {timeit(lambda: x.add_(1)):6.2f} us")
