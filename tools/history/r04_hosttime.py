"""Round 4: host time to ENQUEUE one 8-window MedNeXt-S forward (python + ctypes + allocator; no synchronisation inside the loop)
against the GPU time of the same forwards -- is the three-stream window pipeline host-bound?"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402

dev = torch.device("cuda:0")
model = bench.build_model(dev)
x = torch.rand(8, 112, 112, 112, 1, device=dev)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(3):
        model.forward_cl(x)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        model.forward_cl(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e3 * (t1 - t0) / n:.2f} ms per forward; GPU complete {1e3 * (t2 - t0) / n:.2f} ms per forward (one stream)")
    xs = torch.rand(8, 16, 16, 16, 1, device=dev)
    for _ in range(3):
        model.forward_cl(xs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model.forward_cl(xs)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"tiny input (8 x 16^3): {1e3 * (t1 - t0) / n:.2f} ms per forward, launch-bound: host + launch latency floor")
