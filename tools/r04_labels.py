"""Round 4: every profiler label of one 8-window MedNeXt-S forward (HIP events per launch, single stream), sorted by time, with the
algorithmic bytes the wrapper declares and the rate they imply -- the table the by_label / by_symbol entries of the bench line are cut from.

    python tools/r04_labels.py [n_batches]
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
model = bench.build_model(dev)
x = torch.rand(8, 112, 112, 112, 1, device=dev)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(3):
        model.forward_cl(x)
    torch.cuda.synchronize()
    with ops.profiled() as prof:
        for _ in range(n):
            model.forward_cl(x)
summ = prof.summary()
tot = sum(r["ms"] for r in summ.values()) / n
print(f"{'label':46s} {'n/step':>6s} {'us/launch':>10s} {'ms/step':>8s} {'GB/s':>7s}   symbol      total {tot:.3f} ms/step")
for name, r in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
    us = r["ms"] / r["launches"] * 1e3
    print(f"{name:46s} {r['launches'] / n:6.1f} {us:10.1f} {r['ms'] / n:8.3f} {r['bytes'] / r['launches'] / us / 1e3 if us else 0:7.0f}   {r.get('symbol') or ''}")
