"""RSUNet training step timing (forward + backward + AdamW) on synthetic patches; run under rocprofv3 for the breakdown.
    python tools/history/rsunet_train_probe.py [--dtype bf16] [--patch 18,160,160] [--batch 2] [--steps 5]"""
import argparse
import sys
import time
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--patch", default="18,160,160")
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--norm", default="batch")
ap.add_argument("--width", default="16,32,64,128")
ap.add_argument("--mode", default="both", choices=["both", "train", "infer"])
ap.add_argument("--gc", default="on", choices=["on", "off", "freeze"])
ap.add_argument("--ops", action="store_true")
a = ap.parse_args()
patch = tuple(int(v) for v in a.patch.split(","))
m = RSUNet(1, 3, width=[int(v) for v in a.width.split(",")], norm=a.norm, activation="relu").cuda().train()
m.compute_dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
x = torch.randn(a.batch, 1, *patch, device="cuda")
y = (torch.rand(a.batch, 3, *patch, device="cuda") > 0.5).float()


def step():
    opt.zero_grad(set_to_none=True)
    loss = F.binary_cross_entropy_with_logits(m(x), y)
    loss.backward()
    opt.step()
    return loss


vox = a.batch * patch[0] * patch[1] * patch[2]
if a.mode != "infer":
    for _ in range(5):
        step()
    import gc
    if a.gc == "off":
        gc.disable()
    elif a.gc == "freeze":
        gc.collect(); gc.freeze()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        l = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"rsunet train {a.dtype} norm={a.norm} batch {a.batch} patch {patch}: {dt * 1e3:.1f} ms/step, {vox / dt:.3e} voxels/s, loss {float(l.detach()):.4f}")
if a.ops:
    from pytorch_connectomics_amd import hip_ops as ops
    m.train()
    with ops.profiled() as prof:
        for _ in range(2):
            step()
    summ = prof.summary()
    tot = sum(v["ms"] for v in summ.values())
    print(f"total kernel ms per step: {tot / 2:.2f}")
    for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:45]:
        print(f"  {k:34s} launches={v['launches'] // 2:4d} ms={v['ms'] / 2:8.3f}/step avg_us={v['ms'] / v['launches'] * 1e3:8.1f} "
              f"GB/s={v['bytes'] / max(v['ms'], 1e-9) / 1e6:8.1f}")
if a.mode == "train":
    raise SystemExit(0)
with torch.no_grad():
    m.eval()
    for _ in range(2):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        m(x)
    torch.cuda.synchronize()
    print(f"rsunet infer fwd: {(time.perf_counter() - t0) / a.steps * 1e3:.1f} ms")
