"""Per-label and per-family breakdown of the MedNeXt-S training step (4 x 112^3 bf16, fused loss + FusedAdamW): HIP-event timings
of every launch (single stream), grouped by kernel family and by resolution level."""
import re, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402
from pytorch_connectomics_amd.models.architectures.mednext import create_mednext_v1  # noqa: E402
from pytorch_connectomics_amd.training.fused import FusedAdamW, bce_dice_loss  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
m = create_mednext_v1(1, 1, "S", kernel_size=3).to(dev).train()
m.compute_dtype = torch.bfloat16
x = torch.rand(4, 1, 112, 112, 112, device=dev)
y = (torch.rand(4, 1, 112, 112, 112, device=dev) > 0.85).float()
opt = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0)


def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = bce_dice_loss(m(x), y)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
with ops.profiled() as prof:
    for _ in range(2):
        step()
summ = prof.summary()
tot = sum(r["ms"] for r in summ.values()) / 2
print(f"kernel ms per step {tot:.2f}")
fam = {}
for k, r in summ.items():
    f = k.split("[")[0]
    d = fam.setdefault(f, [0.0, 0, 0])
    d[0] += r["ms"] / 2; d[1] += r["launches"] // 2; d[2] += r["bytes"] / 2
for f, (ms, n, b) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f"  {f:28s} {ms:7.3f} ms  {n:4d} launches  {b / max(ms, 1e-9) / 1e6:8.1f} GB/s")
print("all labels:")
for k, r in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k:36s} {r['ms'] / 2:7.3f} ms  {r['launches'] // 2:3d} x {r['ms'] / r['launches'] * 1e3:7.1f} us  {r['bytes'] / max(r['ms'], 1e-9) / 1e6:8.1f} GB/s")
