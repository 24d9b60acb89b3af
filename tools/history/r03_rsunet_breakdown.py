"""Per-family breakdown of the RSUNet / MONAI-style U-Net training steps of bench.py (HIP-event timings, single stream)."""
import sys
from pathlib import Path
from types import SimpleNamespace as NS
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402
from pytorch_connectomics_amd.training.fused import FusedAdamW, bce_dice_loss  # noqa: E402

dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "rsunet"
if which == "rsunet":
    from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet
    m = RSUNet(1, 3, width=[16, 32, 64, 128], norm="batch", activation="relu").to(dev).train()
    patch, out_ch = (18, 160, 160), 3
else:
    from pytorch_connectomics_amd.models import build_model as bm
    cfg = NS(model=NS(arch=NS(type="monai_unet"), in_channels=1, out_channels=1, input_size=[24, 256, 256],
                      monai=NS(filters=[32, 64, 128, 256], num_res_units=2, kernel_size=3, norm="batch", dropout=0.0, upsample_mode="deconv")))
    m = bm(cfg).to(dev).train()
    patch, out_ch = (24, 256, 256), 1
for mod in (m, getattr(m, "model", m)):
    if hasattr(mod, "compute_dtype"):
        mod.compute_dtype = torch.bfloat16
opt = FusedAdamW(m.parameters(), lr=1e-4, weight_decay=1e-2, max_grad_norm=1.0)
x = torch.rand(2, 1, *patch, device=dev)
y = (torch.rand(2, out_ch, *patch, device=dev) > 0.85).float()


def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = bce_dice_loss(m(x), y)
    loss.backward()
    opt.step()


for _ in range(4):
    step()
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print(f"{which}: {1e2 * (time.perf_counter() - t0):.2f} ms per step (wall)")
with ops.profiled() as prof:
    for _ in range(2):
        step()
summ = prof.summary()
print(f"kernel ms per step {sum(r['ms'] for r in summ.values()) / 2:.2f}, launches per step {sum(r['launches'] for r in summ.values()) // 2}")
fam = {}
for k, r in summ.items():
    d = fam.setdefault(k.split("[")[0], [0.0, 0, 0])
    d[0] += r["ms"] / 2; d[1] += r["launches"] // 2; d[2] += r.get("flops", 0) / 2
for f, (ms, n, fl) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"  {f:28s} {ms:7.3f} ms  {n:4d} launches  avg {ms / max(n, 1) * 1e3:7.1f} us" + (f"  {fl / ms / 1e9:7.1f} TFLOP/s" if fl else ""))
for k, r in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:14]:
    print(f"     {k:40s} {r['ms'] / 2:7.3f} ms  {r['launches'] // 2:3d} x {r['ms'] / r['launches'] * 1e3:7.1f} us")
