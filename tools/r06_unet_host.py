"""Host enqueue time against wall time of the dense-conv training steps (RSUNet stock / MONAI-style)."""
import sys
import time
from pathlib import Path
from types import SimpleNamespace as NS

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from pytorch_connectomics_amd.training.fused import FusedAdamW, bce_dice_loss  # noqa: E402
from pytorch_connectomics_amd.utils.hostgc import quiesce_gc  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def run(name, make, patch):
    torch.manual_seed(0)
    m = make().to(dev).train()
    inner = getattr(m, "model", m)
    for mod in (m, inner):
        if hasattr(mod, "compute_dtype"):
            mod.compute_dtype = torch.bfloat16
    opt = FusedAdamW(m.parameters(), lr=1e-4, weight_decay=1e-2, max_grad_norm=1.0)
    x = torch.rand(2, 1, *patch, device=dev)
    y = (torch.rand(2, 1, *patch, device=dev) > 0.85).float()
    marks = []

    def tstep():
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        out = m(x)
        t1 = time.perf_counter()
        loss, _ = bce_dice_loss(out, y)
        loss.backward()
        t2 = time.perf_counter()
        opt.step()
        t3 = time.perf_counter()
        marks.append((t1 - t0, t2 - t1, t3 - t2))

    for _ in range(4):
        tstep()
    quiesce_gc()
    torch.cuda.synchronize()
    marks.clear()
    n = 8
    t0 = time.perf_counter()
    for _ in range(n):
        tstep()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    ta = time.perf_counter() - t0
    f = sum(a for a, _, _ in marks) / n * 1e3
    b = sum(b_ for _, b_, _ in marks) / n * 1e3
    o = sum(c for _, _, c in marks) / n * 1e3
    print(f"{name}: host {th / n * 1e3:.2f} ms per step (fwd {f:.2f}, loss+bwd {b:.2f}, opt {o:.2f}), wall {ta / n * 1e3:.2f} ms per step", flush=True)


from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet  # noqa: E402
run("rsunet stock", lambda: RSUNet(1, 1, **bench.RSUNET_STOCK), (18, 256, 256))
run("rsunet pow2", lambda: RSUNet(1, 1, width=[16, 32, 64, 128], norm="batch", activation="relu"), (18, 256, 256))
from pytorch_connectomics_amd.models import build_model as bm  # noqa: E402
cfg = NS(model=NS(arch=NS(type="monai_unet"), in_channels=1, out_channels=1, input_size=[24, 256, 256],
                  monai=NS(filters=[32, 64, 128, 256], num_res_units=2, kernel_size=3, norm="batch", dropout=0.0, upsample_mode="deconv")))
run("monai", lambda: bm(cfg), (24, 256, 256))
