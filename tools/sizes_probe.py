import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from types import SimpleNamespace as NS
from pytorch_connectomics_amd.models import build_model
for size, k in (("S", 3), ("B", 3), ("M", 3), ("L", 3), ("S", 5), ("L", 5)):
    cfg = NS(model=NS(arch=NS(type="mednext"), in_channels=1, out_channels=3, mednext=NS(size=size, kernel_size=k),
                      loss=NS(deep_supervision=False), heads=None))
    torch.manual_seed(0)
    m = build_model(cfg).cuda().eval()
    x = torch.rand(2, 1, 64, 64, 64, device="cuda")
    with torch.no_grad():
        y32 = m(x)
        m.model.compute_dtype = torch.bfloat16
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y16 = m(x)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = sum(p.numel() for p in m.parameters())
    print(f"MedNeXt-{size} k{k}: params {n/1e6:.1f} M, out {tuple(y32.shape)}, |fp32-bf16| max {(y32 - y16.float()).abs().max():.3e} "
          f"(scale {y32.abs().max():.2f}), bf16 fwd {dt*1e3:.1f} ms", flush=True)
