"""Round 5: one MedNeXt-L k3 forward (+ the three MitoEM heads, 7 channels) over sw_batch 160^3 windows in bf16 -- BASELINE configs[3]'s
network call -- with every profiler label (HIP events per launch, one stream) sorted by time.  Under rocprofv3 the same command gives
the kernel table / HBM counters committed as profiles/r05_mednext_l_*.

    python tools/r05_l_forward.py [n_forwards] [sw_batch] [--no-table]
"""
import sys
from pathlib import Path
from types import SimpleNamespace as NS

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402
from pytorch_connectomics_amd.models import build_model  # noqa: E402

dev = torch.device("cuda:0")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 3
sw = int(args[1]) if len(args) > 1 else 2
heads = {"aff_r1": {"out_channels": 3, "num_blocks": 1, "hidden_channels": 8},
         "aff_r5": {"out_channels": 3, "num_blocks": 1, "hidden_channels": 8},
         "sdt": {"out_channels": 1, "num_blocks": 1, "hidden_channels": 8}}
cfg = NS(model=NS(arch=NS(type="mednext"), in_channels=1, out_channels=7, mednext=NS(size="L", kernel_size=3),
                  loss=NS(deep_supervision=False), heads=heads, primary_head="aff_r1"))
torch.manual_seed(0)
model = build_model(cfg).to(dev).eval()
model.model.compute_dtype = torch.bfloat16
x = torch.rand(sw, 160, 160, 160, 1, device=dev)
VOX = sw * 160 ** 3
with torch.no_grad():
    for _ in range(2):
        model.forward_cl(x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        model.forward_cl(x)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    print(f"MedNeXt-L k3 + 3 heads, {sw} x 160^3 bf16: {ms:.2f} ms per forward = {VOX / ms / 1e6:.3f}e9 window-voxels/s; "
          f"SURVEY 8(d): 2 322 B and 4.781e5 FLOP per window-voxel -> {VOX * 2322 / ms / 1e9:.2f} TB/s = {VOX * 2322 / ms / 1e9 / 8:.3f} of HBM, "
          f"{VOX * 4.781e5 / ms / 1e12:.1f} TFLOP/s = {VOX * 4.781e5 / ms / 1e12 / 2500:.3f} of the dense bf16 MFMA peak", flush=True)
    if "--no-table" not in sys.argv:
        with ops.profiled() as prof:
            for _ in range(n):
                model.forward_cl(x)
        summ = prof.summary()
        tot = sum(r["ms"] for r in summ.values()) / n
        print(f"{'label':52s} {'n/fwd':>6s} {'us/launch':>10s} {'ms/fwd':>8s} {'GB/s':>7s}   symbol      total {tot:.3f} ms per forward (one stream, event-timed)")
        for name, r in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
            us = r["ms"] / r["launches"] * 1e3
            print(f"{name:52s} {r['launches'] / n:6.1f} {us:10.1f} {r['ms'] / n:8.3f} {r['bytes'] / r['launches'] / us / 1e3 if us else 0:7.0f}   {r.get('symbol') or ''}")
