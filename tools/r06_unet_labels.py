"""Every profiler label of one MONAI-style U-Net (or RSUNet) training step, sorted by time (HIP events per launch, bench.py's model and shapes).
    python tools/r06_unet_labels.py [monai|rsunet] [top N]"""
import sys
from pathlib import Path
from types import SimpleNamespace as NS

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "monai"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
captured = {}
orig = bench.dominant


def spy(summ, n_steps, **kw):
    captured["summ"], captured["n"] = summ, n_steps
    return orig(summ, n_steps, **kw)


bench.dominant = spy
args = NS(train_steps=10, no_roofline=False)
leg = bench.monai_unet_leg if which == "monai" else bench.rsunet_leg
r = leg(torch.device("cuda", 0), args)
print(which, "train ms", round(r["train_ms_per_step"], 3), "infer ms", round(r["infer_ms_per_forward"], 3))
summ, n = captured["summ"], captured["n"]
total = sum(v["ms"] for v in summ.values()) / n
print(f"kernel ms per step {total:.3f}, labels {len(summ)}")
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:top]:
    print(f"  {k:44s} launches/step={v['launches'] / n:5.1f} ms/step={v['ms'] / n:7.3f} us/launch={v['ms'] / v['launches'] * 1e3:8.1f} "
          f"GB/s={v['bytes'] / max(v['ms'], 1e-9) / 1e6:8.1f} TFLOP/s={v['flops'] / max(v['ms'], 1e-9) / 1e9:7.1f}")
