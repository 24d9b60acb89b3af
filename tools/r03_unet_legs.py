"""RSUNet / MONAI-style U-Net legs of bench.py alone (training ms per step, forward ms)."""
import json, sys
from pathlib import Path
from types import SimpleNamespace as NS
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
args = NS(train_steps=10, no_roofline="--roofline" not in sys.argv)
dev = torch.device("cuda", 0)
for name, leg in (("monai_unet", bench.monai_unet_leg), ("rsunet", bench.rsunet_leg)):
    r = leg(dev, args)
    print(name, "train ms", round(r["train_ms_per_step"], 3), "infer ms", round(r["infer_ms_per_forward"], 3))
    if r.get("train_roofline"):
        print("   ", json.dumps(r["train_roofline"]["kernels_ms_per_step"]), r["train_roofline"]["kernel_ms_total_per_step"])
