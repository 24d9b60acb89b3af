"""A few training steps of one dense-conv leg of bench.py (rsunet | rsunet_pow2 | monai) for a kernel trace."""
import sys
from pathlib import Path
from types import SimpleNamespace as NS

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "rsunet"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
args = NS(train_steps=6, no_roofline=True)
if which == "monai":
    r = bench.monai_unet_leg(dev, args)
else:
    from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet
    if which == "rsunet":
        r = bench._unet_leg(dev, args, lambda: RSUNet(1, 1, **bench.RSUNET_STOCK), "rsunet stock", (18, 256, 256), 2, 1)
    else:
        r = bench._unet_leg(dev, args, lambda: RSUNet(1, 1, width=[16, 32, 64, 128], norm="batch", activation="relu"), "pow2", (18, 256, 256), 2, 1)
print({k: r[k] for k in ("model", "train_ms_per_step", "infer_ms_per_forward")})
