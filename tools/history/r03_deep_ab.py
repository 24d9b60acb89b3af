"""Round 3: deep-level (7^3 / 14^3) GEMMs of the MedNeXt-S training step -- output-channel split of pw_fast (blockIdx.z) and the
un-fused mixer below a row threshold, measured on bench.py's training leg.  python tools/history/r03_deep_ab.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import bench
from pytorch_connectomics_amd import hip_ops as ops
from pytorch_connectomics_amd.training import autograd as ag


class A: steps = 3; warmup = 1; train_batch = 4; train_steps = 8; no_roofline = True


dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for zsplit, min_rows in ((0, 0), (1, 0), (1, 2048), (1, 16384), (1, 100000), (0, 0), (1, 16384)):
    ops.set_tuning("pw_fast_zsplit", zsplit)
    ag.FUSED_TRAIN_MIXER_MIN_ROWS = min_rows
    r = bench.train_leg(dev, 0, 1, A, torch.cuda.synchronize)
    print(f"zsplit={zsplit} fused_mixer_min_rows={min_rows:6d}: {r['ms_per_step']:.3f} ms/step  loss {r['final_loss']:.6f}", flush=True)
