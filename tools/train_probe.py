"""Three MedNeXt-S bf16 training steps (4 x 112^3) for profiling; same step as bench.py's training leg."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench

class A: steps = 3; warmup = 1; train_batch = 4
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
print(bench.train_leg(dev, 0, 1, A, torch.cuda.synchronize))
