"""Host-side cost of one training step (cProfile): where the Python time goes when the GPU is not the limiter."""
import cProfile, pstats, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import bench

class A: steps = 3; warmup = 2; train_batch = 4
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
bench.train_leg(dev, 0, 1, A, torch.cuda.synchronize)
pr = cProfile.Profile()
pr.enable()
out = bench.train_leg(dev, 0, 1, A, torch.cuda.synchronize)
pr.disable()
print(out["ms_per_step"])
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
