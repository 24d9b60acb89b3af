"""Host-side cost of one RSUNet training step (cProfile)."""
import cProfile, pstats, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import torch.nn.functional as F
from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet

m = RSUNet(1, 3, width=[16, 32, 64, 128], norm="batch", activation="relu").cuda().train()
m.compute_dtype = torch.bfloat16
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
x = torch.randn(2, 1, 18, 160, 160, device="cuda")
y = (torch.rand(2, 3, 18, 160, 160, device="cuda") > 0.5).float()


def step():
    opt.zero_grad(set_to_none=True)
    F.binary_cross_entropy_with_logits(m(x), y).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
