"""How fast is the CPU oracle on the GPU box at torch's DEFAULT thread count vs 32 / 16 threads?  (The GPU tests run the oracle at
BASELINE sizes; bench.py's sweep says 256 threads are 13x slower than 32.)"""
import os, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import mednext_oracle as MO
from pytorch_connectomics_amd.models.architectures.mednext import create_mednext_v1
torch.manual_seed(0)
m = create_mednext_v1(1, 1, "S", 3)
st = {k: v.detach().float() for k, v in m.state_dict().items()}
kw = dict(n_channels=32, exp_r=2, kernel_size=3, block_counts=[2] * 9)
x = torch.rand(1, 1, 112, 112, 112)
print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads(), flush=True)
for n in (torch.get_num_threads(), 32, 16, 64):
    torch.set_num_threads(n)
    with torch.no_grad():
        t0 = time.perf_counter(); MO.forward(st, x, **kw); dt = time.perf_counter() - t0
    print(f"threads {n:4d}: {dt:6.2f} s per 112^3 forward", flush=True)
