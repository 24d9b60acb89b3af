"""Round 4: stride-2 z-march at C = 32 with 8 x 8 (default) against 4 x 8 output footprints (knob dwconv_s2_tyo4)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pytorch_connectomics_amd import hip_ops as ops
dev = torch.device("cuda:0")
x = torch.randn(8, 112, 112, 112, 32, device=dev).to(torch.bfloat16)
taps = torch.randn(27, 32, device=dev) * 0.2
def timeit(fn, reps=20, warm=6):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
outs = {}
for rnd in range(2):
    for k in (1, 0):
        ops.set_tuning("dwconv_s2_tyo4", k)
        y, st = ops.dwconv3d(x, taps, None, K=3, stride=2)
        outs[k] = y
        print("tyo4", k, round(timeit(lambda: ops.dwconv3d(x, taps, None, K=3, stride=2)), 1), "us", flush=True)
print("equal", torch.equal(outs[0], outs[1]))
