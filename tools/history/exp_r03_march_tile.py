"""Round-3 experiment: z-march depthwise conv footprint 8 x 8 / 256 threads vs 8 x 16 / 512 threads (knob dwconv_march_tx16):
same values, launch time in isolation at the network's level-0 shape (8 x 112^3 x 32 bf16)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from pytorch_connectomics_amd import _native as nat  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for shape in ((8, 112, 112, 112, 32), (8, 56, 56, 56, 64), (1, 112, 112, 112, 32), (2, 48, 80, 112, 32)):
    x = torch.randn(shape, device=dev).bfloat16()
    taps = (torch.randn(27, shape[-1], device=dev) * 0.2).contiguous()
    bias = torch.randn(shape[-1], device=dev)
    res = {}
    for knob in (0, 1):
        nat.check(nat.lib().pytc_set_tuning(b"dwconv_march_tx16", knob), "set")
        y, st = ops.dwconv3d(x, taps, bias, K=3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.dwconv3d(x, taps, bias, K=3)
        e1.record()
        torch.cuda.synchronize()
        res[knob] = (y, st.sum(1), e0.elapsed_time(e1) / reps * 1e3, st.shape[1])
    same = torch.equal(res[0][0], res[1][0])
    dst = float((res[0][1] - res[1][1]).abs().max() / res[0][1].abs().max())
    print(f"{shape}: 8x8 {res[0][2]:.1f} us ({res[0][3]} slots)  8x16 {res[1][2]:.1f} us ({res[1][3]} slots)  outputs identical {same}  "
          f"statistics rel diff {dst:.2e}")
nat.check(nat.lib().pytc_set_tuning(b"dwconv_march_tx16", 0), "set")
