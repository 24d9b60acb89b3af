"""Round-3 probe: the fused residual data-gradient kernel (dwconv3d_res) and the other z-march launches at BASELINE shapes,
against the un-fused kernels / fp32 math."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

torch.manual_seed(0)
dev = torch.device("cuda")
for N, D, C in ((1, 112, 32), (4, 112, 32), (1, 56, 64), (2, 112, 32), (1, 64, 32), (1, 96, 32), (1, 112, 64)):
    x = torch.randn(N, D, D, D, C, device=dev).bfloat16()
    res = torch.randn(N, D, D, D, C, device=dev).bfloat16()
    taps = (torch.randn(27, C, device=dev) * 0.2).contiguous()
    y_f = ops.dwconv3d_res(x, taps, res, K=3)
    y_c, _ = ops.dwconv3d(x, taps, None, K=3, stride=1, stats=False)
    want = (y_c.float() + res.float())
    d = (y_f.float() - want).abs()
    bad = d > 0.05 * want.abs().clamp_min(1.0)
    print(f"N={N} D={D} C={C}: fused-vs-unfused max |d| {float(d.max()):.4f}  bad elements {int(bad.sum())} of {bad.numel()}")
    if int(bad.sum()):
        idx = bad.nonzero()
        print("   first", idx[0].tolist(), "last", idx[-1].tolist(), "distinct n", idx[:, 0].unique().tolist()[:8], "distinct z", idx[:, 1].unique().tolist()[:40])
    # the plain march against fp32 torch conv on a sub-block (spot check)
    xs = x[:1, :24, :24, :24].float().permute(0, 4, 1, 2, 3)
    w = taps.t().reshape(C, 1, 3, 3, 3)
    ref = torch.nn.functional.conv3d(xs, w, padding=1, groups=C).permute(0, 2, 3, 4, 1)
    got = y_c[:1, 1:22, 1:22, 1:22].float()
    print("   plain march vs torch (interior) max |d|", float((got - ref[:, 1:22, 1:22, 1:22]).abs().max()))
