"""conv3d (LDS-tiled kernel) at the deep shapes of the MONAI-style U-Net: time per launch, and with the kernel's parts switched off
(knob conv_tile_probe: 1 = no matrix loop, 2 = no staging loads; wrong results, measurements only)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=30, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


shapes = [(2, 3, 32, 32, 256, 256, 3), (2, 6, 64, 64, 128, 128, 3), (2, 12, 128, 128, 64, 64, 3), (2, 24, 256, 256, 32, 32, 3)]
if "--rsunet" in sys.argv:      # the stock RSUNet profile's padded widths (18 -> 24, 36 -> 40; 48, 64, 80), depth_2d 1: k133 at full resolution
    shapes = [(2, 18, 256, 256, 24, 24, 1), (2, 18, 128, 128, 40, 40, 3), (2, 18, 64, 64, 48, 48, 3), (2, 18, 32, 32, 64, 64, 3), (2, 18, 16, 16, 80, 80, 3)]
for (N, D, H, W, ci, co, kd) in shapes:
    x = torch.randn(N, D, H, W, ci, device=dev).bfloat16()
    w = torch.randn(co, ci, kd, 3, 3, device=dev) * 0.05
    wp = ops.conv3d_pack_weight(w, torch.bfloat16)
    out = []
    for probe in (0, 1, 2, 3):
        ops.set_tuning("conv_tile_probe", probe)
        out.append(timeit(lambda: ops.conv3d(x, wp, c_out=co, kernel=(kd, 3, 3))))
    ops.set_tuning("conv_tile_probe", 0)
    gf = 2 * N * D * H * W * ci * co * 9 * kd / 1e9
    mb = N * D * H * W * (ci + co) * 2 / 1e6
    print(f"{ci:4d}->{co:4d} k{kd}33 @ {N}x{D}x{H}x{W}: {out[0]:7.1f} us ({gf / out[0] * 1e3 / 1e3:6.1f} TFLOP/s, {mb / out[0] / 1e0 * 1e-3 * 1e3:7.1f} GB/s) | no matrix loop {out[1]:7.1f} | no staging loads {out[2]:7.1f} | no epilogue {out[3]:7.1f}")
