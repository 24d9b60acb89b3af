"""Round 5: the fused level-0 block (statistics pass + pytc_dwmix_fwd) against the two-launch schedule it replaces, 8 x 112^3 x 32.

    python tools/r05_dwmix_bench.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pytorch_connectomics_amd import _native as nat  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16


def timeit(fn, reps=20, warm=6):
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


def run(N, D, c_hid):
    H = W = D
    x = torch.randn(N, D, H, W, 32, device=dev).to(bf)
    taps, b1 = torch.randn(27, 32, device=dev) * 0.2, torch.randn(32, device=dev)
    gamma, beta = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
    w2, b2 = (torch.randn(c_hid, 32, device=dev) / 32 ** 0.5).contiguous(), torch.randn(c_hid, device=dev) * 0.1
    w3 = ops.pw_pack_weight_paired((torch.randn(32, c_hid, device=dev) / c_hid ** 0.5).contiguous(), f16=True)
    b3 = torch.randn(32, device=dev) * 0.1
    rows = D * H * W
    t = torch.empty_like(x)
    y = torch.empty_like(x)
    kw = dict(N=N, rows_per_sample=rows, c_in=32, c_hid=c_hid, c_out=32)
    _, st = ops.dwconv3d(x, taps, b1, K=3, y=t)
    w2n, b2n = ops.groupnorm_fold_mlp(st, float(rows), gamma, beta, 1e-5, w2, b2)
    a = timeit(lambda: ops.dwconv3d(x, taps, b1, K=3, y=t))
    f = timeit(lambda: ops.groupnorm_fold_mlp(st, float(rows), gamma, beta, 1e-5, w2, b2))
    m = timeit(lambda: ops.pw_mlp(t, None, w2n, b2n, w3, b3, res=x, res_mode=nat.RES_ADD, y=y.view(N, rows, 32), **kw))
    s = timeit(lambda: ops.dwconv3d(x, taps, b1, K=3, store=False))
    d = timeit(lambda: ops.dwmix(x, taps, b1, w2n, b2n, w3, b3, c_hid=c_hid, residual=True, y=y))

    def two():
        _, st_ = ops.dwconv3d(x, taps, b1, K=3, y=t)
        a_, b_ = ops.groupnorm_fold_mlp(st_, float(rows), gamma, beta, 1e-5, w2, b2)
        ops.pw_mlp(t, None, a_, b_, w3, b3, res=x, res_mode=nat.RES_ADD, y=y.view(N, rows, 32), **kw)

    def fused():
        _, st_ = ops.dwconv3d(x, taps, b1, K=3, store=False)
        a_, b_ = ops.groupnorm_fold_mlp(st_, float(rows), gamma, beta, 1e-5, w2, b2)
        ops.dwmix(x, taps, b1, a_, b_, w3, b3, c_hid=c_hid, residual=True, y=y)
    t2, tf = timeit(two), timeit(fused)
    gb = x.numel() * 2 / 1e9
    print(f"{N} x {D}^3 x 32 -> {c_hid} -> 32 ({gb:.2f} GB per tensor): dwconv {a:6.1f} + fold {f:5.1f} + mixer {m:6.1f} us | statistics pass {s:6.1f} + "
          f"fused kernel {d:6.1f} us ({2 * gb / d * 1e3:.0f} GB/s on x + y) | block, back to back: two-launch {t2:6.1f} us, fused {tf:6.1f} us "
          f"({t2 / tf:.2f}x; {3 * gb / tf * 1e3:.0f} GB/s on the 3-pass floor)", flush=True)


if __name__ == "__main__":
    run(8, 112, 64)
    run(8, 112, 64)
    run(2, 160, 96)
    run(8, 56, 64)
