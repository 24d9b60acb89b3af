"""Kernel micro-benchmarks at the BASELINE shapes (MedNeXt-S, 8 windows of 112^3, bf16) with A/B
switching of kernel variants through pytc_set_tuning.  Run on the GPU box:

    python tools/kbench.py [dwconv] [mlp] [convT] ...
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pytorch_connectomics_amd import _native as nat  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16


def timeit(fn, reps=10, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3   # us


def knob(k, v):
    nat.check(nat.lib().pytc_set_tuning(k.encode(), int(v)), "set_tuning")


def bench_dwconv():
    for (N, D, C) in ((8, 112, 32), (8, 56, 64), (8, 28, 128)):
        x = torch.randn(N, D, D, D, C, device=dev).to(bf)
        taps = torch.randn(27, C, device=dev)
        b = torch.randn(C, device=dev)
        nbytes = 2 * x.numel() * 2
        knob("dwconv_march_variant", 2)
        ref, rst = ops.dwconv3d(x, taps, b, K=3)
        for var in (0, 1, 2, 3):
            knob("dwconv_march_variant", var)
            y, st = ops.dwconv3d(x, taps, b, K=3)
            print(f"variant {var}: bit-exact vs variant 2: y={bool(torch.equal(y, ref))} stats={bool(torch.equal(st, rst))}")
            for wgs in (2048, 4096, 6144, 8192):
                knob("dwconv_march_variant", var)
                knob("dwconv_march_wgs", wgs)
                us = timeit(lambda: ops.dwconv3d(x, taps, b, K=3))
                print(f"dwconv3d N{N} {D}^3 C{C} variant{var} target_wgs={wgs}: "
                      f"{us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s", flush=True)
        knob("dwconv_march_variant", 0)
        knob("dwconv_march_wgs", 4096)


def bench_mlp():
    for (N, D, cin, chid, cout, mode) in ((8, 112, 32, 64, 32, "add"), (8, 112, 64, 128, 32, "up"),
                                          (8, 56, 64, 128, 64, "add"), (8, 28, 128, 256, 128, "add"),
                                          (8, 14, 256, 512, 256, "add"), (8, 7, 512, 1024, 512, "add")):
        rows = D ** 3
        t = torch.randn(N, rows, cin, device=dev).to(bf)
        ab = torch.rand(N, 2, cin, device=dev)
        w2 = ops.pw_pack_weight_paired(torch.randn(chid, cin, device=dev) / cin ** 0.5)
        w3 = ops.pw_pack_weight_paired(torch.randn(cout, chid, device=dev) / chid ** 0.5)
        b2, b3 = torch.randn(chid, device=dev), torch.randn(cout, device=dev)
        res = torch.randn(N, rows, cout, device=dev).to(bf)
        y = torch.empty(N, rows, cout, device=dev, dtype=bf)
        kw = dict(N=N, rows_per_sample=rows, c_in=cin, c_hid=chid, c_out=cout, y=y)
        if mode == "add":
            fn = lambda: ops.pw_mlp(t, ab, w2, b2, w3, b3, res=res, res_mode=nat.RES_ADD, **kw)
        else:
            low = torch.randn(N, (D // 2) ** 3, cout, device=dev).to(bf)
            fn = lambda: ops.pw_mlp(t, ab, w2, b2, w3, b3, res=res, res_mode=nat.RES_UPSAMPLE, grid=(D, D, D),
                                    res_low=low, res_bias=b3, **kw)
        nbytes = N * rows * 2 * (cin + 2 * cout)
        flops = 2 * N * rows * (cin * chid + chid * cout)
        for var, exact in ((0, 0), (1, 0), (2, 0), (0, 1)):
            knob("mlp_variant", var)
            knob("mlp_exact_gelu", exact)
            us = timeit(fn)
            print(f"pw_mlp {cin}->{chid}->{cout} {D}^3 {mode} variant{var} exact_gelu={exact}: {us:8.1f} us  "
                  f"{nbytes / us / 1e3:7.1f} GB/s  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
        knob("mlp_variant", 0)
        knob("mlp_exact_gelu", 0)


def bench_dw_wgrad():
    N = 4
    for (D, C) in ((112, 32), (56, 64)):
        g = torch.randn(N, D, D, D, C, device=dev).to(bf)
        x = torch.randn(N, D, D, D, C, device=dev).to(bf)
        us = timeit(lambda: ops.dw_wgrad(g, x, K=3, stride=1))
        print(f"dw_wgrad march N{N} {D}^3 C{C}: {us:8.1f} us  {2 * g.numel() * 2 / us / 1e3:7.1f} GB/s", flush=True)
        gs = torch.randn(N, D // 2, D // 2, D // 2, C, device=dev).to(bf)
        us = timeit(lambda: ops.dw_wgrad(gs, x, K=3, stride=2))
        print(f"dw_wgrad stride2 N{N} {D}^3->{D // 2}^3 C{C}: {us:8.1f} us  {(gs.numel() + x.numel()) * 2 / us / 1e3:7.1f} GB/s", flush=True)


def bench_mlp_up():
    """Up-block mixer shapes: cost of the up-sampling epilogue vs a plain residual add vs no residual."""
    for (N, D, cin, chid, cout) in ((8, 56, 128, 256, 64), (8, 28, 256, 512, 128), (8, 112, 64, 128, 32)):
        rows = D ** 3
        t = torch.randn(N, rows, cin, device=dev).to(bf)
        ab = torch.rand(N, 2, cin, device=dev)
        w2 = ops.pw_pack_weight_paired(torch.randn(chid, cin, device=dev) / cin ** 0.5)
        w3 = ops.pw_pack_weight_paired(torch.randn(cout, chid, device=dev) / chid ** 0.5)
        b2, b3 = torch.randn(chid, device=dev), torch.randn(cout, device=dev)
        res = torch.randn(N, rows, cout, device=dev).to(bf)
        low = torch.randn(N, (D // 2) ** 3, cout, device=dev).to(bf)
        y = torch.empty(N, rows, cout, device=dev, dtype=bf)
        kw = dict(N=N, rows_per_sample=rows, c_in=cin, c_hid=chid, c_out=cout, y=y)
        for wl in (0, 1):
            knob("mlp_gelu_lut", wl)
            a = timeit(lambda: ops.pw_mlp(t, ab, w2, b2, w3, b3, **kw))
            b = timeit(lambda: ops.pw_mlp(t, ab, w2, b2, w3, b3, res=res, res_mode=nat.RES_ADD, **kw))
            c = timeit(lambda: ops.pw_mlp(t, ab, w2, b2, w3, b3, res=res, res_mode=nat.RES_UPSAMPLE, grid=(D, D, D),
                                          res_low=low, res_bias=b3, **kw))
            print(f"pw_mlp {cin}->{chid}->{cout} {D}^3 gelu_lut={wl}: none {a:7.1f} us, add {b:7.1f} us, up {c:7.1f} us", flush=True)
    knob("mlp_gelu_lut", 1)


def bench_mlp_cold():
    """Deep-level mixers with COLD weights/activations (a 1 GiB copy runs between launches), as inside the network."""
    big = torch.empty(1 << 28, device=dev, dtype=torch.float32)
    big2 = torch.empty_like(big)
    for (N, D, cin, chid, cout) in ((8, 28, 128, 256, 128), (8, 14, 256, 512, 256), (8, 7, 512, 1024, 512)):
        rows = D ** 3
        t = torch.randn(N, rows, cin, device=dev).to(bf)
        ab = torch.rand(N, 2, cin, device=dev)
        w2 = ops.pw_pack_weight_paired(torch.randn(chid, cin, device=dev) / cin ** 0.5)
        w3 = ops.pw_pack_weight_paired(torch.randn(cout, chid, device=dev) / chid ** 0.5)
        b2, b3 = torch.randn(chid, device=dev), torch.randn(cout, device=dev)
        res = torch.randn(N, rows, cout, device=dev).to(bf)
        y = torch.empty(N, rows, cout, device=dev, dtype=bf)
        kw = dict(N=N, rows_per_sample=rows, c_in=cin, c_hid=chid, c_out=cout, y=y)
        fn = lambda: ops.pw_mlp(t, ab, w2, b2, w3, b3, res=res, res_mode=nat.RES_ADD, **kw)
        hot = timeit(fn)
        tot = 0.0
        for _ in range(5):
            big2.copy_(big)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record()
            torch.cuda.synchronize()
            tot += s.elapsed_time(e) * 1e3
        print(f"pw_mlp {cin}->{chid}->{cout} {D}^3: hot {hot:7.1f} us, cold {tot / 5:7.1f} us", flush=True)


def bench_convT():
    N, D, C = 8, 56, 64
    x = torch.randn(N, D, D, D, C, device=dev).to(bf)
    taps = torch.randn(27, C, device=dev)
    b = torch.randn(C, device=dev)
    nbytes = x.numel() * 2 * 9
    knob("dwconvT_cell", 0)
    y0, s0 = ops.dwconv3d(x, taps, b, K=3, transposed=True)
    us = timeit(lambda: ops.dwconv3d(x, taps, b, K=3, transposed=True))
    print(f"dwconvT3d (per-output kernel) N{N} {D}^3->{2 * D}^3 C{C}: {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s", flush=True)
    knob("dwconvT_cell", 1)
    y1, s1 = ops.dwconv3d(x, taps, b, K=3, transposed=True)
    us = timeit(lambda: ops.dwconv3d(x, taps, b, K=3, transposed=True))
    print(f"dwconvT3d (cell kernel)       N{N} {D}^3->{2 * D}^3 C{C}: {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s  "
          f"y bit-exact={bool(torch.equal(y0, y1))} stats close={bool(torch.allclose(s0.sum(1), s1.sum(1), rtol=1e-4))}", flush=True)
    x = torch.randn(N, 2 * D, 2 * D, 2 * D, 32, device=dev).to(bf)
    taps = torch.randn(27, 32, device=dev)
    for gk in (0, 1):
        knob("dwconv_gather", gk)
        us = timeit(lambda: ops.dwconv3d(x, taps, b[:32].contiguous(), K=3, stride=2))
        print(f"dwconv3d s2 N{N} {2 * D}^3->{D}^3 C32 branchless={gk}: {us:8.1f} us", flush=True)
    x = torch.randn(N, 14, 14, 14, 256, device=dev).to(bf)
    taps = torch.randn(27, 256, device=dev)
    b = torch.randn(256, device=dev)
    for gk in (0, 1):
        knob("dwconv_gather", gk)
        us = timeit(lambda: ops.dwconv3d(x, taps, b, K=3))
        print(f"dwconv3d N{N} 14^3 C256 branchless={gk}: {us:8.1f} us", flush=True)


def bench_copy():
    a = torch.empty(8 * 112 ** 3 * 32, device=dev, dtype=bf)
    b = torch.empty_like(a)
    us = timeit(lambda: b.copy_(a))
    print(f"torch copy 0.72 GB: {us:8.1f} us  {2 * a.numel() * 2 / us / 1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["copy", "dwconv", "mlp", "convT"]
    torch.manual_seed(0)
    for wname in which:
        {"dwconv": bench_dwconv, "mlp": bench_mlp, "mlp_cold": bench_mlp_cold, "dw_wgrad": bench_dw_wgrad, "mlp_up": bench_mlp_up, "convT": bench_convT, "copy": bench_copy}[wname]()
