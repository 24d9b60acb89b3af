"""Round-4 micro-benchmarks of the level-0 / level-1 kernels in the configuration the network runs them in (packed-fp16 GELU +
f16 projection image, whole 8-window batch) -- tools/kbench.py predates the f16 projection.  Targets: mix0 up0 dw0 convT0 mix1 up1.

    python tools/r04_kb.py [targets...]          (PYTC_KB_REPS / PYTC_KB_WARM override the 10 / 5 repetitions)
"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pytorch_connectomics_amd import _native as nat  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
REPS, WARM = int(os.environ.get("PYTC_KB_REPS", "10")), int(os.environ.get("PYTC_KB_WARM", "5"))


def timeit(fn):
    for _ in range(WARM):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REPS):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / REPS * 1e3


def knob(k, v):
    nat.check(nat.lib().pytc_set_tuning(k.encode(), int(v)), "set_tuning")


ZERO = os.environ.get("PYTC_KB_ZERO") == "1"          # zero-filled activations: the DVFS / power probe (same instructions, no toggling)


def mixer(N, D, cin, chid, cout, mode):
    rows = D ** 3
    t = torch.randn(N, rows, cin, device=dev).to(bf)
    if ZERO:
        t.zero_()
    ab = torch.rand(N, 2, cin, device=dev)
    w2 = ops.pw_pack_weight_paired(torch.randn(chid, cin, device=dev) / cin ** 0.5)
    w3 = ops.pw_pack_weight_paired(torch.randn(cout, chid, device=dev) / chid ** 0.5, f16=True)
    b2, b3 = torch.randn(chid, device=dev), torch.randn(cout, device=dev)
    res = torch.randn(N, rows, cout, device=dev).to(bf)
    if ZERO:
        res.zero_()
    y = torch.empty(N, rows, cout, device=dev, dtype=bf)
    kw = dict(N=N, rows_per_sample=rows, c_in=cin, c_hid=chid, c_out=cout, y=y)
    if mode == "none":
        fn = lambda: ops.pw_mlp(t, ab, w2, b2, w3, b3, **kw)      # noqa: E731
        nbytes = N * rows * 2 * (cin + cout)
    elif mode == "add":
        fn = lambda: ops.pw_mlp(t, ab, w2, b2, w3, b3, res=res, res_mode=nat.RES_ADD, **kw)      # noqa: E731
        nbytes = N * rows * 2 * (cin + 2 * cout)
    else:
        low = torch.randn(N, (D // 2) ** 3, cout, device=dev).to(bf)
        fn = lambda: ops.pw_mlp(t, ab, w2, b2, w3, b3, res=res, res_mode=nat.RES_UPSAMPLE, grid=(D, D, D), res_low=low,      # noqa: E731
                                res_bias=b3, **kw)
        nbytes = N * rows * 2 * (cin + 2 * cout) + low.numel() * 2
    us = timeit(fn)
    print(f"pw_mlp {cin}->{chid}->{cout} {D}^3 x{N} {mode}: {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s", flush=True)


def copy(N, D, C):
    x = torch.randn(N, D, D, D, C, device=dev).to(bf)
    if ZERO:
        x.zero_()
    y = torch.empty_like(x)
    us = timeit(lambda: y.copy_(x))
    print(f"copy_ bf16 {D}^3 x{N} C{C}: {us:8.1f} us  {2 * x.numel() * 2 / us / 1e3:7.1f} GB/s", flush=True)


def dw(N, D, C):
    x = torch.randn(N, D, D, D, C, device=dev).to(bf)
    if ZERO:
        x.zero_()
    taps = torch.randn(27, C, device=dev)
    b = torch.randn(C, device=dev)
    us = timeit(lambda: ops.dwconv3d(x, taps, b, K=3))
    print(f"dwconv3d k3 {D}^3 x{N} C{C}: {us:8.1f} us  {2 * x.numel() * 2 / us / 1e3:7.1f} GB/s", flush=True)


def convT(N, D, C):
    x = torch.randn(N, D, D, D, C, device=dev).to(bf)
    taps = torch.randn(27, C, device=dev)
    b = torch.randn(C, device=dev)
    us = timeit(lambda: ops.dwconv3d(x, taps, b, K=3, transposed=True))
    print(f"dwconvT3d k3 {D}^3 -> {2 * D}^3 x{N} C{C}: {us:8.1f} us  {9 * x.numel() * 2 / us / 1e3:7.1f} GB/s", flush=True)


def deep(N, D, cin, chid, cout):
    """fused mixer against the two-GEMM schedule of the deep levels at one shape (both bit-identical)"""
    rows = D ** 3
    t = torch.randn(N, rows, cin, device=dev).to(bf)
    ab = torch.rand(N, 2, cin, device=dev)
    w2f, w3f = torch.randn(chid, cin, device=dev) / cin ** 0.5, torch.randn(cout, chid, device=dev) / chid ** 0.5
    b2, b3 = torch.randn(chid, device=dev), torch.randn(cout, device=dev)
    res = torch.randn(N, rows, cout, device=dev).to(bf)
    w2p, w3p = ops.pw_pack_weight_paired(w2f), ops.pw_pack_weight_paired(w3f, f16=True)
    w2, w3 = w2f.to(bf).contiguous(), w3f.to(torch.float16).contiguous()
    kw = dict(N=N, rows_per_sample=rows)
    fused = lambda: ops.pw_mlp(t, ab, w2p, b2, w3p, b3, res=res, res_mode=nat.RES_ADD, c_in=cin, c_hid=chid, c_out=cout, **kw)      # noqa: E731
    flops = 2 * N * rows * (cin * chid + chid * cout)
    uf = timeit(fused)
    line = f"deep {cin}->{chid}->{cout} {D}^3 x{N} ({N * rows} rows): fused {uf:7.1f} us ({flops / uf / 1e6:6.1f} TFLOP/s)"
    for knob_rows in (64, 128):
        knob("pw_gemm_rows", knob_rows)
        g1 = lambda: ops.pw_gemm(t, w2, b2, ab=ab, gelu=True, **kw)      # noqa: E731
        h = g1()
        g2 = lambda: ops.pw_gemm(h, w3, b3, res=res, res_mode=nat.RES_ADD, **kw)      # noqa: E731
        u1, u2 = timeit(g1), timeit(g2)
        line += f" | BR{knob_rows}: {u1:6.1f} + {u2:6.1f} = {u1 + u2:6.1f} us ({flops / (u1 + u2) / 1e6:6.1f} TFLOP/s)"
    knob("pw_gemm_rows", 0)
    print(line, flush=True)


def gemm_sweep():
    """one GEMM launch at the L3-up expand shape (21 952 rows -> 1024 channels) against K, with / without affine, GELU epilogue"""
    N, rows, cout = 8, 14 ** 3, 1024
    for cin in (64, 128, 256, 512, 1024):
        x = torch.randn(N, rows, cin, device=dev).to(bf)
        w = (torch.randn(cout, cin, device=dev) / cin ** 0.5).to(bf).contiguous()
        b = torch.randn(cout, device=dev)
        ab = torch.rand(N, 2, cin, device=dev)
        kw = dict(N=N, rows_per_sample=rows)
        line = f"gemm {rows * N} x {cin} -> {cout}:"
        for label, fn in (("gelu+affine", lambda: ops.pw_gemm(x, w, b, ab=ab, gelu=True, **kw)),
                          ("gelu", lambda: ops.pw_gemm(x, w, b, gelu=True, **kw)),
                          ("plain bf16 out", lambda: ops.pw_gemm(x, w, b, **kw))):
            us = timeit(fn)
            line += f"  {label} {us:6.1f} us ({2 * N * rows * cin * cout / us / 1e6:6.1f} TF)"
        print(line, flush=True)


TARGETS = {"gemm_sweep": gemm_sweep, "deep4": lambda: deep(8, 7, 512, 1024, 512), "deep3": lambda: deep(8, 14, 256, 512, 256),
           "deep3up": lambda: deep(8, 14, 512, 1024, 256), "deep2": lambda: deep(8, 28, 128, 256, 128),
           "deep2up": lambda: deep(8, 28, 256, 512, 128), "deep4dn": lambda: deep(8, 7, 256, 512, 512), "deepL3": lambda: deep(2, 20, 256, 2048, 256), "deepL4": lambda: deep(2, 10, 512, 4096, 512),
           "copy0": lambda: copy(8, 112, 32), "copy64": lambda: copy(8, 112, 64), "mix0": lambda: mixer(8, 112, 32, 64, 32, "add"), "up0": lambda: mixer(8, 112, 64, 128, 32, "up"),
           "dw0": lambda: dw(8, 112, 32), "convT0": lambda: convT(8, 56, 64),
           "mix1": lambda: mixer(8, 56, 64, 128, 64, "add"), "up1": lambda: mixer(8, 56, 128, 256, 64, "up"),
           "dw1": lambda: dw(8, 56, 64)}

if __name__ == "__main__":
    args = sys.argv[1:]
    for a in [a for a in args if "=" in a]:
        knob(*a.split("="))
    for name in [a for a in args if "=" not in a] or list(TARGETS):
        if name.startswith("mixer:"):            # mixer:N,D,cin,chid,cout,mode
            f = name[6:].split(",")
            mixer(*(int(v) for v in f[:5]), f[5])
        else:
            TARGETS[name]()
