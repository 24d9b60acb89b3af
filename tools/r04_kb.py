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


TARGETS = {"copy0": lambda: copy(8, 112, 32), "copy64": lambda: copy(8, 112, 64), "mix0": lambda: mixer(8, 112, 32, 64, 32, "add"), "up0": lambda: mixer(8, 112, 64, 128, 32, "up"),
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
