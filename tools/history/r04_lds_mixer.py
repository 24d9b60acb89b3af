"""Round 4: the LDS-resident persistent mixer (pytc_pw_mlp_lds_fwd) against the streaming one (pytc_pw_mlp_fwd) at the three mid-level
shapes, every launch variant (knob mlp_lds_variant), folded and affine operands; bit-identity is asserted before anything is timed.

    python tools/history/r04_lds_mixer.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pytorch_connectomics_amd import _native as nat  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16


def knob(k, v):
    nat.check(nat.lib().pytc_set_tuning(k.encode(), int(v)), "set_tuning")


def timeit(fn, reps=20, warm=8):
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


def case(N, D, cin, chid, cout, mode, folded):
    rows = D ** 3
    torch.manual_seed(1)
    t = torch.randn(N, rows, cin, device=dev).to(bf)
    w2f = torch.randn(chid, cin, device=dev) / cin ** 0.5
    w3 = ops.pw_pack_weight_paired(torch.randn(cout, chid, device=dev) / chid ** 0.5, f16=True)
    b3 = torch.randn(cout, device=dev)
    if folded:
        ab = None
        w2 = torch.stack([ops.pw_pack_weight_paired(w2f * (1 + 0.1 * n)) for n in range(N)])
        b2 = torch.randn(N, chid, device=dev)
    else:
        ab = torch.rand(N, 2, cin, device=dev)
        w2 = ops.pw_pack_weight_paired(w2f)
        b2 = torch.randn(chid, device=dev)
    res = torch.randn(N, rows, cout, device=dev).to(bf)
    kw = dict(N=N, rows_per_sample=rows, c_in=cin, c_hid=chid, c_out=cout)
    if mode == "add":
        kw.update(res=res, res_mode=nat.RES_ADD)
    elif mode == "up":
        low = torch.randn(N, (D // 2) ** 3, cout, device=dev).to(bf)
        kw.update(res=res, res_mode=nat.RES_UPSAMPLE, grid=(D, D, D), res_low=low, res_bias=b3)
    y0 = torch.empty(N, rows, cout, device=dev, dtype=bf)
    y1 = torch.empty_like(y0)
    ops.pw_mlp(t, ab, w2, b2, w3, b3, y=y0, **kw)
    base = timeit(lambda: ops.pw_mlp(t, ab, w2, b2, w3, b3, y=y0, **kw))
    line = f"{cin}->{chid}->{cout} {D}^3 x{N} {mode:4s} {'fold' if folded else 'ab  '}: stream {base:7.1f} us | lds"
    for v in (1, 2, 3, 4):
        knob("mlp_lds_variant", v)
        y1.zero_()
        ops.pw_mlp(t, ab, w2, b2, w3, b3, y=y1, lds=True, **kw)
        assert torch.equal(y0, y1), (cin, chid, cout, mode, folded, v, (y0.float() - y1.float()).abs().max().item())
        us = timeit(lambda: ops.pw_mlp(t, ab, w2, b2, w3, b3, y=y1, lds=True, **kw))
        line += f"  v{v} {us:7.1f}"
    knob("mlp_lds_variant", 0)
    print(line, flush=True)


if __name__ == "__main__":
    case(8, 112, 64, 128, 32, "up", True)
    for folded in (True, False):
        case(8, 56, 64, 128, 64, "add", folded)
        case(8, 56, 128, 256, 64, "up", folded)
        case(8, 28, 128, 256, 128, "add", folded)
    case(3, 18, 64, 128, 32, "up", False)
    case(3, 19, 64, 128, 64, "add", True)       # ragged tail, shares crossing samples
    case(5, 14, 128, 256, 64, "up", True)
    case(1, 56, 64, 128, 64, "none", True)
