"""Isolated timing of the fused stem + first depthwise conv kernel (pytc_stem_dwconv3d_fwd) at the bench shape, VALU form
(`stem_mfma` = 0) against the f16-MFMA form, and their difference on the same input."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pytorch_connectomics_amd import _native as nat  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402


def knob(k, v):
    nat.check(nat.lib().pytc_set_tuning(k.encode(), int(v)), "set_tuning")


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    N, D, H, W, C = 8, 112, 112, 112, 32
    x = torch.rand(N, D, H, W, 1, device=dev)
    sw, sb = torch.randn(C, device=dev) * 0.5, torch.randn(C, device=dev) * 0.1
    taps = torch.randn(27, C, device=dev) * 0.2
    b1 = torch.randn(C, device=dev) * 0.1
    packed = ops.stem_dwconv3d_pack(sw, sb, taps, b1)
    outs = {}
    for name, v in (("valu", 0), ("mfma", 1)):
        knob("stem_mfma", v)
        for _ in range(3):
            t, st = ops.stem_dwconv3d(x, packed)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            t, st = ops.stem_dwconv3d(x, packed)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"stem_dwconv {name}: {us:8.1f} us  = {t.numel() * 2 / us / 1e6:6.2f} TB/s of output", flush=True)
        outs[name] = (t.float(), st.sum(1))
    d = (outs["valu"][0] - outs["mfma"][0]).abs()
    print(f"max |valu - mfma| = {float(d.max()):.4e} (values up to {float(outs['valu'][0].abs().max()):.2f}), mean {float(d.mean()):.3e};"
          f" stats rel diff {float(((outs['valu'][1] - outs['mfma'][1]).abs() / outs['valu'][1].abs().clamp_min(1e-6)).max()):.3e}")
    knob("stem_mfma", 1)
    for zr in (8, 16, 32, 56, 112):
        knob("stem_mfma_zr", zr)
        for _ in range(3):
            ops.stem_dwconv3d(x, packed)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.stem_dwconv3d(x, packed)
        e1.record()
        torch.cuda.synchronize()
        print(f"mfma, z extent per workgroup {zr:3d}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
    knob("stem_mfma_zr", 0)
    for sp in (0, 1):
        knob("stem_store_permute", sp)
        for _ in range(3):
            t2, _st = ops.stem_dwconv3d(x, packed)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.stem_dwconv3d(x, packed)
        e1.record()
        torch.cuda.synchronize()
        print(f"mfma, store permute {sp}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us  (max diff vs default {float((t2.float() - outs['mfma'][0]).abs().max()):.1e})", flush=True)
    knob("stem_store_permute", 1)
    # what a write-only pass over the same 0.72 GB costs on this chip (torch fill kernel), and a read-only pass (sum)
    buf = torch.empty_like(t, dtype=torch.bfloat16)
    for name, fn in (("fill_ (write only)", lambda: buf.fill_(1.0)), ("sum (read only)", lambda: buf.sum()),
                     ("copy_ (read + write)", lambda: buf.copy_(t.to(torch.bfloat16) if t.dtype != torch.bfloat16 else t))):
        src = t.to(torch.bfloat16)
        if name.startswith("copy"):
            fn = lambda: buf.copy_(src)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{name:22s}: {us:8.1f} us = {buf.numel() * 2 / us / 1e6:6.2f} TB/s per direction", flush=True)


if __name__ == "__main__":
    main()
