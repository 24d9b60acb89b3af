"""Round-2 experiments on the GPU box (not imported by the product):
  nsweep   MedNeXt-S forward_cl per-window time at batch 1 / 2 / 4 / 8 (is the 256 MB Infinity Cache worth a depth-first,
           per-window schedule?  one level-0 tensor of one window is 90 MB, of a batch of 8: 720 MB)
  lut      fused mixer GELU variants on the whole forward
"""
import sys
import time
from pathlib import Path
from types import SimpleNamespace as NS

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pytorch_connectomics_amd import _native as nat  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402
from pytorch_connectomics_amd.models import build_model  # noqa: E402

dev = torch.device("cuda:0")


def knob(k, v):
    nat.check(nat.lib().pytc_set_tuning(k.encode(), int(v)), "set_tuning")


def model_s():
    cfg = NS(model=NS(arch=NS(type="mednext"), in_channels=1, out_channels=1, mednext=NS(size="S", kernel_size=3),
                      loss=NS(deep_supervision=False), heads=None))
    torch.manual_seed(0)
    m = build_model(cfg).to(dev).eval()
    m.model.compute_dtype = torch.bfloat16
    return m


def time_forward(m, x, reps=6):
    with torch.no_grad():
        for _ in range(3):
            m.forward_cl(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            m.forward_cl(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def nsweep():
    m = model_s()
    for n in (1, 2, 4, 8, 16):
        x = torch.rand(n, 112, 112, 112, 1, device=dev)
        ms = time_forward(m, x, reps=max(3, 24 // n))
        print(f"nsweep N={n:2d}: {ms:8.3f} ms / forward = {ms / n:7.3f} ms per 112^3 window", flush=True)
    # the same at batch 8, executed as 8 / 4 / 2 sequential sub-batches (what a depth-first engine schedule would do)
    x = torch.rand(8, 112, 112, 112, 1, device=dev)
    for sub in (1, 2, 4, 8):
        with torch.no_grad():
            def run():
                return [m.forward_cl(x[i:i + sub]) for i in range(0, 8, sub)]
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                run()
            torch.cuda.synchronize()
        print(f"nsweep 8 windows as sub-batches of {sub}: {(time.perf_counter() - t0) / 4 * 1e3:8.3f} ms", flush=True)


def lut():
    m = model_s()
    x = torch.rand(8, 112, 112, 112, 1, device=dev)
    for name, kv in (("fast", {}), ("lut", {"mlp_gelu_lut": 1}), ("exact", {"mlp_exact_gelu": 1})):
        for k, v in kv.items():
            knob(k, v)
        print(f"gelu {name}: {time_forward(m, x):8.3f} ms / 8 windows", flush=True)
        for k in kv:
            knob(k, 0)
    for wgs in (512, 1024, 2048, 4096):
        knob("dwconv_march_wgs", wgs)
        print(f"dwconv_march_wgs {wgs}: {time_forward(m, x):8.3f} ms / 8 windows", flush=True)
    knob("dwconv_march_wgs", 1024)


def fwd():
    """MedNeXt-S forward, 8 windows, with whatever knobs are set."""
    m = model_s()
    x = torch.rand(8, 112, 112, 112, 1, device=dev)
    print(f"forward: {time_forward(m, x, reps=10):8.3f} ms / 8 windows", flush=True)


def upfuse():
    m = model_s()
    x = torch.rand(8, 112, 112, 112, 1, device=dev)
    hip = m.model._hip
    for name, on, cin in (("unfused", False, (64, 128)), ("fused L0", True, (64,)), ("fused L0+L1", True, (64, 128))):
        hip.fuse_up, hip.fuse_up_cin = on, cin
        print(f"up blocks {name}: {time_forward(m, x):8.3f} ms / 8 windows", flush=True)
    hip.fuse_up, hip.fuse_up_cin = True, (64, 128)
    with torch.no_grad(), ops.profiled() as prof:
        for _ in range(3):
            m.forward_cl(x)
    for k, v in sorted(prof.summary().items(), key=lambda kv: -kv[1]["ms"])[:14]:
        print(f"   {k:36s} launches={v['launches'] / 3:5.1f} ms/fwd={v['ms'] / 3:7.3f} GB/s={v['bytes'] / max(v['ms'], 1e-9) / 1e6:8.1f}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["nsweep", "lut", "upfuse"]
    for w in which:
        if w.startswith("knob:"):                     # knob:<name>=<int> sets a tuning knob for the experiments that follow
            k, v = w[5:].split("=")
            knob(k, int(v))
            print(f"[knob] {k} = {v}", flush=True)
            continue
        globals()[w]()
