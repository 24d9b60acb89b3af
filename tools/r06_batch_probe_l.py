"""MedNeXt-L (3 MitoEM heads, bf16) forward time against the number of 160^3 windows per call, and calls of 2 on 2..6 streams."""
import sys
import time
from pathlib import Path
from types import SimpleNamespace as NS

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from pytorch_connectomics_amd.models import build_model as bm
    heads = {"aff_r1": {"out_channels": 3, "num_blocks": 1, "hidden_channels": 8},
             "aff_r5": {"out_channels": 3, "num_blocks": 1, "hidden_channels": 8},
             "sdt": {"out_channels": 1, "num_blocks": 1, "hidden_channels": 8}}
    mcfg = NS(model=NS(arch=NS(type="mednext"), in_channels=1, out_channels=7, mednext=NS(size="L", kernel_size=3),
                       loss=NS(deep_supervision=False), heads=heads, primary_head="aff_r1"))
    torch.manual_seed(0)
    model = bm(mcfg).to(dev).eval()
    model.model.compute_dtype = torch.bfloat16
    fwd = model.forward_cl
    with torch.no_grad():
        res = {}
        for n in (1, 2, 4, 6, 8):
            x = torch.rand(n, 160, 160, 160, 1, device=dev)
            for _ in range(3):
                fwd(x)
            torch.cuda.synchronize()
            reps = max(3, 16 // n)
            t0 = time.perf_counter()
            for _ in range(reps):
                fwd(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            res[n] = dt
            print(f"windows per call {n}: {dt * 1e3:7.3f} ms per call, {dt * 1e3 / n * 2:6.3f} ms per 2 windows", flush=True)
        slope = (res[8] - res[2]) / 6
        print(f"fit over 2..8: {slope * 2e3:.3f} ms per 2 windows + {(res[2] - 2 * slope) * 1e3:.3f} ms per call", flush=True)
        for n, k in ((2, 2), (2, 4), (2, 6), (4, 2), (4, 3)):
            lanes = [torch.cuda.Stream(device=dev) for _ in range(k)]
            xs = [torch.rand(n, 160, 160, 160, 1, device=dev) for _ in range(k)]
            torch.cuda.synchronize()
            for i in range(2 * k):
                with torch.cuda.stream(lanes[i % k]):
                    fwd(xs[i % k])
            torch.cuda.synchronize()
            reps = 6 * k
            t0 = time.perf_counter()
            for i in range(reps):
                with torch.cuda.stream(lanes[i % k]):
                    fwd(xs[i % k])
            th = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(f"{n} windows per call on {k} streams: {dt * 1e3 / n * 2:6.3f} ms per 2 windows (host {th / reps * 1e3 / n * 2:.3f})", flush=True)


if __name__ == "__main__":
    main()
