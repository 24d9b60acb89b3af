"""Round-6 probe: MedNeXt-S bf16 forward time against the number of 112^3 windows per network call (one stream), and window batches of
16 on two / three streams -- how much of the 8-window forward is latency-bound launches that do not grow with the batch?"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    model = bench.build_model(dev)
    fwd = model.forward_cl
    with torch.no_grad():
        res = {}
        for n in (2, 4, 8, 12, 16, 24):
            x = torch.rand(n, 112, 112, 112, 1, device=dev)
            for _ in range(3):
                fwd(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = max(4, 64 // n)
            for _ in range(reps):
                fwd(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            res[n] = dt
            print(f"windows per call {n:3d}: {dt * 1e3:7.3f} ms per call, {dt * 1e3 / n * 8:6.3f} ms per 8 windows", flush=True)
        slope = (res[24] - res[8]) / 16
        print(f"fit over 8..24: {slope * 8e3:.3f} ms per 8 windows + {(res[8] - 8 * slope) * 1e3:.3f} ms per call", flush=True)
        for n, k in ((8, 3), (16, 2), (16, 3), (24, 2)):
            lanes = [torch.cuda.Stream(device=dev) for _ in range(k)]
            xs = [torch.rand(n, 112, 112, 112, 1, device=dev) for _ in range(k)]
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
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(f"{n} windows per call on {k} streams: {dt * 1e3 / n * 8:6.3f} ms per 8 windows", flush=True)


if __name__ == "__main__":
    main()
