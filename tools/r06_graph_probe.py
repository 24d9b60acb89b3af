"""Round-6 probe: is the window pipeline host-bound, and does a captured (hipGraph) batch forward help?

  (1) product engine call as bench.py times it (3 window streams)
  (2) host enqueue time of the batch forward against its GPU time (one stream, no sync inside the loop)
  (3) the same forward captured in a hipGraph, replayed on one stream and on three streams round-robin
"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    model = bench.build_model(dev)
    eng = bench.make_engine()
    g = torch.Generator(device=dev).manual_seed(7)
    vol = torch.rand((1, 1) + bench.VOLUME, device=dev, generator=g)
    _, starts = eng.plan(bench.VOLUME)
    n_b = (len(starts) - 1 + 7) // 8
    with torch.no_grad():
        for _ in range(2):
            eng(vol, model)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng(vol, model)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"engine: {dt * 1e3:.1f} ms per volume, {dt * 1e3 / len(starts) * 8:.3f} ms per 8 windows", flush=True)

        fwd = model.forward_cl
        x = ops.gather_windows(vol[0], starts[1:9], bench.ROI, view=0, pad_mode="constant", cval=0.0)
        for _ in range(3):
            y_ref = fwd(x)
        torch.cuda.synchronize()
        n = 12
        t0 = time.perf_counter()
        for _ in range(n):
            y = fwd(x)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"eager one stream: host enqueue {t_host / n * 1e3:.3f} ms per batch, wall {t_all / n * 1e3:.3f} ms per batch", flush=True)

        # --- hipGraph capture of the batch forward
        lanes = [torch.cuda.Stream(device=dev) for _ in range(3)]
        graphs = []
        for s in lanes:
            xs = x.clone()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    fwd(xs)
            s.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                ys = fwd(xs)
            graphs.append((gr, xs, ys))
        torch.cuda.synchronize()
        gr, xs, ys = graphs[0]
        gr.replay()
        torch.cuda.synchronize()
        print("graph output equals eager bits:", bool(torch.equal(ys, y_ref)), flush=True)
        t0 = time.perf_counter()
        for _ in range(n):
            gr.replay()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"graph one stream: host {t_host / n * 1e3:.3f} ms per batch, wall {t_all / n * 1e3:.3f} ms per batch", flush=True)
        for k in (2, 3):
            t0 = time.perf_counter()
            for i in range(4 * n):
                s = lanes[i % k]
                with torch.cuda.stream(s):
                    graphs[i % k][0].replay()
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            print(f"graph {k} streams round-robin: wall {t_all / (4 * n) * 1e3:.3f} ms per batch", flush=True)
        # eager on k streams round-robin for the same comparison (no gather / blend)
        for k in (2, 3):
            t0 = time.perf_counter()
            for i in range(4 * n):
                s = lanes[i % k]
                with torch.cuda.stream(s):
                    fwd(graphs[i % k][1])
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            print(f"eager {k} streams round-robin: host {t_host / (4 * n) * 1e3:.3f}  wall {t_all / (4 * n) * 1e3:.3f} ms per batch", flush=True)


if __name__ == "__main__":
    main()
