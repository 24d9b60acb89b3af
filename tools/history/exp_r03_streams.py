"""Round-3 experiment: window batches spread over 1..4 HIP streams (EagerSlidingWindowEngine.pipeline_streams).
Prints seconds per whole-volume pass, ms per 8-window batch, whether the result is bit-identical to the one-stream pass,
and the host-side issue time of one pass (no sync)."""
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    vol_shape = tuple(int(v) for v in os.environ.get("PYTC_BENCH_VOLUME", "165x1024x768").split("x"))
    model = bench.build_model(dev)
    eng = bench.make_engine()
    g = torch.Generator(device=dev).manual_seed(7)
    vol = torch.rand((1, 1) + vol_shape, device=dev, generator=g)
    _, starts = eng.plan(vol_shape)
    nb = (len(starts) - 1 + 7) // 8
    ref = None
    out = {}
    with torch.no_grad():
        for n in [int(v) for v in os.environ.get("STREAMS", "1,2,3,4,1,2").split(",")]:
            eng.pipeline_streams = n
            eng(vol, model)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                y = eng(vol, model)
                t_issue = time.perf_counter() - t0
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            same = None
            if ref is None:
                ref = y.clone()
            else:
                same = bool(torch.equal(ref, y))
            rec = {"streams": n, "s_per_volume": min(ts), "ms_per_8_windows": 1e3 * min(ts) / (nb + 0.3),
                   "host_issue_s": t_issue, "bit_identical_to_first": same}
            print(json.dumps(rec), flush=True)
            out.setdefault(str(n), []).append(rec)
            del y
    print(json.dumps(out))


if __name__ == "__main__":
    main()
