"""Where a chunk of the C4 leg spends its time: staging of the host box, window loop, finalize + activation + crop (synchronised phases)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from pytorch_connectomics_amd.inference import lazy as lz  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
T = {}
real_stage, real_lanes = lz._stage_host_box, lz._window_lanes
real_fin = lz.ops.blend_finalize


def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w


lz._stage_host_box = timed("stage", real_stage)
lz.ops.blend_finalize = timed("finalize", real_fin)
marks = {}


def lanes(*a, **k):
    torch.cuda.synchronize()
    marks["loop0"] = time.perf_counter()
    return real_lanes(*a, **k)


lz._window_lanes = lanes
real_lsw = lz._lazy_sliding_window.__wrapped__ if hasattr(lz._lazy_sliding_window, "__wrapped__") else None
t_all0 = time.perf_counter()
rec = bench.c4_chunked_leg(dev)
print({k: round(v, 4) for k, v in T.items()}, "seconds over 3 chunks (1 warm-up + 2 timed)")
print("seconds per chunk (timed two):", rec["seconds_per_chunk"], "window_voxels_per_s", rec["window_voxels_per_s"])
