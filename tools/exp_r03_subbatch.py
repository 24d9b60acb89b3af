"""Round-3 experiment: the full-resolution level of MedNeXt-S run depth-first over sample slices of a window batch
(MedNeXt.l0_subbatch / PYTC_L0_SUBBATCH) so that a slice's level-0 tensors can be re-read from the 256 MiB Infinity Cache, crossed
with the number of window streams and the fused up block.  Whole-volume engine passes (bench.py's step); prints one JSON record per
configuration: seconds per volume, ms per 8-window batch, bit-identity against the first configuration.  Appends every record to
gpurun_out/r03_subbatch.jsonl as it is produced."""
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    from pytorch_connectomics_amd.models.architectures.mednext import MedNeXt
    dev = torch.device("cuda", 0)
    vol_shape = tuple(int(v) for v in os.environ.get("PYTC_BENCH_VOLUME", "165x1024x768").split("x"))
    model = bench.build_model(dev)
    nets = [m for m in model.modules() if isinstance(m, MedNeXt)]
    eng = bench.make_engine()
    g = torch.Generator(device=dev).manual_seed(7)
    vol = torch.rand((1, 1) + vol_shape, device=dev, generator=g)
    _, starts = eng.plan(vol_shape)
    nb = (len(starts) - 1 + 7) // 8
    out_path = ROOT / "gpurun_out" / "r03_subbatch.jsonl"
    out_path.parent.mkdir(exist_ok=True)
    # (l0_subbatch, window streams, fused up block)
    default = [(0, 2, 0), (2, 2, 0), (1, 2, 0), (4, 2, 0), (0, 1, 0), (2, 1, 0), (1, 1, 0), (4, 1, 0), (0, 2, 1), (2, 2, 1),
               (2, 3, 0), (0, 2, 0)]
    spec = os.environ.get("CONFIGS")
    configs = [tuple(int(v) for v in c.split(":")) for c in spec.split(",")] if spec else default
    reps = int(os.environ.get("REPS", "3"))
    ref = None
    best = None
    with torch.no_grad():
        for sb, streams, fuse_up in configs:
            for n in nets:
                n.l0_subbatch = sb
                n._hip.fuse_up = bool(fuse_up)
            eng.pipeline_streams = streams
            eng(vol, model)
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                y = eng(vol, model)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            same = None
            if ref is None:
                ref = y.clone()
            else:
                same = bool(torch.equal(ref, y))
            rec = {"l0_subbatch": sb, "streams": streams, "fuse_up": fuse_up, "s_per_volume": min(ts),
                   "s_per_volume_all": [round(t, 4) for t in ts], "ms_per_8_windows": 1e3 * min(ts) / (nb + 0.3),
                   "bit_identical_to_first": same, "peak_gb": torch.cuda.max_memory_allocated() / 1e9}
            print(json.dumps(rec), flush=True)
            with open(out_path, "a") as f:
                f.write(json.dumps(rec) + "\n")
            if same is not False and fuse_up == 0 and (best is None or rec["s_per_volume"] < best["s_per_volume"]):
                best = rec
            del y
    with open(ROOT / "gpurun_out" / "r03_subbatch_best.json", "w") as f:
        json.dump(best, f)
    print("BEST", json.dumps(best))


if __name__ == "__main__":
    main()
