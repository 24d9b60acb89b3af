"""BASELINE configs[2]'s inference leg (bench.py: c3_affinity_tta16_leg) on its own, for profiling:
    python tools/r06_c3_leg.py            one timed pass, the bench record as JSON
    python tools/r06_c3_leg.py --host     the same under cProfile: where the HOST time goes (top 35 by cumulative time)
    python tools/r06_c3_leg.py --labels   HIP-event time per profiler label of one pass (one stream)
rocprofv3 wraps the first form (tools/r06_call1.sh)."""
import cProfile
import io
import json
import os
import pstats
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    model3 = bench.build_model(dev, out_channels=3)
    if "--host" in sys.argv:
        bench.c3_affinity_tta16_leg(dev, model3)            # warm everything
        pr = cProfile.Profile()
        pr.enable()
        rec = bench.c3_affinity_tta16_leg(dev, model3)
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
        print(s.getvalue()[:9000])
    else:
        rec = bench.c3_affinity_tta16_leg(dev, model3)
        if "--twice" in sys.argv:
            rec = bench.c3_affinity_tta16_leg(dev, model3)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
