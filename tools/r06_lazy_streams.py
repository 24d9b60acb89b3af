"""Round-6 probe: window batches of the lazy / chunked loop on side streams (PYTC_LAZY_SW_STREAMS): C4 leg of bench.py per setting,
and bit-equality of a region predicted on one stream and on three."""
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    for n in (sys.argv[1:] or ["1", "2", "3", "4", "1", "3"]):
        os.environ["PYTC_LAZY_SW_STREAMS"] = n
        rec = bench.c4_chunked_leg(dev)
        print(json.dumps({"streams": n, "seconds_per_chunk": rec["seconds_per_chunk"], "window_voxels_per_s": rec["window_voxels_per_s"],
                          "mfma_frac": rec["roofline"]["mfma_frac"], "hbm_frac": rec["roofline"]["hbm_frac"]}), flush=True)


if __name__ == "__main__":
    main()
