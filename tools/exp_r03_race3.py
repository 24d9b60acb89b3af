"""Round-3 probe: per-batch network outputs inside the engine, 1 stream vs 2 streams (with and without a device sync per batch)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    shape = (165, 448, 448)
    model = bench.build_model(dev)
    eng = bench.make_engine()
    g = torch.Generator(device=dev).manual_seed(7)
    vol = torch.rand((1, 1) + shape, device=dev, generator=g)
    preds = []
    SYNC = [False]
    orig = eng._run_network

    def rn(network, batch):
        y = orig(network, batch)
        preds.append((batch.clone(), y.clone()))
        if SYNC[0]:
            torch.cuda.synchronize()
        return y
    eng._run_network = rn
    with torch.no_grad():
        eng.pipeline_streams = 1
        eng(vol, model); torch.cuda.synchronize()
        base = list(preds); preds.clear()
        for n, sync in ((2, True), (2, False), (2, False)):
            eng.pipeline_streams = n
            SYNC[0] = sync
            eng(vol, model); torch.cuda.synchronize()
            cur = list(preds); preds.clear()
            badx = [i for i, (a, b) in enumerate(zip(base, cur)) if not torch.equal(a[0], b[0])]
            bady = [i for i, (a, b) in enumerate(zip(base, cur)) if not torch.equal(a[1], b[1])]
            print(f"streams {n} sync {sync}: batches {len(cur)}  input differs in {badx}  output differs in {bady}")
            for i in bady[:3]:
                d = (base[i][1].float() - cur[i][1].float()).abs()
                per_win = d.flatten(1).max(1).values.tolist()
                print("   batch", i, "max", float(d.max()), "per-window max", [round(v, 4) for v in per_win])
        # the same batch forwarded alone on a side stream after the pass
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            y = model.model.forward_cl(base[3][0])
        torch.cuda.synchronize()
        print("side-stream alone equals 1-stream:", torch.equal(y, base[3][1]))


if __name__ == "__main__":
    main()
