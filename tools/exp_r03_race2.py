"""Round-3 probe: where does the 2-stream engine pass differ from the 1-stream pass?"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


class Ident:
    def forward_cl(self, x):
        return (x * 0.5 + 0.25).contiguous()


def diff(a, b, tag):
    d = (a - b).abs()
    nz = (d > 0)
    n = int(nz.sum())
    print(tag, "differing voxels", n, "max", float(d.max()))
    if n:
        idx = nz[0, 0].nonzero()
        print("   bbox z", int(idx[:, 0].min()), int(idx[:, 0].max()), "y", int(idx[:, 1].min()), int(idx[:, 1].max()),
              "x", int(idx[:, 2].min()), int(idx[:, 2].max()))
        rel = d[nz] / a[nz].abs().clamp_min(1e-6)
        print("   rel max", float(rel.max()), "median", float(rel.median()))


def main():
    dev = torch.device("cuda", 0)
    shape = (165, 448, 448)
    model = bench.build_model(dev)
    eng = bench.make_engine()
    g = torch.Generator(device=dev).manual_seed(7)
    vol = torch.rand((1, 1) + shape, device=dev, generator=g)
    with torch.no_grad():
        for net, name in ((Ident(), "ident"), (model, "mednext")):
            eng.pipeline_streams = 1
            a = eng(vol, net).clone()
            a2 = eng(vol, net).clone()
            diff(a, a2, f"{name}: 1 vs 1")
            for n in (2, 2, 4):
                eng.pipeline_streams = n
                b = eng(vol, net).clone()
                torch.cuda.synchronize()
                diff(a, b, f"{name}: 1 vs {n}")


if __name__ == "__main__":
    main()
