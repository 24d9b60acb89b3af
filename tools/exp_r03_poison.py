"""Round-3 probe: does any kernel of the bf16 forward read memory it did not write?  Every torch.empty / empty_like of
hip_ops is replaced by a NaN-filled (or 1e30-filled) allocation and each op output is compared with the clean run."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

REC = None
FILL = None


class TorchProxy:
    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def empty(*a, **k):
        t = torch.empty(*a, **k)
        if FILL is not None and t.is_floating_point():
            t.fill_(FILL)
        return t

    @staticmethod
    def empty_like(x, **k):
        t = torch.empty_like(x, **k)
        if FILL is not None and t.is_floating_point():
            t.fill_(FILL)
        return t


def wrap(name):
    orig = getattr(ops, name)

    def f(*a, **k):
        out = orig(*a, **k)
        if REC is not None:
            outs = out if isinstance(out, tuple) else (out,)
            for j, o in enumerate(outs):
                if isinstance(o, torch.Tensor):
                    REC.append((f"{name}#{j}", tuple(o.shape), o.clone()))
        return out
    setattr(ops, name, f)


def main():
    global REC, FILL
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    ops.torch = TorchProxy()
    for n in ("dwconv3d", "groupnorm_finalize", "pw_mlp", "pw_mlp_head", "pw_mlp_stemres", "stem_dwconv3d", "pw_conv"):
        wrap(n)
    g = torch.Generator(device=dev).manual_seed(3)
    for N in (8, 3, 1):
        xa = torch.rand((N, 112, 112, 112, 1), device=dev, generator=g)
        with torch.no_grad():
            FILL = 0.0
            REC = []
            ya = model.model.forward_cl(xa); torch.cuda.synchronize()
            clean = REC
            for fill in (float("nan"), 1e30, 3.0):
                FILL = fill
                REC = []
                yb = model.model.forward_cl(xa); torch.cuda.synchronize()
                bad = [(i, a[0], a[1], int((~((a[2] == b[2]) | (a[2].isnan() & b[2].isnan()))).sum()))
                       for i, (a, b) in enumerate(zip(clean, REC)) if not torch.equal(a[2], b[2])]
                print(f"N={N} fill={fill}: output identical {torch.equal(ya, yb)}; differing ops {len(bad)} of {len(clean)}")
                for b in bad[:5]:
                    print("   ", b)


if __name__ == "__main__":
    main()
