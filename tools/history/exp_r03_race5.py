"""Round-3 probe: alternate window batches over two HIP streams (as the engine does) and find the first op whose output
differs from the same batch's quiet forward."""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

REC = None
CLONE = os.environ.get("CLONE", "0") == "1"
NAMES = ("dwconv3d", "groupnorm_finalize", "pw_mlp", "pw_mlp_head", "pw_mlp_stemres", "stem_dwconv3d", "pw_conv")


def wrap(name):
    orig = getattr(ops, name)

    def f(*a, **k):
        out = orig(*a, **k)
        if REC is not None:
            outs = out if isinstance(out, tuple) else (out,)
            for j, o in enumerate(outs):
                if isinstance(o, torch.Tensor):
                    REC.append((f"{name}#{j}", tuple(o.shape), o.clone() if CLONE else o))
        return out
    setattr(ops, name, f)


def main():
    global REC
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    for n in NAMES:
        wrap(n)
    NB, N = int(os.environ.get("NB", "6")), int(os.environ.get("N", "4"))
    g = torch.Generator(device=dev).manual_seed(3)
    xs = [torch.rand((N, 112, 112, 112, 1), device=dev, generator=g) for _ in range(NB)]
    lanes = [torch.cuda.Stream(), torch.cuda.Stream()]
    with torch.no_grad():
        model.model.forward_cl(xs[0][:1]); torch.cuda.synchronize()
        quiet = []
        for x in xs:
            REC = []
            model.model.forward_cl(x); torch.cuda.synchronize()
            quiet.append(REC)
        for trial in range(2):
            conc = []
            for s in lanes:
                s.wait_stream(torch.cuda.current_stream())
            for i, x in enumerate(xs):
                with torch.cuda.stream(lanes[i % 2]):
                    REC = []
                    model.model.forward_cl(x)
                    conc.append(REC)
            REC = None
            torch.cuda.synchronize()
            for i in range(NB):
                bad = [(j, a[0], a[1]) for j, (a, b) in enumerate(zip(quiet[i], conc[i])) if not torch.equal(a[2], b[2])]
                print(f"trial {trial} batch {i}: differing ops {len(bad)} of {len(quiet[i])}", bad[:3])
                if bad:
                    j = bad[0][0]
                    a, b = quiet[i][j][2].float(), conc[i][j][2].float()
                    d = (a - b).abs()
                    nz = (d > 0).nonzero()
                    print("    first op", bad[0][1], "n differing", nz.shape[0], "max", float(d.max()), "first idx", nz[:4].tolist(),
                          "last idx", nz[-2:].tolist())
                    print("    quiet", a[tuple(nz[0].tolist())].item(), "conc", b[tuple(nz[0].tolist())].item())


if __name__ == "__main__":
    main()
