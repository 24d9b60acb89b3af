"""Round-3 probe: does a window batch's forward change when another batch runs concurrently on a second HIP stream?
Records every op output (dwconv t, statistics, affine, block output) of one forward in a quiet GPU and under a concurrent
forward, and prints the first op whose output differs."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

REC = None


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
    global REC
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    for n in ("dwconv3d", "groupnorm_finalize", "pw_mlp", "pw_mlp_head", "pw_mlp_stemres", "stem_dwconv3d", "pw_conv"):
        wrap(n)
    g = torch.Generator(device=dev).manual_seed(3)
    xa = torch.rand((8, 112, 112, 112, 1), device=dev, generator=g)
    xb = torch.rand((8, 112, 112, 112, 1), device=dev, generator=g)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.no_grad():
        model.model.forward_cl(xa[:1]); torch.cuda.synchronize()
        REC = []
        ya = model.model.forward_cl(xa); torch.cuda.synchronize()
        quiet = REC
        REC = []
        ya2 = model.model.forward_cl(xa); torch.cuda.synchronize()
        print("quiet vs quiet identical:", all(torch.equal(a[2], b[2]) for a, b in zip(quiet, REC)), torch.equal(ya, ya2))
        for trial in range(3):
            REC = None
            s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s2):
                for _ in range(3):
                    model.model.forward_cl(xb)
            with torch.cuda.stream(s1):
                REC = []
                yc = model.model.forward_cl(xa)
                conc = REC
                REC = None
            torch.cuda.synchronize()
            bad = [(i, a[0], a[1], float((a[2].float() - b[2].float()).abs().max()), int((a[2] != b[2]).sum()))
                   for i, (a, b) in enumerate(zip(quiet, conc)) if not torch.equal(a[2], b[2])]
            print(f"trial {trial}: output identical {torch.equal(ya, yc)}; differing ops {len(bad)} of {len(quiet)}")
            for b in bad[:6]:
                print("   ", b)


if __name__ == "__main__":
    main()
