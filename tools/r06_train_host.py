"""Round-6 probe: host enqueue time of one MedNeXt-S training step (4 x 112^3, bf16) against its GPU time."""
import sys
import time
from pathlib import Path
from types import SimpleNamespace as NS

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from pytorch_connectomics_amd.config import ConfigNode, schema_defaults
    from pytorch_connectomics_amd.models import build_model as bm
    from pytorch_connectomics_amd.training.fused import bce_dice_loss
    from pytorch_connectomics_amd.training.module import build_optimizer, synthetic_batches
    from pytorch_connectomics_amd.utils.hostgc import quiesce_gc
    cfg = ConfigNode(schema_defaults())
    cfg.model.arch.type, cfg.model.in_channels, cfg.model.out_channels = "mednext", 1, 1
    cfg.model.mednext.size, cfg.model.mednext.kernel_size = "S", 3
    cfg.optimization.optimizer.name, cfg.optimization.optimizer.lr = "AdamW", 1e-3
    cfg.optimization.gradient_clip_val = 1.0
    torch.manual_seed(0)
    model = bm(cfg).to(dev).train()
    model.model.compute_dtype = torch.bfloat16
    opt = build_optimizer(cfg, model)
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    it = synthetic_batches(nb, bench.ROI, seed=11, device=dev)
    pool = [next(it) for _ in range(2)]

    def tstep(i, marks=None):
        b = pool[i % 2]
        opt.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        out = model(b["image"])
        t1 = time.perf_counter()
        loss, _ = bce_dice_loss(out, b["label"])
        t2 = time.perf_counter()
        loss.backward()
        t3 = time.perf_counter()
        opt.step()
        t4 = time.perf_counter()
        if marks is not None:
            marks.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
        return loss

    for i in range(3):
        tstep(i)
    quiesce_gc()
    torch.cuda.synchronize()
    n = 6
    marks = []
    t0 = time.perf_counter()
    for i in range(n):
        tstep(i, marks)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    ta = time.perf_counter() - t0
    print(f"batch {nb}: host enqueue {th / n * 1e3:.2f} ms per step, wall {ta / n * 1e3:.2f} ms per step")
    # host-only cost: a step enqueued while the GPU is far behind would block in the queue; time single steps from an idle GPU instead
    for i in range(3):
        torch.cuda.synchronize()
        m = []
        t0 = time.perf_counter()
        tstep(i, m)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        ta = time.perf_counter() - t0
        print(f"  from idle: host {th * 1e3:.2f} ms (fwd {m[0][0] * 1e3:.2f} loss {m[0][1] * 1e3:.2f} bwd {m[0][2] * 1e3:.2f} opt {m[0][3] * 1e3:.2f}), wall {ta * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
