"""Round-6 probe: the training forward + backward (MedNeXt-S, 112^3, bf16) (a) eager on one stream with the whole batch, (b) as a
hipGraph of the same, (c) as a hipGraph with the batch split into two half-batch lanes on two streams (fork / join inside the capture)."""
import sys
import time
from pathlib import Path

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
    cfg = ConfigNode(schema_defaults())
    cfg.model.arch.type, cfg.model.in_channels, cfg.model.out_channels = "mednext", 1, 1
    cfg.model.mednext.size, cfg.model.mednext.kernel_size = "S", 3
    cfg.optimization.optimizer.name, cfg.optimization.optimizer.lr = "AdamW", 1e-3
    cfg.optimization.gradient_clip_val = 1.0
    torch.manual_seed(0)
    model = bm(cfg).to(dev).train()
    model.model.compute_dtype = torch.bfloat16
    opt = build_optimizer(cfg, model)
    nb = 4
    b = next(synthetic_batches(nb, bench.ROI, seed=11, device=dev))
    x, y = b["image"], b["label"]

    def fb(xs, ys, scale=1.0):
        out = model(xs)
        loss, _ = bce_dice_loss(out, ys)
        (loss * scale).backward() if scale != 1.0 else loss.backward()
        return loss

    def timeit(fn, n=8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        return th / n * 1e3, (time.perf_counter() - t0) / n * 1e3

    def eager():
        opt.zero_grad(set_to_none=True)
        fb(x, y)

    for _ in range(3):
        eager()
    print("eager fwd+bwd, batch 4, one stream: host %.2f ms, wall %.2f ms" % timeit(eager), flush=True)
    g_ref = [p.grad.detach().clone() for p in model.parameters() if p.grad is not None]

    # (b) one-lane graph
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            fb(x, y)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    opt.zero_grad(set_to_none=True)
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        fb(x, y)
    g1.replay()
    torch.cuda.synchronize()
    d = max(float((p.grad - r).abs().max()) for p, r in zip([p for p in model.parameters() if p.grad is not None], g_ref))
    print("one-lane graph: max |grad - eager grad| = %.3g" % d, flush=True)
    print("graph fwd+bwd, batch 4, one lane: host %.2f ms, wall %.2f ms" % timeit(g1.replay), flush=True)
    del g1

    # (c) two half-batch lanes
    xa, xb, ya, yb = x[:2].contiguous(), x[2:].contiguous(), y[:2].contiguous(), y[2:].contiguous()
    la, lb = torch.cuda.Stream(), torch.cuda.Stream()

    def two_lanes():
        cur = torch.cuda.current_stream()
        la.wait_stream(cur)
        lb.wait_stream(cur)
        with torch.cuda.stream(la):
            oa = model(xa)
            l_a, _ = bce_dice_loss(oa, ya)
        with torch.cuda.stream(lb):
            ob = model(xb)
            l_b, _ = bce_dice_loss(ob, yb)
        cur.wait_stream(la)
        cur.wait_stream(lb)
        loss = (l_a + l_b) * 0.5
        loss.backward()
        return loss

    def eager2():
        opt.zero_grad(set_to_none=True)
        two_lanes()

    for _ in range(3):
        eager2()
    print("eager fwd+bwd, 2 lanes x 2 samples: host %.2f ms, wall %.2f ms" % timeit(eager2), flush=True)
    d = max(float((p.grad - r).abs().max() / (r.abs().max() + 1e-12)) for p, r in zip([p for p in model.parameters() if p.grad is not None], g_ref))
    print("two lanes eager: max relative grad deviation from the one-pass step %.3g" % d, flush=True)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            two_lanes()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    opt.zero_grad(set_to_none=True)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        two_lanes()
    g2.replay()
    torch.cuda.synchronize()
    d = max(float((p.grad - r).abs().max() / (r.abs().max() + 1e-12)) for p, r in zip([p for p in model.parameters() if p.grad is not None], g_ref))
    print("two-lane graph: max relative grad deviation %.3g" % d, flush=True)
    print("graph fwd+bwd, 2 lanes x 2 samples: host %.2f ms, wall %.2f ms" % timeit(g2.replay), flush=True)


if __name__ == "__main__":
    main()
