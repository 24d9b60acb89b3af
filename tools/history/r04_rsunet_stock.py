"""RSUNet as the reference ships it (config/profiles/arch_profiles.yaml:34-44: width [18, 36, 48, 64, 80], GroupNorm(4), ELU, down
(1,2,2) x 4, depth_2d 1) against the hand-picked bench widths: training step + inference forward timings with the per-op table.
    python tools/history/r04_rsunet_stock.py [stock] [padded] [bench]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402
from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet  # noqa: E402
from pytorch_connectomics_amd.training.fused import FusedAdamW, bce_dice_loss  # noqa: E402

dev = torch.device("cuda:0")
CONFIGS = {
    "stock": dict(width=[18, 36, 48, 64, 80], norm="group", activation="elu", num_groups=4, down_factors=[(1, 2, 2)] * 4, depth_2d=1,
                  kernel_2d=(1, 3, 3)),
    "padded": dict(width=[24, 40, 48, 64, 80], norm="group", activation="elu", num_groups=4, down_factors=[(1, 2, 2)] * 4, depth_2d=1,
                   kernel_2d=(1, 3, 3)),
    "bench": dict(width=[16, 32, 64, 128], norm="batch", activation="relu"),
}


def run(name, patch=(18, 256, 256), batch=2, steps=5):
    torch.manual_seed(0)
    m = RSUNet(1, 1, **CONFIGS[name]).to(dev).train()
    m.compute_dtype = torch.bfloat16
    opt = FusedAdamW(m.parameters(), lr=1e-4, weight_decay=1e-2, max_grad_norm=1.0)
    x = torch.rand(batch, 1, *patch, device=dev)
    y = (torch.rand(batch, 1, *patch, device=dev) > 0.85).float()

    def tstep():
        opt.zero_grad(set_to_none=True)
        loss, _ = bce_dice_loss(m(x), y)
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        tstep()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        loss = tstep()
    e1.record()
    torch.cuda.synchronize()
    t_train = e0.elapsed_time(e1) / steps
    with ops.profiled() as prof:
        tstep()
    summ = prof.summary()
    m.eval()
    with torch.no_grad():
        for _ in range(2):
            m(x)
        e0.record()
        for _ in range(steps):
            m(x)
        e1.record()
        torch.cuda.synchronize()
    t_inf = e0.elapsed_time(e1) / steps
    flops = sum(r.get("flops", 0) for r in summ.values())
    print(f"== {name}: width {CONFIGS[name]['width']} patch {batch}x{patch}: train {t_train:.2f} ms/step (loss {float(loss):.4f}), "
          f"inference {t_inf:.2f} ms/forward, profiled-step kernel sum {sum(r['ms'] for r in summ.values()):.2f} ms, "
          f"conv GFLOP/step {flops / 1e9:.1f}", flush=True)
    for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:14]:
        print(f"   {k:44s} n={v['launches']:3d} ms={v['ms']:7.3f}", flush=True)


if __name__ == "__main__":
    for name in sys.argv[1:] or list(CONFIGS):
        run(name)
