"""Round-3 experiment: weight gradients on a side stream (training/autograd.py SIDE_STREAM_WGRAD): bit-identity of every gradient,
then ms per MedNeXt-S training step (4 x 112^3, bf16, fused loss + FusedAdamW) with the lane off / on."""
import sys
import time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from pytorch_connectomics_amd.models.architectures.mednext import create_mednext_v1  # noqa: E402
from pytorch_connectomics_amd.training import autograd as AG  # noqa: E402
from pytorch_connectomics_amd.training.fused import FusedAdamW, bce_dice_loss  # noqa: E402
from pytorch_connectomics_amd.utils.hostgc import quiesce_gc  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
m = create_mednext_v1(1, 1, "S", kernel_size=3).to(dev).train()
m.compute_dtype = torch.bfloat16
x = torch.rand(2, 1, 64, 64, 64, device=dev)
y = (torch.rand(2, 1, 64, 64, 64, device=dev) > 0.85).float()
grads = {}
for on in (False, True, True):
    AG.SIDE_STREAM_WGRAD = on
    m.zero_grad(set_to_none=True)
    loss, _ = bce_dice_loss(m(x), y)
    loss.backward()
    torch.cuda.synchronize()
    g = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    if not grads:
        grads = g
    else:
        bad = [n for n in g if not torch.equal(g[n], grads[n])]
        print(f"lane {on}: gradients differing from the single-stream backward: {len(bad)} of {len(g)} {bad[:4]}")

x = torch.rand(4, 1, 112, 112, 112, device=dev)
y = (torch.rand(4, 1, 112, 112, 112, device=dev) > 0.85).float()
opt = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0)


def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = bce_dice_loss(m(x), y)
    loss.backward()
    opt.step()


for on in (False, True, False, True):
    AG.SIDE_STREAM_WGRAD = on
    for _ in range(3):
        step()
    quiesce_gc()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    print(f"lane {on}: {1e3 * (time.perf_counter() - t0) / 8:.2f} ms per step", flush=True)

# host issue time vs GPU time of a step
for on in (False, True):
    AG.SIDE_STREAM_WGRAD = on
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"lane {on}: host issue {1e3 * t_issue / 4:.2f} ms per step, wall {1e3 * t_all / 4:.2f} ms per step")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(2):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
