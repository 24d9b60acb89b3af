"""Differential fuzz over batch sizes and volume shapes the fixtures do not hold (round 6): the bf16 schedule (every fast kernel) against the
fp32 schedule of the SAME weights (generic kernels) -- inference logits and training gradients of MedNeXt (S widths / an L-like stage),
with samples that differ by an order of magnitude in scale and offset, so that a kernel that takes the wrong sample's statistics, a
tile that reads past a ragged edge or a slot rule that breaks at an odd batch shows as an outlier and not as bf16 noise.
Prints one line per case; exits non-zero when a case leaves the band the regular cases define."""
import sys
from pathlib import Path
import torch
import torch.nn.functional as F
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pytorch_connectomics_amd.models.architectures.mednext import MedNeXt  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
bad = 0


def make(exp_r, counts, k=3, ds=False):
    m = MedNeXt(1, 32, 2, exp_r=exp_r, kernel_size=k, deep_supervision=ds, do_res=True, do_res_up_down=True, block_counts=counts).to(dev)
    return m


def sample_batch(N, shape, g):
    x = torch.rand(N, 1, *shape, device=dev, generator=g)
    sc = torch.tensor([1.0, 8.0, 0.2, 20.0, 3.0, 0.05, 40.0, 1.5][:N], device=dev).view(N, 1, 1, 1, 1)
    of = torch.tensor([0.0, -3.0, 5.0, 1.0, -10.0, 0.5, 2.0, -1.0][:N], device=dev).view(N, 1, 1, 1, 1)
    return x * sc + of


def infer_case(m, N, shape, g):
    x = sample_batch(N, shape, g)
    with torch.no_grad():
        m.eval()
        m.compute_dtype = torch.float32
        ref = m(x).float()
        m.compute_dtype = torch.bfloat16
        got = m(x).float()
    d = (torch.sigmoid(got) - torch.sigmoid(ref)).abs()
    per_sample = d.flatten(1).max(1).values
    return float(d.max()), float(d.mean()), [round(float(v), 4) for v in per_sample]


def train_case(m, N, shape, g):
    x = sample_batch(N, shape, g)
    y = (torch.rand(N, 2, *shape, device=dev, generator=g) > 0.7).float()
    m.train()
    grads = {}
    for dt in (torch.float32, torch.bfloat16):
        m.compute_dtype = dt
        m.zero_grad()
        out = m(x)
        out = out[0] if isinstance(out, (list, tuple)) else out
        F.binary_cross_entropy_with_logits(out.float(), y).backward()
        grads[dt] = {k: p.grad.detach().double().flatten().clone() for k, p in m.named_parameters() if p.grad is not None}
    worst, wname = 1.0, ""
    for k, a in grads[torch.float32].items():
        b = grads[torch.bfloat16][k]
        na, nb = float(a.norm()), float(b.norm())
        if na < 1e-12:
            continue
        c = float((a * b).sum() / (na * nb + 1e-300))
        # bias of a conv that feeds a per-channel GroupNorm: zero in exact arithmetic, noise in either schedule
        if k.endswith("conv1.bias"):
            continue
        if c < worst:
            worst, wname = c, k
    ga = torch.cat(list(grads[torch.float32].values())); gb = torch.cat([grads[torch.bfloat16][k] for k in grads[torch.float32]])
    return float((ga * gb).sum() / (ga.norm() * gb.norm())), worst, wname


g = torch.Generator(device=dev).manual_seed(1)
mS = make(2, [2] * 9)
mL = make([3, 4, 8, 8, 8, 8, 8, 4, 3], [1, 1, 2, 2, 2, 2, 2, 1, 1])       # L-like widths, fewer blocks
shapes = [(1, (32, 32, 32)), (3, (32, 48, 64)), (5, (16, 80, 48)), (2, (48, 16, 112)), (3, (64, 64, 32)), (7, (16, 32, 48)), (1, (80, 48, 96)),
          (4, (48, 48, 48)), (3, (112, 32, 16))]
print("== inference: bf16 schedule vs fp32 schedule, max / mean |dP|, per-sample max")
for name, m in (("S", mS), ("L-like", mL)):
    for N, shp in shapes:
        mx, mean, per = infer_case(m, N, shp, g)
        flag = "  <-- OUTLIER" if (mx > 0.12 or mean > 6e-3) else ""
        bad += bool(flag)
        print(f"{name:7s} N={N} {shp}: max {mx:.4f} mean {mean:.5f} per-sample {per}{flag}")
print("== training: gradient cosine (all parameters), worst tensor")
for name, m in (("S", mS), ("L-like", mL)):
    for N, shp in shapes[:7]:
        call, worst, wname = train_case(m, N, shp, g)
        flag = "  <-- OUTLIER" if (call < 0.995 or worst < 0.90) else ""
        bad += bool(flag)
        print(f"{name:7s} N={N} {shp}: cos {call:.5f} worst {worst:.4f} ({wname}){flag}")
sys.exit(1 if bad else 0)
