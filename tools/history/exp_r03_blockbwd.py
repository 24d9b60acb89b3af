"""Round-3 probe: backward of ONE MedNeXt block (32 -> 64 -> 32, k3, residual) in bf16 at BASELINE sizes against torch autograd through
the oracle block in fp32, under the training path's switches."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import mednext_oracle as MO  # noqa: E402
from pytorch_connectomics_amd.models.architectures.mednext import MedNeXtBlock  # noqa: E402
from pytorch_connectomics_amd.training import autograd as AG  # noqa: E402


def run(N, D, flags):
    torch.manual_seed(1)
    blk = MedNeXtBlock(32, 32, exp_r=2, kernel_size=3, do_res=True)
    st = {"b." + k: v.detach().clone() for k, v in blk.state_dict().items()}
    g = torch.Generator().manual_seed(2)
    x = torch.randn(N, 32, D, D, D, generator=g) * 0.5
    gy = torch.randn(N, 32, D, D, D, generator=g) * 0.01
    params = {k: v.clone().requires_grad_(True) for k, v in st.items()}
    xr = x.clone().requires_grad_(True)
    y = MO.block_forward(xr, params, "b", 3)
    y.backward(gy)
    for k, v in flags.items():
        setattr(AG, k, v)
    blk = blk.cuda()
    xc = x.permute(0, 2, 3, 4, 1).contiguous().cuda().bfloat16().requires_grad_(True)
    yc = AG._block(blk, xc)
    yc.backward(gy.permute(0, 2, 3, 4, 1).contiguous().cuda().bfloat16())
    out = {}
    def rel(a, b):
        return float((a.float().cpu() - b).norm() / (b.norm() + 1e-30))
    out["y"] = rel(yc.detach().permute(0, 4, 1, 2, 3), y.detach())
    out["dx"] = rel(xc.grad.permute(0, 4, 1, 2, 3), xr.grad)
    for n, p in blk.named_parameters():
        if n != "conv1.bias":
            out[n] = rel(p.grad, params["b." + n].grad)
    return out


for N, D in ((1, 32), (1, 64), (1, 112), (2, 112)):
    for flags in ({}, {"FUSED_RESIDUAL_DGRAD": False}, {"FUSED_RESIDUAL_DGRAD": True, "FUSED_TRAIN_MIXER": False}):
        r = run(N, D, flags)
        AG.FUSED_RESIDUAL_DGRAD, AG.FUSED_TRAIN_MIXER = True, True
        print(f"N={N} D={D} {flags}: " + " ".join(f"{k}={v:.4f}" for k, v in r.items()), flush=True)
