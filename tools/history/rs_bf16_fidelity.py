import torch, torch.nn.functional as F
from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet
from oracle import rsunet_oracle as RO
torch.manual_seed(5)
kw = dict(width=[8, 16, 24], norm="group", num_groups=8, activation="relu")
m = RSUNet(1, 1, **kw).cuda().train()
x = torch.randn(2, 1, 8, 32, 32, device="cuda")
y = (torch.rand(2, 1, 8, 32, 32, device="cuda") > 0.5).float()
F.binary_cross_entropy_with_logits(m(x), y).backward()
ref = {n: p.grad.clone() for n, p in m.named_parameters()}
m.zero_grad(); m.compute_dtype = torch.bfloat16
F.binary_cross_entropy_with_logits(m(x), y).backward()
# torch autocast on the oracle
params = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
with torch.autocast("cuda", dtype=torch.bfloat16):
    out = RO.forward(params, x.clone(), **kw)
F.binary_cross_entropy_with_logits(out.float(), y).backward()
for n, p in m.named_parameters():
    g, r, t = p.grad.flatten(), ref[n].flatten(), params[n].grad.flatten()
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm() + 1e-20))
    print(f"{n:44s} {p.numel():6d} hip-bf16 {cos(g, r):.4f}  torch-autocast {cos(t, r):.4f}")
