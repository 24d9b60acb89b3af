import torch
from pytorch_connectomics_amd import hip_ops as ops
from pytorch_connectomics_amd.models.architectures.mednext import create_mednext_v1
m = create_mednext_v1(1, 3, "B", 5).cuda().eval(); m.compute_dtype = torch.bfloat16
x = torch.randn(1, 160, 160, 160, 1, device="cuda")
with torch.no_grad():
    m.forward_cl(x)
    with ops.profiled() as prof:
        for _ in range(2): m.forward_cl(x)
s = prof.summary()
for k, v in sorted(s.items(), key=lambda kv: -kv[1]["ms"])[:10]:
    print("%-34s launches/fwd %5.1f  ms/fwd %7.3f" % (k, v["launches"] / 2, v["ms"] / 2))
