"""ADVICE r04 (fold_norm with large channel offsets): mean |error| of the norm-folded mixer and of the affine-prologue form against fp64 math
for activations with |mean| / std = 0 ... 256.  Run in two trees for a before / after record (profiles/r05_fold_large_offsets.txt)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
import torch.nn.functional as F
from pytorch_connectomics_amd import hip_ops as ops
dev = torch.device("cuda:0"); bf = torch.bfloat16
for offset in (0.0, 8.0, 64.0, 256.0):
    torch.manual_seed(int(offset) + 3)
    N, rows, C, chid, cout, slots = 2, 2000, 32, 64, 32, 11
    t = (torch.randn(N, rows, C) + offset * (torch.rand(1, 1, C) + 0.5) * torch.sign(torch.randn(1, 1, C))).to(bf)
    tf = t.double(); s1, s2 = tf.sum(1), (tf * tf).sum(1)
    wts = torch.full((N, slots, 1), 1.0 / slots, dtype=torch.float64)
    stats = torch.stack([wts * s1[:, None], wts * s2[:, None]], 2).float().contiguous().to(dev)
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C) * 0.3
    w2, b2 = torch.randn(chid, C) / C ** 0.5, torch.randn(chid) * 0.5
    w3, b3 = torch.randn(cout, chid) / chid ** 0.5, torch.randn(cout) * 0.5
    w2n, b2n, ab = ops.groupnorm_fold_mlp(stats, float(rows), gamma.to(dev), beta.to(dev), 1e-5, w2.to(dev), b2.to(dev), want_ab=True)
    mean = s1 / rows; var = s2 / rows - mean * mean
    a = gamma.double()[None] / torch.sqrt(var + 1e-5); b = beta.double()[None] - mean * a
    ref = (F.gelu((tf * a[:, None] + b[:, None]) @ w2.double().t() + b2.double()) @ w3.double().t() + b3.double()).float()
    w3p = ops.pw_pack_weight_paired(w3.to(dev), f16=True)
    args = dict(N=N, rows_per_sample=rows, c_in=C, c_hid=chid, c_out=cout)
    y = ops.pw_mlp(t.to(dev), None, w2n, b2n, w3p, b3.to(dev), **args).float().cpu()
    y_aff = ops.pw_mlp(t.to(dev), ab, ops.pw_pack_weight_paired(w2.to(dev)), b2.to(dev), w3p, b3.to(dev), **args).float().cpu()
    print(f"offset {offset:6.1f}: mean |error| folded {float((y - ref).abs().mean()):.5f}  affine prologue {float((y_aff - ref).abs().mean()):.5f}", flush=True)
