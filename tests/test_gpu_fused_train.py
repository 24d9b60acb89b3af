"""GPU parity of the train-step epilogue kernels: fused BCE + Dice loss against the torch
formulas of training/module.py (which restate losses.py:190-266 + MONAI DiceLoss), multi-tensor clip + AdamW (+ EMA)
against torch.optim.AdamW + clip_grad_norm_."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_loss(x, t, w, w_bce, w_dice, pw):
    """BCE takes the mask as its weight argument; Dice has none and sees it through its inputs (orchestrator.py:648-655):
    invalid voxels carry the clamp floor as logit (sigmoid(-20) = 2e-9: the kernel uses exactly 0) and 0 as target."""
    from pytorch_connectomics_amd.training.module import _mask_for_unweighted_loss, dice_loss_sigmoid, weighted_bce_with_logits
    xd, td = _mask_for_unweighted_loss(x, t, w, -20.0)
    return w_bce * weighted_bce_with_logits(x, t, w, pw) + w_dice * dice_loss_sigmoid(xd, td)


@pytest.mark.parametrize("C,layout,mask,pw,wb,wd", [(1, "ncdhw", False, None, 1.0, 1.0), (3, "cl", False, None, 1.0, 0.5),
                                                    (3, "cl", True, 2.5, 0.7, 1.0), (2, "ncdhw", True, None, 1.0, 0.0),
                                                    (4, "sliced", False, 0.5, 0.0, 1.0)])
def test_bce_dice_loss_matches_torch(C, layout, mask, pw, wb, wd):
    from pytorch_connectomics_amd.training.fused import bce_dice_loss
    torch.manual_seed(C)
    N, D, H, W = 2, 9, 20, 33
    base = torch.randn(N, D, H, W, C + (2 if layout == "sliced" else 0), device="cuda") * 3
    if layout == "ncdhw":
        x = base.permute(0, 4, 1, 2, 3).contiguous()
    elif layout == "cl":
        x = base.permute(0, 4, 1, 2, 3)                 # channels-last memory viewed as NCDHW (the network's output)
    else:
        x = base.permute(0, 4, 1, 2, 3)[:, 1:1 + C]     # a channel slice of it
    t = (torch.rand(N, C, D, H, W, device="cuda") > 0.7).float()
    w = (torch.rand(N, 1, D, H, W, device="cuda") > 0.2).float() if mask else None
    xr = x.detach().clone().requires_grad_()
    ref = _ref_loss(xr, t, w, wb, wd, pw)
    ref.backward()
    xg = x.detach().clone(memory_format=torch.preserve_format).requires_grad_()
    if layout != "ncdhw":                                # keep the strided view
        holder = base.detach().clone().requires_grad_()
        xg = holder.permute(0, 4, 1, 2, 3) if layout == "cl" else holder.permute(0, 4, 1, 2, 3)[:, 1:1 + C]
    loss, parts = bce_dice_loss(xg, t, w, w_bce=wb, w_dice=wd, pos_weight=pw)
    torch.testing.assert_close(loss.detach(), ref.detach(), rtol=2e-5, atol=1e-6)
    (loss * 1.5).backward()
    got = (holder.grad.permute(0, 4, 1, 2, 3) if layout == "cl" else
           holder.grad.permute(0, 4, 1, 2, 3)[:, 1:1 + C] if layout == "sliced" else xg.grad)
    torch.testing.assert_close(got, 1.5 * xr.grad, rtol=2e-4, atol=1e-9)
    assert abs(float(parts[0]) - float(loss.detach())) == 0.0
    loss2, _ = bce_dice_loss(xg.detach(), t, w, w_bce=wb, w_dice=wd, pos_weight=pw)
    assert torch.equal(loss2, loss.detach())            # deterministic


def test_fused_adamw_matches_torch_adamw_with_clip_and_ema():
    from pytorch_connectomics_amd.training.fused import FusedAdamW
    torch.manual_seed(0)
    shapes = [(1,), (7,), (64, 64), (5000,), (3, 70000), (16, 1, 3, 3, 3)]
    pa = [torch.randn(s, device="cuda").requires_grad_() for s in shapes]
    pb = [p.detach().clone().requires_grad_() for p in pa]
    ga = [{"params": pa[:3], "lr": 1e-2, "weight_decay": 0.0}, {"params": pa[3:], "lr": 2e-2, "weight_decay": 0.1}]
    gb = [{"params": pb[:3], "lr": 1e-2, "weight_decay": 0.0}, {"params": pb[3:], "lr": 2e-2, "weight_decay": 0.1}]
    fa = FusedAdamW(ga, betas=(0.9, 0.99), eps=1e-8, max_grad_norm=0.5, ema_decay=0.9, ema_warmup_steps=1)
    tb = torch.optim.AdamW(gb, betas=(0.9, 0.99), eps=1e-8)
    ema = [p.detach().clone() for p in pb]
    for it in range(6):
        grads = [torch.randn_like(p) * (0.01 if it % 2 else 1.0) for p in pa]
        for p, q, g in zip(pa, pb, grads):
            p.grad, q.grad = g.clone(), g.clone()
        nrm = torch.nn.utils.clip_grad_norm_(pb, 0.5)
        tb.step()
        fa.step()
        d = 0.0 if it + 1 <= 1 else 0.9
        for e, q in zip(ema, pb):
            e.mul_(d).add_(q.detach(), alpha=1 - d)
        torch.testing.assert_close(fa.last_grad_norm, nrm, rtol=1e-5, atol=0)
        for p, q in zip(pa, pb):
            torch.testing.assert_close(p, q, rtol=2e-5, atol=2e-6)
    for p, e in zip(pa, ema):
        torch.testing.assert_close(fa.ema[p], e, rtol=2e-5, atol=2e-6)
    assert all(p._version > 0 for p in pa)          # raw-pointer updates still bump the version counters (weight caches)
    # optimizer state interchanges with torch.optim.AdamW
    sd = fa.state_dict()
    assert float(sd["state"][0]["step"]) == 6.0
    tb2 = torch.optim.AdamW([{"params": [p.detach().clone().requires_grad_() for p in pa[:3]], "lr": 1e-2, "weight_decay": 0.0},
                             {"params": [p.detach().clone().requires_grad_() for p in pa[3:]], "lr": 2e-2, "weight_decay": 0.1}],
                            betas=(0.9, 0.99), eps=1e-8)
    import copy
    tb2.load_state_dict(copy.deepcopy(sd))       # (load_state_dict keeps same-device tensors by reference)
    fa.max_grad_norm = 0.0
    grads = [torch.randn_like(p) for p in pa]
    for p, q, g in zip(pa, [q for g in tb2.param_groups for q in g["params"]], grads):
        p.grad, q.grad = g.clone(), g.clone()
    tb2.step()
    fa.step()
    for p, q in zip(pa, [q for g in tb2.param_groups for q in g["params"]]):
        torch.testing.assert_close(p, q, rtol=2e-5, atol=2e-6)
    o = FusedAdamW([torch.zeros(3, requires_grad=True)])          # CPU parameters: no CPU path
    o.param_groups[0]["params"][0].grad = torch.ones(3)
    with pytest.raises(RuntimeError):
        o.step()


def test_module_fused_loss_equals_generic_path_with_deep_supervision():
    from types import SimpleNamespace as NS
    from pytorch_connectomics_amd.config import ConfigNode, schema_defaults
    from pytorch_connectomics_amd.training.module import ConnectomicsModule
    cfg = ConfigNode(schema_defaults())
    cfg.model.arch.type, cfg.model.in_channels, cfg.model.out_channels = "mednext_custom", 1, 2
    cfg.model.mednext.base_channels, cfg.model.mednext.exp_r, cfg.model.mednext.kernel_size = 8, 2, 3
    cfg.model.mednext.block_counts = [1] * 9
    cfg.model.deep_supervision = True
    cfg.model.loss.deep_supervision = True
    cfg.model.loss.losses = [{"function": "WeightedBCEWithLogitsLoss", "weight": 1.0, "pos_weight": 2.0},
                             {"function": "DiceLoss", "weight": 0.5, "kwargs": {"sigmoid": True}}]
    torch.manual_seed(3)
    mod = ConnectomicsModule(cfg).cuda().train()
    x = torch.rand(2, 1, 32, 32, 32, device="cuda")
    y = (torch.rand(2, 2, 32, 32, 32, device="cuda") > 0.8).float()
    m = (torch.rand(2, 1, 32, 32, 32, device="cuda") > 0.1).float()
    res = {}
    for fused in (True, False):
        mod.fused_loss = fused
        mod.zero_grad()
        loss = mod.training_step({"image": x, "label": y, "mask": m})
        loss.backward()
        res[fused] = (loss.detach().clone(), {n: p.grad.clone() for n, p in mod.model.named_parameters() if p.grad is not None},
                      dict(mod.last_log))
    torch.testing.assert_close(res[True][0], res[False][0], rtol=1e-5, atol=1e-6)
    assert set(res[True][2]) == set(res[False][2])
    for n, g in res[False][1].items():
        scale = float(g.abs().max().clamp_min(1e-8))
        assert float((res[True][1][n] - g).abs().max()) <= 2e-3 * scale + 1e-7, n


def test_fused_bce_matches_reference_fixture():
    """The fused kernel's BCE term against the reference's WeightedBCEWithLogitsLoss outputs (tests/golden/losses.npz)."""
    import numpy as np
    from pathlib import Path
    from pytorch_connectomics_amd.training.fused import bce_dice_loss
    z = np.load(Path(__file__).parent / "golden" / "losses.npz")
    for n in sorted({k.split("__")[0] for k in z.files if not k.startswith("reg_")}):
        x = torch.from_numpy(z[f"{n}__x"]).cuda().requires_grad_()
        t = torch.from_numpy(z[f"{n}__t"]).cuda()
        w = torch.from_numpy(z[f"{n}__w"]).cuda() if f"{n}__w" in z.files else None
        pw = float(z[f"{n}__pw"][0])
        loss, parts = bce_dice_loss(x, t, w, w_bce=1.0, w_dice=0.0, pos_weight=None if pw < 0 else pw)
        assert abs(float(loss.detach()) - float(z[f"{n}__loss"][0])) < 2e-6 * max(1.0, float(z[f"{n}__loss"][0])), n
        loss.backward()
        torch.testing.assert_close(x.grad.cpu(), torch.from_numpy(z[f"{n}__grad"]), rtol=2e-4, atol=1e-9)
