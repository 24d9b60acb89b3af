"""CPU tests of the training harness (reference tests/unit/test_connectomics_module.py style: a stand-in model):
loss terms, deep supervision, optimizer grouping, scheduler, checkpoint layout, and DDP over gloo (world 2)."""
import math
import os
from pathlib import Path
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F

from pytorch_connectomics_amd.config import ConfigNode, schema_defaults

GOLD = Path(__file__).parent / "golden"
from pytorch_connectomics_amd.training.module import (ConnectomicsModule, WarmupCosineLR, build_optimizer,
                                                      dice_loss_sigmoid, fit, synthetic_batches,
                                                      weighted_bce_with_logits)


class SimpleModel(nn.Module):
    def __init__(self, ds=False):
        super().__init__()
        self.conv = nn.Conv3d(1, 1, 3, padding=1)
        self.norm = nn.GroupNorm(1, 1)
        self.ds = ds

    def forward(self, x):
        y = self.norm(self.conv(x))
        if not self.ds:
            return y
        return {"output": y, "ds_1": F.avg_pool3d(y, 2), "ds_2": F.avg_pool3d(y, 4)}


def _cfg(**opt):
    c = ConfigNode(schema_defaults())
    c.optimization.optimizer.lr = 1e-2
    for k, v in opt.items():
        c.optimization[k] = v
    return c


def test_loss_terms_and_deep_supervision():
    torch.manual_seed(0)
    logits, target = torch.randn(2, 1, 8, 8, 8), (torch.rand(2, 1, 8, 8, 8) > 0.7).float()
    p = torch.sigmoid(logits)
    inter, den = (p * target).sum((2, 3, 4)), p.sum((2, 3, 4)) + target.sum((2, 3, 4))
    assert torch.allclose(dice_loss_sigmoid(logits, target), (1 - (2 * inter + 1e-5) / (den + 1e-5)).mean())
    assert torch.allclose(weighted_bce_with_logits(logits, target), F.binary_cross_entropy_with_logits(logits, target))
    mask = (torch.rand_like(logits) > 0.5).float()
    ref = (F.binary_cross_entropy_with_logits(logits, target, reduction="none") * mask).sum() / mask.sum()
    assert torch.allclose(weighted_bce_with_logits(logits, target, weight=mask), ref)
    cfg = _cfg()
    cfg.model.loss.deep_supervision = True
    m = ConnectomicsModule(cfg, model=SimpleModel(ds=True))
    batch = {"image": torch.rand(2, 1, 8, 8, 8), "label": target}
    loss = m.training_step(batch)
    out = m(batch["image"])
    base = lambda o, t: weighted_bce_with_logits(o, t) + dice_loss_sigmoid(o, t)
    rs = lambda t, n: F.interpolate(t, size=(n, n, n), mode="trilinear", align_corners=False).clamp(-1, 1)   # float targets
    want = base(out["output"], target) + 0.5 * base(out["ds_1"], rs(target, 4)) + 0.25 * base(out["ds_2"], rs(target, 2))
    assert torch.allclose(loss, want, atol=1e-6) and loss.requires_grad
    val = m.validation_step(batch)
    assert 0 <= float(val["val_jaccard"]) <= 1
    cfg.model.loss.losses = [{"function": "Nope"}]
    with pytest.raises(ValueError, match="Unknown loss"):
        ConnectomicsModule(cfg, model=SimpleModel())
    cfg.model.loss.losses = [{"function": "MSELoss", "weight": 2.0, "pred_slice": "0:1", "target_slice": "0:1"}]
    cfg.model.loss.deep_supervision = False
    m2 = ConnectomicsModule(cfg, model=SimpleModel())
    assert torch.allclose(m2.training_step(batch), 2 * F.mse_loss(m2(batch["image"]), target))


def test_optimizer_grouping_and_scheduler():
    cfg = _cfg()
    cfg.optimization.optimizer.weight_decay = 0.01
    model = SimpleModel()
    opt = build_optimizer(cfg, model)
    wd = {id(p): g["weight_decay"] for g in opt.param_groups for p in g["params"]}
    assert len(opt.param_groups) == 2                                               # merged by (lr, weight decay)
    assert wd[id(model.conv.weight)] == 0.01 and wd[id(model.conv.bias)] == 0.01
    assert wd[id(model.norm.weight)] == 0.0 and wd[id(model.norm.bias)] == 0.0      # no decay on norm params
    sch = WarmupCosineLR(opt, max_iters=100, warmup_iters=10, warmup_factor=0.001)
    lrs = []
    for _ in range(100):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step(); sch.step()
    assert lrs[0] == pytest.approx(1e-2 * 0.001) and max(lrs) < 1e-2 and lrs[-1] < lrs[20]
    assert lrs[50] == pytest.approx(1e-2 * 0.5 * (1 + math.cos(math.pi * 50 / 100)))
    cfg.optimization.optimizer.name = "lion"
    with pytest.raises(ValueError, match="Unknown optimizer"):
        build_optimizer(cfg, model)


def test_fit_and_checkpoint_roundtrip(tmp_path):
    torch.manual_seed(0)
    cfg = _cfg(gradient_clip_val=1.0)
    m = ConnectomicsModule(cfg, model=SimpleModel())
    hist, opt = fit(m, synthetic_batches(2, (8, 8, 8)), max_steps=25, device=torch.device("cpu"), log=None)
    assert hist[-1] < hist[0] and m.global_step == 25
    ck = m.checkpoint_dict(opt)
    assert set(ck["state_dict"]) == {"model.conv.weight", "model.conv.bias", "model.norm.weight", "model.norm.bias"}
    torch.save(ck, tmp_path / "last.ckpt")
    m2 = ConnectomicsModule(cfg, model=SimpleModel())
    m2.load_checkpoint_dict(torch.load(tmp_path / "last.ckpt", weights_only=False))
    x = torch.rand(1, 1, 8, 8, 8)
    assert torch.equal(m(x), m2(x)) and m2.global_step == 25


def _ddp_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)                                  # same init on every rank
    m = ConnectomicsModule(_cfg(), model=SimpleModel())
    fit(m, synthetic_batches(2, (8, 8, 8), seed=42 + rank), max_steps=5, device=torch.device("cpu"), ddp=True, log=None)
    flat = torch.cat([p.detach().flatten() for p in m.model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    torch.distributed.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1])          # gradients were all-reduced: replicas stay identical
    # and differ from single-process training on rank-0 data only
    torch.manual_seed(0)
    solo = ConnectomicsModule(_cfg(), model=SimpleModel())
    fit(solo, synthetic_batches(2, (8, 8, 8), seed=42), max_steps=5, device=torch.device("cpu"), log=None)
    assert not torch.equal(torch.cat([p.detach().flatten() for p in solo.model.parameters()]), flat)
    torch.distributed.destroy_process_group()


def test_ddp_gloo_two_ranks():
    mp.spawn(_ddp_worker, args=(2, 29500 + (os.getpid() + 7) % 2000), nprocs=2, join=True)


def _ddp_worker8(rank, world, port):
    """The 8-rank job of BASELINE configs[2] in miniature (gloo here, RCCL on the node): every rank its own data stream, UNEVEN local
    batches (1 or 2 samples, what a DistributedSampler without drop_last leaves the last ranks with), one gradient all-reduce per step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    bs = lambda r: 1 + (r % 2)                            # noqa: E731
    cfg = _cfg()
    cfg.optimization.optimizer.name, cfg.optimization.optimizer.lr, cfg.optimization.optimizer.weight_decay = "SGD", 0.1, 0.0
    torch.manual_seed(0)
    m = ConnectomicsModule(cfg, model=SimpleModel())
    w0 = [p.detach().clone() for p in m.model.parameters()]
    fit(m, synthetic_batches(bs(rank), (8, 8, 8), seed=42 + rank), max_steps=1, device=torch.device("cpu"), ddp=True, log=None)
    flat = torch.cat([p.detach().flatten() for p in m.model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    torch.distributed.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], g) for g in gathered[1:])          # replicas identical on all 8 ranks
    if rank == 0:
        # the step every rank took is SGD on the MEAN over ranks of the local (per-rank mean) gradients -- DDP's contract, whatever
        # the local batch sizes: recomputed here from the eight data streams on one process
        grads = None
        for r in range(world):
            torch.manual_seed(0)
            solo = ConnectomicsModule(cfg, model=SimpleModel())
            batch = next(iter(synthetic_batches(bs(r), (8, 8, 8), seed=42 + r)))
            loss, _ = solo._compute_loss(solo(batch["image"]), batch["label"])
            loss.backward()
            g = [p.grad.clone() for p in solo.model.parameters()]
            grads = g if grads is None else [a + b for a, b in zip(grads, g)]
        want = torch.cat([(w - 0.1 * g / world).flatten() for w, g in zip(w0, grads)])
        assert torch.allclose(flat, want, rtol=1e-5, atol=1e-7), float((flat - want).abs().max())
    # three more steps keep the replicas in lock step
    fit(m, synthetic_batches(bs(rank), (8, 8, 8), seed=142 + rank), max_steps=4, device=torch.device("cpu"), ddp=True, log=None)
    flat = torch.cat([p.detach().flatten() for p in m.model.parameters()])
    torch.distributed.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], g) for g in gathered[1:])
    torch.distributed.destroy_process_group()


def test_ddp_gloo_eight_ranks_with_uneven_local_batches():
    mp.spawn(_ddp_worker8, args=(8, 31500 + (os.getpid() + 11) % 2000), nprocs=8, join=True)


def _ddp_balancing_worker(rank, world, port, strategy):
    """fit(ddp=True) with an adaptive loss weighter (tutorials/mitoEM/common.yaml trains with `uncertainty`) raised
    'Expected to mark a variable ready only once' at the first backward -- the weighter's parameters were DDP parameters used outside
    the wrapped forward.  Now DDP wraps the network and the weighter's gradients are averaged by hand."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg = _cfg(gradient_clip_val=1.0)
    cfg.model.loss.losses = [{"function": "WeightedBCEWithLogitsLoss", "weight": 1.0},
                             {"function": "DiceLoss", "weight": 0.5, "kwargs": {"sigmoid": True}}]
    cfg.model.loss.loss_balancing = {"strategy": strategy, "gradnorm_alpha": 0.0}
    torch.manual_seed(0)
    m = ConnectomicsModule(cfg, model=SimpleModel())
    if strategy == "gradnorm":
        assert m.loss_weighter.alpha == 0.0                    # an explicit 0.0 (equal gradient norms) is not the default 0.5
    w0 = torch.cat([p.detach().flatten().clone() for p in m.loss_weighter.parameters()])
    fit(m, synthetic_batches(2, (8, 8, 8), seed=42 + rank), max_steps=4, device=torch.device("cpu"), ddp=True, log=None)
    flat = torch.cat([p.detach().flatten() for p in list(m.model.parameters()) + list(m.loss_weighter.parameters())])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    torch.distributed.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], g) for g in gathered[1:])       # network AND weighter parameters identical on all ranks
    w1 = torch.cat([p.detach().flatten() for p in m.loss_weighter.parameters()])
    assert not torch.equal(w0, w1)                                       # ... and the weighter did train
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("strategy", ["uncertainty", "gradnorm"])
def test_ddp_with_adaptive_loss_balancing_keeps_weighter_replicas_identical(strategy):
    mp.spawn(_ddp_balancing_worker, args=(2, 33500 + (os.getpid() + 13) % 2000, strategy), nprocs=2, join=True)


def test_weighted_bce_matches_reference_fixture():
    """tests/golden/losses.npz: the reference's WeightedBCEWithLogitsLoss (losses.py:17-44,190-266) values and gradients."""
    import numpy as np
    from pathlib import Path
    z = np.load(Path(__file__).parent / "golden" / "losses.npz")
    names = sorted({k.split("__")[0] for k in z.files if not k.startswith("reg_")})
    assert len(names) == 6
    for n in names:
        x = torch.from_numpy(z[f"{n}__x"]).requires_grad_()
        t = torch.from_numpy(z[f"{n}__t"])
        w = torch.from_numpy(z[f"{n}__w"]) if f"{n}__w" in z.files else None
        pw = float(z[f"{n}__pw"][0])
        v = weighted_bce_with_logits(x, t, w, None if pw < 0 else pw)
        assert abs(float(v) - float(z[f"{n}__loss"][0])) < 1e-6, n
        v.backward()
        assert torch.allclose(x.grad, torch.from_numpy(z[f"{n}__grad"]), atol=1e-8, rtol=1e-5), n


def test_fused_epilogue_host_logic_and_no_cpu_path():
    """training/fused.py without a GPU: the stride collapsing that lets the loss kernel read the channels-last network
    output in place, and the loud failure on CPU tensors (there is no CPU path)."""
    from pytorch_connectomics_amd.training.fused import FusedAdamW, _ncr_strides, bce_dice_loss
    cl = torch.zeros(2, 5, 6, 7, 3).permute(0, 4, 1, 2, 3)            # NDHWC memory viewed as NCDHW
    assert _ncr_strides(cl) == (5 * 6 * 7 * 3, 1, 3)
    assert _ncr_strides(torch.zeros(2, 3, 5, 6, 7)) == (3 * 210, 210, 1)
    assert _ncr_strides(cl[:, 1:3]) == (630, 1, 3)                     # a channel slice stays collapsible
    assert _ncr_strides(torch.zeros(2, 3, 5, 6, 7)[:, :, :, ::2]) is None   # strided spatial dim: needs a copy
    assert _ncr_strides(torch.zeros(2, 1, 5, 6, 7).expand(2, 3, 5, 6, 7)) == (210, 0, 1)   # broadcast mask
    with pytest.raises(RuntimeError, match="no CPU path"):
        bce_dice_loss(torch.zeros(1, 1, 4, 4, 4), torch.zeros(1, 1, 4, 4, 4))
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        FusedAdamW([p]).step()
    # build_optimizer falls back to torch.optim.AdamW for CPU parameters (the fused kernel needs device pointers)
    cfg = _cfg()
    cfg.optimization.optimizer.name = "AdamW"
    opt = build_optimizer(cfg, SimpleModel())
    assert type(opt).__name__ == "AdamW" and not isinstance(opt, FusedAdamW)


def test_named_head_loss_routing():
    """Loss terms read the head they name (pred_head), else model.primary_head, else the only head
    (reference training/losses/orchestrator.py:328-378); a head name on a single-tensor output is an error."""
    class TwoHeads(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Conv3d(1, 1, 1)
            self.b = torch.nn.Conv3d(1, 2, 1)

        def forward(self, x):
            return {"output": {"sem": self.a(x), "aff": self.b(x)}}

    torch.manual_seed(0)
    HEADS = {"sem": {"out_channels": 1}, "aff": {"out_channels": 2}}
    cfg = _cfg()
    cfg.model.heads = HEADS
    cfg.model.loss.losses = [{"function": "WeightedBCEWithLogitsLoss", "weight": 1.0, "pred_head": "sem", "target_slice": "0:1"},
                             {"function": "DiceLoss", "weight": 0.5, "pred_head": "aff", "target_slice": "1:3", "kwargs": {"sigmoid": True}},
                             {"function": "MSELoss", "weight": 0.25, "pred_head": "aff", "pred_slice": "0:1", "target_slice": "0:1"}]
    net = TwoHeads()
    m = ConnectomicsModule(cfg, model=net)
    x = torch.rand(2, 1, 4, 4, 4)
    y = (torch.rand(2, 3, 4, 4, 4) > 0.5).float()
    loss = m.training_step({"image": x, "label": y})
    out = net(x)["output"]
    want = (weighted_bce_with_logits(out["sem"], y[:, 0:1]) + 0.5 * dice_loss_sigmoid(out["aff"], y[:, 1:3])
            + 0.25 * F.mse_loss(out["aff"][:, 0:1], y[:, 0:1]))
    assert torch.allclose(loss, want, atol=1e-6) and loss.requires_grad
    assert {"loss_0_WeightedBCEWithLogitsLoss", "loss_1_DiceLoss", "loss_2_MSELoss", "train_loss_total"} <= set(m.last_log)
    # unnamed terms fall back to model.primary_head
    cfg2 = _cfg()
    cfg2.model.heads, cfg2.model.primary_head = HEADS, "sem"
    cfg2.model.loss.losses = [{"function": "DiceLoss", "weight": 1.0, "target_slice": "0:1", "kwargs": {"sigmoid": True}}]
    m2 = ConnectomicsModule(cfg2, model=net)
    assert torch.allclose(m2.training_step({"image": x, "label": y}), dice_loss_sigmoid(out["sem"], y[:, 0:1]), atol=1e-6)
    cfg2.model.primary_head = None                      # changed behind the module's back: caught when the loss runs ...
    with pytest.raises(ValueError, match="multiple heads"):
        m2.training_step({"image": x, "label": y})
    with pytest.raises(ValueError, match=r"losses\[0\] must define pred_head or model.primary_head when model.heads has multiple entries"):
        ConnectomicsModule(cfg2, model=net)             # ... and refused up front when the module is built (plan.py:213-218)
    # the head's own target_slice is the default target of its terms (plan.py:189-193)
    cfg4 = _cfg()
    cfg4.model.heads = {"sem": {"out_channels": 1, "target_slice": "2:3"}, "aff": {"out_channels": 2, "target_slice": "0:2"}}
    cfg4.model.loss.losses = [{"function": "DiceLoss", "weight": 1.0, "pred_head": "sem", "kwargs": {"sigmoid": True}}]
    assert torch.allclose(ConnectomicsModule(cfg4, model=net).training_step({"image": x, "label": y}),
                          dice_loss_sigmoid(out["sem"], y[:, 2:3]), atol=1e-6)
    cfg3 = _cfg()
    cfg3.model.heads = HEADS
    cfg3.model.loss.losses = [{"function": "DiceLoss", "weight": 1.0, "pred_head": "nope"}]
    with pytest.raises(ValueError, match=r"pred_head='nope' is not one of the configured model.heads \['aff', 'sem'\]"):
        ConnectomicsModule(cfg3, model=net)
    cfg3.model.heads = None
    with pytest.raises(ValueError, match="uses pred_head/pred2_head but model.heads is not configured"):
        ConnectomicsModule(cfg3, model=net)
    cfg5 = _cfg()                                       # heads configured, single-tensor network: caught when the loss runs
    cfg5.model.heads, cfg5.model.loss.losses = HEADS, [{"function": "DiceLoss", "weight": 1.0, "pred_head": "sem"}]
    with pytest.raises(ValueError, match="single tensor"):
        ConnectomicsModule(cfg5, model=SimpleModel()).training_step({"image": torch.rand(2, 1, 8, 8, 8), "label": torch.rand(2, 1, 8, 8, 8)})


def test_warmup_cosine_lr_matches_reference_fixture():
    """tests/golden/lr_schedule.json: per-iteration learning rates of the reference's WarmupCosineLR
    (training/optimization/lr_scheduler.py:51-106) for two parameter groups."""
    import json
    from pathlib import Path
    from pytorch_connectomics_amd.training.module import WarmupCosineLR
    for case in json.loads((Path(__file__).parent / "golden" / "lr_schedule.json").read_text()):
        ps = [torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))]
        opt = torch.optim.SGD([{"params": [ps[0]], "lr": 0.1}, {"params": [ps[1]], "lr": 0.02}], lr=0.1)
        sch = WarmupCosineLR(opt, max_iters=case["max_iters"], warmup_factor=case["warmup_factor"],
                             warmup_iters=case["warmup_iters"], eta_min=case["eta_min"])
        for want in case["lrs"]:
            got = [g["lr"] for g in opt.param_groups]
            assert all(abs(a - b) <= 1e-12 + 1e-9 * abs(b) for a, b in zip(got, want)), (case["max_iters"], got, want)
            opt.step()
            sch.step()


def test_optimizer_groups_match_reference_fixture():
    """tests/golden/optimizer_groups.json: per-parameter (lr, weight_decay), class and defaults of the reference's
    build_optimizer (training/optimization/build.py:47-160) incl. shared / frozen parameters and the norm / bias rules."""
    import json
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).parent / "golden"))
    rows = json.loads((Path(__file__).parent / "golden" / "optimizer_groups.json").read_text())

    def model():
        torch.manual_seed(0)
        m = torch.nn.Sequential()
        m.add_module("conv", torch.nn.Conv3d(1, 4, 3))
        m.add_module("gn", torch.nn.GroupNorm(2, 4))
        m.add_module("act", torch.nn.PReLU())
        m.add_module("conv2", torch.nn.Conv3d(4, 4, 1, bias=False))
        m.add_module("bn", torch.nn.BatchNorm3d(4))
        m.add_module("head", torch.nn.Conv3d(4, 2, 1))
        m.add_module("tied", torch.nn.Conv3d(4, 2, 1))
        m.tied.weight = m.head.weight
        m.conv2.weight.requires_grad_(False)
        return m

    for r in rows:
        cfg = _cfg()
        for k, v in r["cfg"].items():
            setattr(cfg.optimization.optimizer, k, v)
        m = model()
        opt = build_optimizer(cfg, m)
        assert type(opt).__name__ == r["class"], r["cfg"]
        names = {id(p): n for n, p in m.named_parameters()}
        per = {}
        for g in opt.param_groups:
            for p in g["params"]:
                assert names[id(p)] not in per
                per[names[id(p)]] = [g["lr"], g["weight_decay"]]
        assert per == {k: v for k, v in r["per_param"].items()}, (r["cfg"], per, r["per_param"])
        g0 = opt.param_groups[0]
        if r["betas"]:
            assert list(g0["betas"]) == r["betas"] and g0["eps"] == r["eps"]
        if r["momentum"] is not None:
            assert g0["momentum"] == r["momentum"]


def test_regression_losses_match_reference_fixture():
    """WeightedMSELoss / WeightedMAELoss / SmoothL1Loss (losses.py:140-187, 725-800) incl. tanh and weight maps."""
    import numpy as np
    from pathlib import Path
    from pytorch_connectomics_amd.training.module import _LOSSES
    z = np.load(Path(__file__).parent / "golden" / "losses.npz")
    cases = {"mse_plain": ("WeightedMSELoss", {}), "mse_tanh_mask": ("WeightedMSELoss", {"tanh": True}), "mae_real": ("WeightedMAELoss", {}),
             "huber_beta": ("SmoothL1Loss", {"beta": 0.3, "tanh": True}), "huber_none_valid": ("SmoothL1Loss", {})}
    for n, (fn, kw) in cases.items():
        x = torch.from_numpy(z[f"reg_{n}__x"]).requires_grad_()
        t = torch.from_numpy(z[f"reg_{n}__t"])
        w = torch.from_numpy(z[f"reg_{n}__w"]) if f"reg_{n}__w" in z.files else None
        v = _LOSSES[fn](x, t, weight=w, pos_weight=None, **kw)
        assert abs(float(v.detach()) - float(z[f"reg_{n}__loss"][0])) < 1e-6, n
        v.backward()
        assert torch.allclose(x.grad, torch.from_numpy(z[f"reg_{n}__grad"]), atol=1e-8, rtol=1e-5), n


def test_deep_supervision_loss_matches_reference_orchestrator_fixture():
    """tests/golden/ds_loss.npz (make_golden.py --ds_loss): LossOrchestrator.compute_deep_supervision_loss /
    compute_standard_loss on 5 scales with logits beyond the +-20 clamp, a real-valued target channel (trilinear resize +
    range clamp) and a batch mask (nearest resize): totals, per-scale values and every gradient."""
    from pytorch_connectomics_amd.training.module import match_target_to_output
    g = np.load(GOLD / "ds_loss.npz")
    cfg = _cfg()
    cfg.model.loss.deep_supervision = True
    cfg.model.loss.losses = [{"function": "WeightedBCEWithLogitsLoss", "weight": 1.0}]
    m = ConnectomicsModule(cfg, model=SimpleModel())
    names = ["output", "ds_1", "ds_2", "ds_3", "ds_4"]
    lab, mask = torch.from_numpy(g["label"]), torch.from_numpy(g["mask"])
    for case, mk in (("nomask", None), ("mask", mask)):
        outs = {k: torch.from_numpy(g[f"in_{k}"]).clone().requires_grad_(True) for k in names}
        tot, _ = m._compute_loss(outs, lab, mk)
        tot.backward()
        assert float(tot) == pytest.approx(float(g[f"{case}__total"]), rel=2e-6)
        for k in names:
            assert torch.allclose(outs[k].grad, torch.from_numpy(g[f"{case}__grad_{k}"]), rtol=1e-5, atol=1e-9), (case, k)
        # per-scale values (unweighted), the reference logs them as train_loss_scale_i
        for i, k in enumerate(names):
            tgt = match_target_to_output(lab, outs[k])
            mm = None if mk is None else F.interpolate(mk, size=outs[k].shape[2:], mode="nearest")
            v, _ = m._term_loss(outs[k].detach(), tgt, mm)
            assert float(v) == pytest.approx(float(g[f"{case}__scales"][i]), rel=2e-6), (case, i)
    cfg.model.loss.deep_supervision = False
    m1 = ConnectomicsModule(cfg, model=SimpleModel())
    o = torch.from_numpy(g["in_output"]).clone().requires_grad_(True)
    tot, _ = m1._compute_loss(o, lab, mask)
    tot.backward()
    assert float(tot) == pytest.approx(float(g["standard__total"]), rel=2e-6)                  # the clamp applies here too
    assert torch.allclose(o.grad, torch.from_numpy(g["standard__grad_output"]), rtol=1e-5, atol=1e-9)
    assert (o.grad[torch.from_numpy(g["in_output"]).abs() > 20] == 0).all()                    # clamped logits get no gradient
    sdt = match_target_to_output(torch.from_numpy(g["sdt"]), torch.zeros(1, 1, 4, 5, 5))
    assert torch.allclose(sdt, torch.from_numpy(g["sdt_resized"]), atol=1e-7)
    il = match_target_to_output(torch.from_numpy(g["ilab"]), torch.zeros(1, 1, 4, 5, 5))
    assert il.dtype == torch.int64 and torch.equal(il, torch.from_numpy(g["ilab_resized"]))


def test_scheduler_table_defaults_and_interval():
    """optimization/build.py:155-324 + schema/optimization.py defaults: CosineAnnealingLR(T_max=max_epochs, eta_min=min_lr)
    stepping once per EPOCH; `name: null` = none; 'step' interval; unknown interval is an error."""
    from torch.optim import lr_scheduler as LS
    from pytorch_connectomics_amd.training.module import build_lr_scheduler, resolve_training_steps
    d = schema_defaults()["optimization"]
    assert d["gradient_clip_val"] == 1.0 and d["precision"] == "16-mixed" and d["n_steps_per_epoch"] == -1
    assert d["scheduler"]["name"] == "CosineAnnealingLR" and d["scheduler"]["min_lr"] == 1e-5 and d["scheduler"]["interval"] == "epoch"
    cfg = _cfg(max_epochs=4)
    model = SimpleModel()
    opt = build_optimizer(cfg, model)
    sch = build_lr_scheduler(cfg, opt)
    assert isinstance(sch, LS.CosineAnnealingLR) and sch.T_max == 4 and sch.eta_min == 1e-5
    for name, kind in (("StepLR", LS.StepLR), ("MultiStepLR", LS.MultiStepLR), ("ReduceLROnPlateau", LS.ReduceLROnPlateau),
                       ("CosineAnnealingWarmRestarts", LS.CosineAnnealingWarmRestarts), ("WarmupCosineLR", WarmupCosineLR),
                       ("constant", LS.LambdaLR), ("somethingelse", LS.CosineAnnealingLR)):
        cfg.optimization.scheduler.name = name
        assert isinstance(build_lr_scheduler(cfg, build_optimizer(cfg, model)), kind), name
    cfg.optimization.scheduler.name = None
    assert build_lr_scheduler(cfg, opt) is None
    # epoch interval: 12 steps, 3 per epoch -> 4 scheduler steps; lr constant within an epoch
    cfg.optimization.scheduler.name = "CosineAnnealingLR"
    cfg.optimization.n_steps_per_epoch = 3
    total, per = resolve_training_steps(cfg)
    assert (total, per) == (12, 3)
    lrs = []
    m = ConnectomicsModule(cfg, model=SimpleModel())
    fit(m, synthetic_batches(1, (8, 8, 8)), max_steps=total, steps_per_epoch=per, device=torch.device("cpu"),
        log=lambda msg: lrs.append(float(msg.rsplit("lr ", 1)[1])), log_every=1)
    assert m.current_epoch == 4 and m._scheduler.last_epoch == 4
    # the line is logged after the step's scheduler update: the third step of an epoch already shows the next epoch's lr
    assert lrs[0] == lrs[1] == pytest.approx(1e-2) and lrs[2] < lrs[1] and lrs[2] == lrs[3] == lrs[4] and lrs[-1] < lrs[6]
    cfg.optimization.scheduler.interval = "step"
    m = ConnectomicsModule(cfg, model=SimpleModel())
    fit(m, synthetic_batches(1, (8, 8, 8)), max_steps=6, steps_per_epoch=3, device=torch.device("cpu"), log=None)
    assert m._scheduler.last_epoch == 6
    cfg.optimization.scheduler.interval = "hourly"
    with pytest.raises(ValueError, match="interval"):
        fit(ConnectomicsModule(cfg, model=SimpleModel()), synthetic_batches(1, (8, 8, 8)), max_steps=2, device=torch.device("cpu"), log=None)
    # n_steps_per_epoch = -1 (schema default) with an endless sampler: explicit error, fast_dev_run / max_steps still work
    cfg.optimization.n_steps_per_epoch = -1
    with pytest.raises(ValueError, match="n_steps_per_epoch"):
        resolve_training_steps(cfg)
    assert resolve_training_steps(cfg, fast_dev_run=3) == (3, 3)
    assert resolve_training_steps(cfg, dataset_steps_per_epoch=7) == (28, 7)


def test_resume_continues_optimizer_scheduler_and_counters(tmp_path):
    """Save at step 6 of 12, reload into a fresh module, continue: identical weights and LR trajectory to the uninterrupted
    run (Adam moments, scheduler position, epoch / step counters all restored; lightning/model.py + trainer resume)."""
    def make():
        torch.manual_seed(0)
        cfg = _cfg(max_epochs=4, n_steps_per_epoch=3)
        return ConnectomicsModule(cfg, model=SimpleModel())
    dev = torch.device("cpu")
    full = make()
    fit(full, synthetic_batches(2, (8, 8, 8), seed=5), max_steps=12, steps_per_epoch=3, device=dev, log=None)
    half = make()
    _, opt = fit(half, synthetic_batches(2, (8, 8, 8), seed=5), max_steps=6, steps_per_epoch=3, device=dev, log=None)
    ck = half.checkpoint_dict(opt)
    assert ck["epoch"] == 2 and ck["global_step"] == 6 and ck["lr_schedulers"][0]["last_epoch"] == 2
    torch.save(ck, tmp_path / "mid.ckpt")
    res = make()
    res.load_checkpoint_dict(torch.load(tmp_path / "mid.ckpt", weights_only=True))       # plain tensors / numbers only
    import itertools
    rest = itertools.islice(synthetic_batches(2, (8, 8, 8), seed=5), 6, None)                  # the batches the full run saw next
    hist, opt2 = fit(res, rest, max_steps=12, steps_per_epoch=3, device=dev, log=None)
    assert len(hist) == 6 and res.global_step == 12 and res.current_epoch == 4
    for a, b in zip(full.model.parameters(), res.model.parameters()):
        assert torch.allclose(a, b, rtol=0, atol=1e-7)
    assert opt2.param_groups[0]["lr"] == pytest.approx(full._scheduler.get_last_lr()[0])
    # a resumed run without the optimizer state is a different trajectory (the moments matter)
    cold = make()
    ck2 = {k: v for k, v in ck.items() if k not in ("optimizer_states", "lr_schedulers")}
    cold.load_checkpoint_dict(ck2)
    fit(cold, itertools.islice(synthetic_batches(2, (8, 8, 8), seed=5), 6, None), max_steps=12, steps_per_epoch=3, device=dev, log=None)
    assert not all(torch.allclose(a, b, atol=1e-7) for a, b in zip(full.model.parameters(), cold.model.parameters()))


def test_resume_accepts_reference_optimizer_layout_and_extra_heads():
    """The reference builds ONE param group per parameter (build.py:73-112); its optimizer_states map onto this engine's merged
    groups by model order.  Reference MedNeXt checkpoints also carry out_1..out_4 heads with deep_supervision off."""
    torch.manual_seed(0)
    cfg = _cfg()
    m = ConnectomicsModule(cfg, model=SimpleModel())
    params = [p for p in m.model.parameters() if p.requires_grad]
    ref_opt = torch.optim.AdamW([{"params": [p], "lr": 1e-2, "weight_decay": 0.0 if p.ndim == 1 else 0.01} for p in params])
    for p in params:
        p.grad = torch.randn_like(p)
    ref_opt.step()
    ck = {"state_dict": {"model." + k: v.clone() for k, v in m.model.state_dict().items()}, "global_step": 1, "epoch": 0,
          "optimizer_states": [ref_opt.state_dict()]}
    ck["state_dict"]["model.out_1.conv_out.weight"] = torch.zeros(1, 1, 1, 1, 1)              # ignored extra head
    m2 = ConnectomicsModule(cfg, model=SimpleModel())
    m2.load_checkpoint_dict(ck)
    opt, sched = m2.configure_optimizers()
    assert m2.restore_training_state(opt, sched)
    for p_ref, p_new in zip(params, [p for p in m2.model.parameters() if p.requires_grad]):
        assert torch.equal(ref_opt.state[p_ref]["exp_avg"], opt.state[p_new]["exp_avg"])
        assert torch.equal(ref_opt.state[p_ref]["exp_avg_sq"], opt.state[p_new]["exp_avg_sq"])
        assert float(opt.state[p_new]["step"]) == 1.0


def test_checkpoint_with_adaptive_loss_weights_warns_when_no_balancer_is_configured():
    """a reference run trained with `loss_balancing.strategy: uncertainty` stores `loss_weighter.*` tensors; a module built
    WITHOUT loss balancing cannot use them and says so; one built with it restores them (test_loss_balancing_matches_reference_fixture)."""
    cfg = _cfg()
    m = ConnectomicsModule(cfg, model=SimpleModel())
    ck = {"state_dict": {"model." + k: v.clone() for k, v in m.model.state_dict().items()}, "global_step": 3, "epoch": 1}
    ck["state_dict"]["loss_weighter.log_vars"] = torch.tensor([0.25, -0.5])
    m2 = ConnectomicsModule(cfg, model=SimpleModel())
    with pytest.warns(RuntimeWarning, match="adaptive loss-balancing state"):
        m2.load_checkpoint_dict(ck)
    assert m2.global_step == 3
    cfg.model.loss.loss_balancing = {"strategy": "uncertainty"}
    m3 = ConnectomicsModule(cfg, model=SimpleModel())
    m3.load_checkpoint_dict(ck)
    assert torch.equal(m3.loss_weighter.log_vars.detach(), torch.tensor([0.25, -0.5]))
    cfg.model.loss.loss_balancing = {"strategy": "gradnorm"}
    with pytest.raises(RuntimeError, match="does not match the configured strategy"):
        ConnectomicsModule(cfg, model=SimpleModel()).load_checkpoint_dict(ck)


def test_per_channel_bce_and_auto_pos_weight_match_reference_fixture():
    """tests/golden/losses_extra.npz (make_golden.py --losses_extra): the reference's PerChannelBCEWithLogitsLoss values and
    gradients (per-channel balancing, caps, a channel without positives, valid masks, sum reduction) and the orchestrator's
    scalar `pos_weight: auto`."""
    from pytorch_connectomics_amd.training.module import auto_pos_weight_scalar, per_channel_bce_with_logits
    z = np.load(GOLD / "losses_extra.npz")
    for n in sorted({k.split("__")[0] for k in z.files if k.startswith("pc_")}):
        x = torch.from_numpy(z[f"{n}__x"]).requires_grad_()
        t = torch.from_numpy(z[f"{n}__t"])
        w = torch.from_numpy(z[f"{n}__w"]) if f"{n}__w" in z.files else None
        cap, auto, is_sum = z[f"{n}__kw"]
        v = per_channel_bce_with_logits(x, t, w, auto_pos_weight=bool(auto), max_pos_weight=float(cap),
                                        reduction="sum" if is_sum else "mean")
        v.backward()
        assert float(v.detach()) == pytest.approx(float(z[f"{n}__loss"][0]), rel=2e-6), n
        assert torch.allclose(x.grad, torch.from_numpy(z[f"{n}__grad"]), rtol=1e-5, atol=1e-8), n
    for n in ("auto_sparse", "auto_mask", "auto_nopos"):
        t = torch.from_numpy(z[f"{n}__t"])
        m = torch.from_numpy(z[f"{n}__m"]) if f"{n}__m" in z.files else None
        assert float(auto_pos_weight_scalar(t, m)) == pytest.approx(float(z[f"{n}__pw"][0]), rel=1e-6), n
    # through the module: a config term with pos_weight "auto" and the per-channel loss
    cfg = _cfg()
    cfg.model.loss.losses = [{"function": "WeightedBCEWithLogitsLoss", "weight": 1.0, "pos_weight": "auto", "kwargs": {"reduction": "mean"}},
                             {"function": "PerChannelBCEWithLogitsLoss", "weight": 0.5, "kwargs": {"auto_pos_weight": True, "max_pos_weight": 50.0}}]
    m = ConnectomicsModule(cfg, model=SimpleModel())
    x, t = torch.from_numpy(z["pc_mask__x"]), torch.from_numpy(z["pc_mask__t"])
    w = torch.from_numpy(z["pc_mask__w"])
    tot, parts = m._compute_loss(x, t, w)
    pw = auto_pos_weight_scalar(t, w)
    want = weighted_bce_with_logits(x.clamp(-20, 20), t, w, pw) + 0.5 * float(z["pc_mask__loss"][0])
    assert float(tot) == pytest.approx(float(want), rel=1e-5)


def test_monai_style_losses_follow_their_published_formulas():
    """Dice / Tversky / Focal as the reference configures them (MONAI is not installed: parity unpinned; the formulas are
    checked against direct restatements, the kwargs the tutorial configs use are accepted, masks enter through the inputs)."""
    from pytorch_connectomics_amd.training.module import monai_dice_loss, monai_focal_loss, monai_tversky_loss
    torch.manual_seed(3)
    x, t = torch.randn(2, 3, 4, 5, 6) * 2, (torch.rand(2, 3, 4, 5, 6) > 0.6).float()
    p = torch.sigmoid(x)
    d = 1 - (2 * (p * t).sum((2, 3, 4)) + 1e-5) / (p.sum((2, 3, 4)) + t.sum((2, 3, 4)) + 1e-5)
    assert torch.allclose(monai_dice_loss(x, t, sigmoid=True, smooth_nr="1e-5", smooth_dr="1e-5"), d.mean())      # YAML strings
    assert torch.allclose(monai_dice_loss(x, t, sigmoid=True, include_background=False), d[:, 1:].mean())
    assert torch.allclose(monai_dice_loss(x, t), (1 - (2 * (x * t).sum((2, 3, 4)) + 1e-5) / (x.sum((2, 3, 4)) + t.sum((2, 3, 4)) + 1e-5)).mean())
    tp, fp, fn = (p * t).sum((2, 3, 4)), 0.3 * (p * (1 - t)).sum((2, 3, 4)), 0.7 * ((1 - p) * t).sum((2, 3, 4))
    assert torch.allclose(monai_tversky_loss(x, t, sigmoid=True, alpha=0.3, beta=0.7), (1 - (tp + 1e-5) / (tp + fp + fn + 1e-5)).mean())
    pt = p * t + (1 - p) * (1 - t)
    focal = (-(0.25 * t + 0.75 * (1 - t)) * (1 - pt) ** 2.0 * torch.log(pt)).mean()
    assert torch.allclose(monai_focal_loss(x, t, gamma=2.0, alpha=0.25), focal, rtol=1e-5)
    assert torch.isfinite(monai_focal_loss(torch.full((1, 1, 2, 2, 2), 80.0), torch.zeros(1, 1, 2, 2, 2)))        # stable tails
    cfg = _cfg()
    cfg.model.loss.losses = [{"function": "DiceLoss", "weight": 1.0, "kwargs": {"sigmoid": True, "smooth_nr": "1e-5", "smooth_dr": "1e-5"}},
                             {"function": "TverskyLoss", "weight": 0.5, "kwargs": {"sigmoid": True, "alpha": 0.3, "beta": 0.7}},
                             {"function": "FocalLoss", "weight": 2.0, "kwargs": {"gamma": 2.0, "alpha": 0.25}}]
    m = ConnectomicsModule(cfg, model=SimpleModel())
    mask = (torch.rand(2, 1, 4, 5, 6) > 0.3).float()
    tot, parts = m._compute_loss(x, t, mask)
    xm, tm = x.clamp(-20, 20).masked_fill(~(mask > 0).expand_as(x), -20.0), t * mask
    want = monai_dice_loss(xm, tm, sigmoid=True) + 0.5 * monai_tversky_loss(xm, tm, sigmoid=True, alpha=0.3, beta=0.7) \
        + 2.0 * monai_focal_loss(xm, tm, gamma=2.0, alpha=0.25)
    assert torch.allclose(tot, want, rtol=1e-6) and len(parts) == 4


def test_static_loss_weights_are_the_default():
    """No `loss_balancing` (or strategy none): no weighter, the fused loss path stays on (the adaptive strategies:
    test_loss_balancing_matches_reference_fixture)."""
    cfg = _cfg()
    m = ConnectomicsModule(cfg, model=SimpleModel())
    assert m.loss_weighter is None and m.fused_loss
    cfg.model.loss.loss_balancing = {"strategy": None}
    assert ConnectomicsModule(cfg, model=SimpleModel()).loss_weighter is None


def test_loss_orchestration_matches_reference_orchestrator_fixture():
    """tests/golden/loss_orchestration.npz (make_golden.py --loss_orchestration): the reference's LossOrchestrator on the term lists
    of tests/loss_cases.py.  What it pins beyond the per-loss fixtures: a `weight`-taking regression loss (WeightedMSE / MAE /
    SmoothL1) gets a CLASS-BALANCING weight map by default (foreground : background = neg : pos, mean 1, capped at 10; a numeric
    pos_weight = that weight on the foreground), a term's own `mask_slice` replaces it, the weighted BCE keeps a scalar pos_weight,
    torch's own BCE / MSE see masks through their inputs, `apply_deep_supervision: false` keeps a term off the coarser scales,
    `coefficient` / `pred` / `target` / `mask` are accepted spellings."""
    from loss_cases import CASES, loss_cfg, loss_tensors
    g = np.load(GOLD / "loss_orchestration.npz")
    for index, (label, terms, ds, use_mask) in enumerate(CASES):
        m = ConnectomicsModule(loss_cfg(terms, ds), model=SimpleModel())
        outs, lab, mask = loss_tensors(index)
        outs = {k: v.requires_grad_(True) for k, v in outs.items()}
        total, _ = m._compute_loss(outs if ds else outs["output"], lab, mask if use_mask else None)
        total.backward()
        assert float(total) == pytest.approx(float(g[f"{label}__total"]), rel=3e-6), label
        for k, v in outs.items():
            key = f"{label}__grad_{k}"
            if key in g.files:
                assert torch.allclose(v.grad, torch.from_numpy(g[key]), rtol=2e-5, atol=1e-8), (label, k)
            else:
                assert v.grad is None or float(v.grad.abs().sum()) == 0.0, (label, k)
    with pytest.raises(ValueError, match="pos_weight is only supported for losses with spatial_weight_arg='weight'"):
        ConnectomicsModule(loss_cfg([{"function": "DiceLoss", "weight": 1.0, "pos_weight": 2.0}], False), model=SimpleModel())
    with pytest.raises(ValueError, match="pos_weight must be a positive number or 'auto'"):
        ConnectomicsModule(loss_cfg([{"function": "WeightedMSELoss", "weight": 1.0, "pos_weight": "balanced"}], False), model=SimpleModel())


def test_load_ema_state_dict_finds_the_callback_entry():
    """`load_ema_state_dict(checkpoint)`: the reference's contract (training/lightning/callbacks.py:46-60) -- Lightning keys the
    callback state by class name, possibly followed by its arguments; a checkpoint without EMA gives None, never the raw weights."""
    from pytorch_connectomics_amd.training import load_ema_state_dict
    shadow = {"weight": torch.full((2, 2), 0.271)}
    ck = {"callbacks": {"ModelCheckpoint{'monitor': 'val_loss_total'}": {"best_model_score": 0.5},
                        "EMAWeightsCallback{'decay': 0.9}": {"ema_state": shadow, "updates": 3, "decay": 0.9}}}
    got = load_ema_state_dict(ck)
    assert got is not shadow and torch.equal(got["weight"], shadow["weight"])
    assert load_ema_state_dict({"callbacks": {"EMAWeightsCallback": {"ema_state": {}, "updates": 0}}}) is None
    assert load_ema_state_dict({"callbacks": {"ModelCheckpoint": {}}}) is None
    assert load_ema_state_dict({"callbacks": None}) is None and load_ema_state_dict({}) is None


def test_loss_balancing_matches_reference_fixture():
    """tests/golden/balancing.npz (make_golden.py --balancing): the reference's UncertaintyLossWeighter and GradNormLossWeighter on a
    toy model with three task losses -- totals, task weights, gradients of the weighter's parameters and of the model, the second
    step's ratios against the first step's losses, eval mode."""
    from pytorch_connectomics_amd.training.balancing import (GradNormLossWeighter, UncertaintyLossWeighter, build_loss_weighter,
                                                             select_shared_parameters)
    z = np.load(GOLD / "balancing.npz")
    net = nn.Sequential(nn.Linear(6, 5), nn.Tanh(), nn.Linear(5, 3))

    def reset():
        with torch.no_grad():
            for p, k in zip(net.parameters(), ("w0", "b0", "w2", "b2")):
                p.copy_(torch.from_numpy(z[k]))
        net.zero_grad()
    x, tgt = torch.from_numpy(z["x"]), torch.from_numpy(z["tgt"])

    def tasks():
        y = net(x)
        return [((y[:, 0] - tgt[:, 0]) ** 2).mean() * 1.0, (y[:, 1] - tgt[:, 1]).abs().mean() * 0.5, F.softplus(y[:, 2] * tgt[:, 2]).mean() * 2.0]
    names = ["a", "b", "c"]
    reset()
    uw = UncertaintyLossWeighter(3)
    with torch.no_grad():
        uw.log_vars.copy_(torch.tensor([0.3, -0.2, 0.0]))
    tot, wts, _ = uw.combine(tasks(), names, "train")
    tot.backward()
    assert float(tot) == pytest.approx(float(z["unc_total"]), rel=1e-6) and np.allclose(wts.numpy(), z["unc_weights"], rtol=1e-6)
    assert np.allclose(uw.log_vars.grad.numpy(), z["unc_grad_logvars"], rtol=1e-5) and np.allclose(net[2].weight.grad.numpy(), z["unc_grad_w2"], rtol=1e-5, atol=1e-8)
    for strat in ("last", "first", "all"):
        reset()
        gw = GradNormLossWeighter(3, alpha=0.5, gradnorm_lambda=1.0, shared_parameters=select_shared_parameters(net, strat))
        with torch.no_grad():
            gw.task_weights.copy_(torch.tensor([1.5, 0.7, 1.0]))
        gw.train()
        tot, wts, _ = gw.combine(tasks(), names, "train")
        tot.backward()
        assert float(tot) == pytest.approx(float(z[f"gn_{strat}_total"]), rel=1e-6), strat
        assert np.allclose(wts.numpy(), z[f"gn_{strat}_weights"], rtol=1e-6)
        assert np.allclose(gw.task_weights.grad.numpy(), z[f"gn_{strat}_grad_tw"], rtol=2e-5, atol=1e-7), strat
        assert np.allclose(net[2].weight.grad.numpy(), z[f"gn_{strat}_grad_w2"], rtol=1e-5, atol=1e-8)
        if strat == "last":
            with torch.no_grad():
                for p in net.parameters():
                    p -= 0.05 * p.grad
                gw.task_weights -= 0.1 * gw.task_weights.grad
            net.zero_grad(); gw.task_weights.grad = None
            tot2, wts2, _ = gw.combine(tasks(), names, "train")
            tot2.backward()
            assert float(tot2) == pytest.approx(float(z["gn_step2_total"]), rel=1e-6) and np.allclose(wts2.numpy(), z["gn_step2_weights"], rtol=1e-6)
            assert np.allclose(gw.task_weights.grad.numpy(), z["gn_step2_grad_tw"], rtol=2e-5, atol=1e-7)
            gw.eval()
            assert float(gw.combine(tasks(), names, "val")[0]) == pytest.approx(float(z["gn_eval_total"]), rel=1e-6)
    # through the module: the weighter is a trainable sub-module, the optimizer gets its parameters, the fused loss path steps aside
    cfg = _cfg()
    cfg.model.loss.losses = [{"function": "WeightedBCEWithLogitsLoss", "weight": 1.0}, {"function": "DiceLoss", "weight": 0.5, "kwargs": {"sigmoid": True}}]
    cfg.model.loss.loss_balancing = {"strategy": "gradnorm", "gradnorm_alpha": 0.5}
    m = ConnectomicsModule(cfg, model=SimpleModel())
    assert isinstance(m.loss_weighter, GradNormLossWeighter) and m.loss_weighter.shared_parameters[0] is list(m.model.parameters())[-1]
    opt, _ = m.configure_optimizers()
    assert any(p is m.loss_weighter.task_weights for g in opt.param_groups for p in g["params"])
    batch = {"image": torch.rand(2, 1, 8, 8, 8), "label": (torch.rand(2, 1, 8, 8, 8) > 0.7).float()}
    loss = m.training_step(batch)
    loss.backward()
    assert m.loss_weighter.task_weights.grad is not None and "loss_1_DiceLoss_balance_weight" in m.last_log
    # checkpoint round trip: task weights and the first step's losses travel under the reference's `loss_weighter.` prefix
    ck = m.checkpoint_dict()
    assert {"loss_weighter.task_weights", "loss_weighter.initial_losses"} <= set(ck["state_dict"])
    m2 = ConnectomicsModule(cfg, model=SimpleModel())
    with torch.no_grad():
        m.loss_weighter.task_weights.add_(0.25)
    m2.load_checkpoint_dict(m.checkpoint_dict())
    assert torch.equal(m2.loss_weighter.task_weights, m.loss_weighter.task_weights) and torch.equal(m2.loss_weighter.initial_losses, m.loss_weighter.initial_losses)
    # uncertainty (tutorials/mitoEM/common.yaml:54-55) through the module: total = sum 0.5 exp(-s_i) w_i L_i + 0.5 sum s_i
    cfg.model.loss.loss_balancing = {"strategy": "uncertainty"}
    mu = ConnectomicsModule(cfg, model=SimpleModel())
    assert isinstance(mu.loss_weighter, UncertaintyLossWeighter)
    with torch.no_grad():
        mu.loss_weighter.log_vars.copy_(torch.tensor([0.4, -0.3]))
    loss = mu.training_step(batch)
    raw = [float(mu.last_log["loss_0_WeightedBCEWithLogitsLoss"]), float(mu.last_log["loss_1_DiceLoss"])]
    want = 0.5 * np.exp(-0.4) * 1.0 * raw[0] + 0.5 * np.exp(0.3) * 0.5 * raw[1] + 0.5 * (0.4 - 0.3)
    assert float(loss) == pytest.approx(want, rel=1e-5)
    loss.backward()
    g = mu.loss_weighter.log_vars.grad
    assert g[0].item() == pytest.approx(-0.5 * np.exp(-0.4) * raw[0] + 0.5, rel=1e-5) and g[1].item() == pytest.approx(-0.5 * np.exp(0.3) * 0.5 * raw[1] + 0.5, rel=1e-5)
    # a term that skips the deep-supervision scales cannot be balanced there (the weighter's vector has one entry per term)
    cfg.model.loss.losses[1]["apply_deep_supervision"] = False
    md = ConnectomicsModule(cfg, model=SimpleModel())
    with pytest.raises(ValueError, match="apply_deep_supervision"):
        md._balanced_scale_loss([(torch.zeros(1, 1, 4, 4, 4), torch.zeros(1, 1, 4, 4, 4), None, [(0, md.loss_terms[0])])], "train")
    del cfg.model.loss.losses[1]["apply_deep_supervision"]
    cfg.model.loss.loss_balancing = {"strategy": "pcgrad"}
    with pytest.raises(ValueError, match="Unknown loss balancing strategy"):
        ConnectomicsModule(cfg, model=SimpleModel())
    assert build_loss_weighter(_cfg(), 2, None) is None
