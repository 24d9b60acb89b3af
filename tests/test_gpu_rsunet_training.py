"""GPU parity of the RSUNet training path: every backward kernel against torch autograd of the same op, and a full
forward + backward of three RSUNet configurations against autograd through the CPU oracle (which equals the reference's
rsunet.py forward exactly, tests/test_oracle_golden.py)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cl(x):
    return x.permute(0, 2, 3, 4, 1).contiguous()


def _cf(y):
    return y.permute(0, 4, 1, 2, 3).contiguous()


@pytest.mark.parametrize("ci,co,ks,shape", [(5, 7, (3, 3, 3), (4, 6, 7)), (8, 4, (1, 3, 3), (3, 9, 8)), (3, 70, (1, 1, 1), (4, 5, 6))])
def test_conv3d_backward_kernels(ci, co, ks, shape):
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(ci + co)
    x = torch.randn(2, ci, *shape, requires_grad=True)
    w = (torch.randn(co, ci, *ks) * 0.3).requires_grad_()
    y = F.conv3d(x, w, padding=tuple(k // 2 for k in ks))
    gy = torch.randn_like(y)
    y.backward(gy)
    dW = ops.conv3d_wgrad(_cl(x.detach()).cuda(), _cl(gy).cuda(), ks)
    torch.testing.assert_close(dW.cpu(), w.grad, rtol=1e-4, atol=1e-3)
    wt = w.detach().flip(2, 3, 4).transpose(0, 1).contiguous().cuda()
    dx = ops.conv3d(_cl(gy).cuda(), ops.conv3d_pack_weight(wt, torch.float32), c_out=ci, kernel=ks)
    torch.testing.assert_close(_cf(dx.cpu()), x.grad, rtol=1e-4, atol=1e-4)


def test_maxpool_and_upsample_backward():
    from pytorch_connectomics_amd import hip_ops as ops
    from oracle.rsunet_oracle import bilinear_kernel
    torch.manual_seed(1)
    x = torch.randn(2, 5, 6, 8, 10, requires_grad=True)
    for fac in ((1, 2, 2), (2, 2, 2), (3, 2, 1)):
        x.grad = None
        y = F.max_pool3d(x, fac)
        gy = torch.randn_like(y)
        y.backward(gy)
        dx = ops.maxpool3d_bwd(_cl(x.detach()).cuda(), _cl(gy).cuda(), fac)
        torch.testing.assert_close(_cf(dx.cpu()), x.grad)
    for fac in ((1, 2, 2), (2, 2, 2)):
        x.grad = None
        wk = bilinear_kernel(5, fac)
        import math
        pad = [int(math.ceil((f - 1) / 2.0)) for f in fac]
        y = F.conv_transpose3d(x, wk, stride=fac, padding=pad, groups=5)
        gy = torch.randn_like(y)
        y.backward(gy)
        taps = wk.reshape(5, -1).t().contiguous().cuda()
        dx = ops.dwconv3d_generic(_cl(gy).cuda(), taps, wk.shape[2:], fac, pad, x.shape[2:])
        torch.testing.assert_close(_cf(dx.cpu()), x.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("kind,act", [("group", "relu"), ("instance", "elu"), ("batch", "prelu"), ("none", "leakyrelu")])
def test_norm_act_function_backward(kind, act):
    from pytorch_connectomics_amd.models.architectures.rsunet import NormAct
    from pytorch_connectomics_amd.training.rsunet_autograd import _norm_act
    torch.manual_seed(3)
    C = 12
    na = NormAct(C, kind, act, num_groups=4, negative_slope=0.1, init=0.2, alpha=1.0).train()
    if kind in ("group", "batch"):
        with torch.no_grad():
            na.norm.weight.uniform_(0.5, 1.5)
            na.norm.bias.normal_()
    x = torch.randn(2, C, 4, 5, 6, requires_grad=True)
    ref = na.act(na.norm(x)) if act == "prelu" else na.act(na.norm(x).clone())
    gy = torch.randn_like(ref)
    ref.backward(gy)
    want = {n: p.grad.clone() for n, p in na.named_parameters() if p.grad is not None}
    xg = x.grad.clone()
    rm = na.norm.running_mean.clone() if kind == "batch" else None
    na2 = NormAct(C, kind, act, num_groups=4, negative_slope=0.1, init=0.2, alpha=1.0).train()
    na2.load_state_dict({k: v for k, v in na.state_dict().items() if "running" not in k and "num_batches" not in k}, strict=False)
    na2 = na2.cuda()
    xc = _cl(x.detach()).cuda().requires_grad_()
    out = _norm_act(na2, xc)
    torch.testing.assert_close(_cf(out.detach().cpu()), ref.detach(), rtol=1e-4, atol=1e-5)
    out.backward(_cl(gy).cuda())
    torch.testing.assert_close(_cf(xc.grad.cpu()), xg, rtol=1e-3, atol=2e-5)
    for n, p in na2.named_parameters():
        if n in want:
            torch.testing.assert_close(p.grad.cpu(), want[n], rtol=1e-3, atol=1e-4)
    if kind == "batch":
        torch.testing.assert_close(na2.norm.running_mean.cpu(), rm, rtol=1e-4, atol=1e-6)


def _check_grads(named_grads, ref_of, kinked: bool):
    """fp32 gradients vs a reference.  Smooth activations (ELU): tight per-tensor bound.  Kinked ones (ReLU / leaky /
    PReLU): a pre-activation within fp32 rounding of 0 may land on the other side of the kink than in the reference --
    ONE such voxel moved d(beta) of a layer by 1.2e-2 and everything upstream by 5e-3 in the c1_group fixture (the
    channel sum it enters cancels to 1e-2 of that term) -- so the per-tensor bound is loose there and the direction of
    the whole gradient is checked instead."""
    flat_g, flat_r = [], []
    for n, g in named_grads:
        r = ref_of(n)
        assert g is not None and r is not None, n
        err = float((g - r).abs().max()) / float(r.abs().max().clamp_min(1e-6))
        assert err < (3e-2 if kinked else 2e-4), f"{n}: rel grad err {err:.2e}"
        flat_g.append(g.flatten().double())
        flat_r.append(r.flatten().double())
    g, r = torch.cat(flat_g), torch.cat(flat_r)
    cos = float((g * r).sum() / (g.norm() * r.norm()))
    rel2 = float((g - r).norm() / r.norm())
    assert cos > 0.99995 and rel2 < (1e-2 if kinked else 1e-4), (cos, rel2)


CFGS = {
    "c1_group": dict(width=[8, 16], down_factors=[(2, 2, 2)], norm="group", num_groups=8, activation="relu"),
    "aniso_inst_elu_ds": dict(width=[6, 8, 12], norm="instance", activation="elu", deep_supervision=True),
    "batch_prelu_2d": dict(width=[4, 8, 8], norm="batch", activation="prelu", depth_2d=1, init=0.1),
    "none_leaky": dict(width=[4, 8], norm="none", activation="leakyrelu", negative_slope=0.05),
    # the reference's stock widths (arch_profiles.yaml:34-44) are no multiples of 8: the HIP path carries them as 24 / 40 channels
    # (fp32: 20 / 36) with an all-zero tail -- GroupNorm(3, 18) / (4, 36) on the real channels, BatchNorm as tutorials/syn_cremi.yaml
    "stock_group_elu": dict(width=[18, 36], norm="group", num_groups=4, activation="elu", down_factors=[(1, 2, 2)], depth_2d=1,
                            kernel_2d=(1, 3, 3)),
    "stock_batch_elu": dict(width=[18, 36], norm="batch", num_groups=8, activation="elu", down_factors=[(1, 2, 2)], depth_2d=1,
                            kernel_2d=(1, 3, 3)),
}


@pytest.mark.parametrize("name", list(CFGS))
def test_rsunet_training_step_matches_oracle_autograd(name):
    from oracle import rsunet_oracle as RO
    from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet
    kw = CFGS[name]
    torch.manual_seed(11)
    m = RSUNet(1, 2, **kw).train()
    with torch.no_grad():                       # non-trivial affine parameters
        for mod in m.modules():
            if isinstance(mod, (torch.nn.GroupNorm, torch.nn.BatchNorm3d)):
                mod.weight.uniform_(0.7, 1.3)
                mod.bias.normal_(0, 0.2)
    x = torch.randn(2, 1, 8, 16, 16, generator=torch.Generator().manual_seed(12))
    tgt = torch.randn(2, 2, 8, 16, 16, generator=torch.Generator().manual_seed(13))
    params = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
    okw = {k: v for k, v in kw.items()}
    ref = RO.forward(params, x.clone(), bn_training=True, **okw)

    def loss_of(out):
        if isinstance(out, dict):
            l = F.mse_loss(out["output"], tgt)
            for k in sorted(out):
                if k != "output":
                    l = l + 0.5 * out[k].pow(2).mean()
            return l
        return F.mse_loss(out, tgt)

    ref_loss = loss_of(ref)
    ref_loss.backward()
    mg = m.cuda()
    out = mg(x.cuda())
    tg = tgt.cuda()
    if isinstance(out, dict):
        loss = F.mse_loss(out["output"], tg)
        for k in sorted(out):
            if k != "output":
                loss = loss + 0.5 * out[k].pow(2).mean()
    else:
        loss = F.mse_loss(out, tg)
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 1e-4 * max(1.0, abs(float(ref_loss.detach())))
    loss.backward()
    _check_grads([(n, p.grad.cpu()) for n, p in mg.named_parameters()], lambda n: params[n].grad, kw["activation"] != "elu")


def test_rsunet_bf16_training_direction():
    """bf16 storage: gradients keep the direction of the fp32 ones and SGD steps lower the loss.  On this random-target
    problem torch.autocast(bf16) through the oracle reaches cosines of 0.84-0.87 on the first layers and 0.998 on the last
    (ReLU masks flip on bf16-rounded pre-activations); the HIP path measures the same (tools/history/rs_bf16_fidelity.py)."""
    from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet
    torch.manual_seed(5)
    m = RSUNet(1, 1, width=[8, 16, 24], norm="group", num_groups=8, activation="relu").cuda().train()
    x = torch.randn(2, 1, 8, 32, 32, device="cuda")
    y = (torch.rand(2, 1, 8, 32, 32, device="cuda") > 0.5).float()
    F.binary_cross_entropy_with_logits(m(x), y).backward()
    ref = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad()
    m.compute_dtype = torch.bfloat16
    l0 = F.binary_cross_entropy_with_logits(m(x), y)
    l0.backward()
    for n, p in m.named_parameters():
        if p.grad.numel() < 100:
            continue
        g, r = p.grad.flatten(), ref[n].flatten()
        cos = float((g * r).sum() / (g.norm() * r.norm() + 1e-20))
        assert cos > (0.97 if n.startswith("up_blocks.1.conv") else 0.8), (n, cos)
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    for _ in range(10):
        opt.step()
        opt.zero_grad()
        l1 = F.binary_cross_entropy_with_logits(m(x), y)
        l1.backward()
    assert float(l1.detach()) < float(l0.detach())


def test_cli_train_mode_rsunet(tmp_path):
    """--mode train with the rsunet architecture (batch norm, deep supervision off), then test mode from the checkpoint."""
    from pytorch_connectomics_amd.main import main
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text(f"""
experiment_name: e2e_train_rsunet
save_path: {tmp_path / 'out'}
default:
  model:
    arch: {{type: rsunet}}
    in_channels: 1
    out_channels: 1
    input_size: [8, 32, 32]
    rsunet: {{width: [8, 12, 16], norm: batch, activation: elu}}
  data:
    dataloader: {{batch_size: 2, patch_size: [8, 32, 32]}}
  inference:
    window: {{window_size: [8, 32, 32], overlap: 0.5, sw_batch_size: 2}}
    model: {{channel_activations: [{{channels: ":", activation: sigmoid}}]}}
train:
  optimization:
    max_epochs: 1
    n_steps_per_epoch: 15
    optimizer: {{name: AdamW, lr: 5.0e-3, weight_decay: 0.01}}
test:
  data:
    test: {{image: "random://t?shape=12,40,40"}}
""")
    out = main(["--config", str(cfg), "--mode", "train"])
    assert out["steps"] == 15 and out["last_loss"] < out["first_loss"]
    ck = tmp_path / "out" / "checkpoints" / "last.ckpt"
    m = main(["--config", str(cfg), "--mode", "test", "--checkpoint", str(ck)])
    assert m["output_voxels_per_s"] > 0


def test_minimal_rsunet_tutorial_train_then_infer(tmp_path):
    """BASELINE.json configs[0]: tutorials/minimal_rsunet.yaml — 2 training steps, then sliding-window inference."""
    import os
    from pytorch_connectomics_amd.main import main
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tutorials", "minimal_rsunet.yaml")
    out = main(["--config", cfg, "--mode", "train", f"save_path={tmp_path / 'out'}"])
    assert out["steps"] == 2 and out["last_loss"] == out["last_loss"]
    ck = tmp_path / "out" / "checkpoints" / "last.ckpt"
    assert "model.input_conv.pre.1.weight" in torch.load(ck, weights_only=False)["state_dict"]
    res = main(["--config", cfg, "--mode", "test", "--checkpoint", str(ck), f"save_path={tmp_path / 'out'}"])
    assert res["output_voxels_per_s"] > 0


@pytest.mark.parametrize("ci,co,ks,shape", [(16, 16, (3, 3, 3), (3, 5, 32)), (32, 32, (3, 3, 3), (4, 6, 40)),
                                            (16, 32, (1, 3, 3), (2, 7, 20)), (64, 48, (3, 3, 3), (3, 4, 70)),
                                            (32, 16, (5, 3, 3), (6, 3, 33)), (1, 16, (3, 3, 3), (3, 5, 37)),
                                            (16, 3, (1, 1, 1), (3, 5, 37)), (32, 12, (1, 1, 1), (2, 3, 64)),
                                            (2, 32, (1, 3, 3), (2, 6, 31)), (3, 5, (3, 3, 3), (3, 4, 20)),
                                            (48, 16, (1, 1, 1), (2, 5, 33)),
                                            # channel counts that are multiples of 8 but not of the 16-wide tile (round 4: the
                                            # stock RSUNet widths padded to 24 / 40): partly filled last tiles on either side
                                            (24, 24, (1, 3, 3), (2, 6, 40)), (40, 40, (3, 3, 3), (3, 4, 33)),
                                            (24, 40, (3, 3, 3), (3, 5, 20)), (40, 24, (1, 1, 1), (2, 5, 33)),
                                            (48, 40, (1, 1, 1), (2, 3, 64)), (1, 24, (1, 3, 3), (2, 6, 31)),
                                            (24, 1, (1, 1, 1), (2, 6, 31)), (72, 8, (3, 3, 3), (2, 3, 32))])
def test_conv3d_wgrad_mfma_bf16(ci, co, ks, shape):
    """bf16 weight gradient on MFMA (LDS transpose reads) vs fp32 autograd on the same bf16-rounded operands, and
    vs the VALU kernel (tuning knob) on identical inputs."""
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(ci * 7 + co)
    x = torch.randn(2, ci, *shape).bfloat16().float().requires_grad_()
    w = torch.zeros(co, ci, *ks, requires_grad=True)
    y = F.conv3d(x, w, padding=tuple(k // 2 for k in ks))
    gy = torch.randn_like(y).bfloat16().float()
    y.backward(gy)
    xa, ga = _cl(x.detach()).cuda().bfloat16(), _cl(gy).cuda().bfloat16()
    dW = ops.conv3d_wgrad(xa, ga, ks)
    scale = float(w.grad.abs().max())
    assert float((dW.cpu() - w.grad).abs().max()) < 2e-5 * scale * 10 + 1e-3
    torch.testing.assert_close(dW.cpu(), w.grad, rtol=1e-4, atol=2e-4 * scale)
    ops.set_tuning("conv_wgrad_mfma", 0)
    try:
        dW2 = ops.conv3d_wgrad(xa, ga, ks)
    finally:
        ops.set_tuning("conv_wgrad_mfma", 1)
    torch.testing.assert_close(dW, dW2, rtol=1e-4, atol=2e-4 * scale)
    assert torch.equal(dW, ops.conv3d_wgrad(xa, ga, ks))          # deterministic


@pytest.mark.parametrize("name", list(CFGS))
def test_rsunet_training_step_matches_reference_fixture(name):
    """HIP forward + backward against the REFERENCE RSUNet's own training step (tests/golden/rsunet_train_*.npz,
    generated by tests/golden/make_golden.py --rsunet_train): loss, every parameter gradient, BatchNorm running buffers."""
    import numpy as np
    from pathlib import Path
    from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet
    z = np.load(Path(__file__).parent / "golden" / f"rsunet_train_{name}.npz")
    m = RSUNet(1, 2, **CFGS[name])
    m.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd__")}, strict=True)
    m = m.cuda().train()
    out = m(torch.from_numpy(z["x"]).cuda())
    t = torch.from_numpy(z["t"]).cuda()
    if isinstance(out, dict):
        loss = F.mse_loss(out["output"], t)
        for k in sorted(out):
            if k != "output":
                loss = loss + 0.5 * out[k].pow(2).mean()
    else:
        loss = F.mse_loss(out, t)
    ref_loss = float(z["loss"][0])
    assert abs(float(loss.detach()) - ref_loss) < 1e-4 * max(1.0, ref_loss)
    loss.backward()
    _check_grads([(n, p.grad.cpu()) for n, p in m.named_parameters()], lambda n: torch.from_numpy(z["grad__" + n]),
                 CFGS[name]["activation"] != "elu")
    for n, b in m.named_buffers():
        ref = torch.from_numpy(z["buf__" + n])
        if ref.dtype.is_floating_point:
            torch.testing.assert_close(b.cpu(), ref, rtol=1e-4, atol=1e-6)
        else:
            assert torch.equal(b.cpu(), ref), n


def test_rsunet_eval_after_training_uses_updated_batchnorm_buffers():
    """Eval-mode BatchNorm affines are cached per (buffer, version); the training kernels update the running buffers
    through raw pointers and must invalidate that cache."""
    from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet
    torch.manual_seed(9)
    m = RSUNet(1, 1, width=[8, 16], norm="batch", activation="relu").cuda()
    x = torch.randn(2, 1, 8, 16, 16, device="cuda") * 3 + 1
    with torch.no_grad():
        before = m.eval()(x).clone()
    m.train()
    for _ in range(3):
        m.zero_grad()
        m(x).mean().backward()
    with torch.no_grad():
        after = m.eval()(x)
        m._hip.cache.clear()
        fresh = m(x)
    assert float((after - before).abs().max()) > 1e-4
    assert torch.equal(after, fresh)


def test_rsunet_train_mode_forward_under_no_grad_uses_batch_statistics():
    """model.train() + torch.no_grad() with BatchNorm: batch statistics and running-buffer updates like nn.BatchNorm3d
    (the fused inference path only knows eval-mode affines)."""
    from oracle import rsunet_oracle as RO
    from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet
    torch.manual_seed(4)
    kw = dict(width=[4, 8], norm="batch", activation="relu")
    m = RSUNet(1, 2, **kw)
    st = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(2, 1, 8, 16, 16)
    ref = RO.forward(st, x.clone(), bn_training=True, **kw)
    m = m.cuda().train()
    rm0 = m.input_conv.pre[0].norm.running_mean.clone()
    with torch.no_grad():
        got = m(x.cuda())
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-4, atol=1e-4)
    assert not torch.equal(m.input_conv.pre[0].norm.running_mean, rm0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_vectorised_elementwise_kernels_are_bit_identical_to_the_scalar_ones(dtype):
    """affine_act / act_bwd / norm_bwd_apply_general in their 16-byte forms (C % 8 == 0 for bf16, % 4 for fp32) against the scalar
    kernels they replace (pytc_set_tuning("elementwise_vec", 0)): same arithmetic per element, so torch.equal."""
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    g = torch.Generator().manual_seed(9)
    N, D, H, W, C = 2, 5, 7, 9, 16
    x = torch.randn(N, D, H, W, C, generator=g).cuda().to(dtype)
    d = torch.randn(N, D, H, W, C, generator=g).cuda().to(dtype)
    ab = torch.randn(N, 2, C, generator=g).cuda()
    mr = torch.stack([torch.randn(N, C, generator=g), torch.rand(N, C, generator=g) + 0.5], 1).cuda().contiguous()
    M = torch.randn(N, 2, C, generator=g).cuda()
    gamma = torch.randn(C, generator=g).cuda()

    def run():
        outs = []
        for act, prm in ((nat.ACT_RELU, 0.0), (nat.ACT_LEAKY, 0.2), (nat.ACT_ELU, 1.0), (nat.ACT_NONE, 0.0)):
            outs.append(ops.affine_act(x, ab, act, prm))
            dt, dp = ops.act_bwd(d, x, ab, act, prm, want_prelu=act == nat.ACT_LEAKY)
            outs.append(dt)
            if dp is not None:
                outs.append(dp)
        outs.append(ops.affine_act(x, None, nat.ACT_SIGMOID, 0.0))
        outs.append(ops.norm_bwd_apply_general(d, x, mr, gamma, M))
        outs.append(ops.norm_bwd_apply_general(d, x, mr, None, M))
        return outs
    fast = run()
    ops.set_tuning("elementwise_vec", 0)
    try:
        slow = run()
    finally:
        ops.set_tuning("elementwise_vec", 1)
    assert len(fast) == len(slow) and all(torch.equal(a, b) for a, b in zip(fast, slow))


@pytest.mark.parametrize("kind,act,dtype", [("batch", "relu", torch.bfloat16), ("batch", "prelu", torch.bfloat16),
                                            ("group", "elu", torch.bfloat16), ("instance", "leakyrelu", torch.float32)])
def test_norm_act_backward_without_the_stored_activation_gradient_matches_the_three_pass_form(kind, act, dtype, monkeypatch):
    """pytc_act_norm_bwd_stats / _apply (dt = da * act'(t) recomputed in registers, never stored) against act_bwd ->
    norm_bwd_stats -> norm_bwd_apply_general: the same rounded dt values enter both, so dx agrees to the last bits of the storage
    type and the parameter gradients to fp32 summation order."""
    from pytorch_connectomics_amd.models.architectures.rsunet import NormAct
    from pytorch_connectomics_amd.training import rsunet_autograd as RA
    torch.manual_seed(5)
    C = 16
    na = NormAct(C, kind, act, num_groups=4, negative_slope=0.1, init=0.2, alpha=1.0).train().cuda()
    with torch.no_grad():
        if kind in ("group", "batch"):
            na.norm.weight.uniform_(0.5, 1.5)
            na.norm.bias.normal_()
    x = torch.randn(2, 9, 10, 11, C, device="cuda").to(dtype)
    gy = torch.randn(2, 9, 10, 11, C, device="cuda").to(dtype)
    res = {}
    for flag in (True, False):
        monkeypatch.setattr(RA, "FUSED_ACT_NORM_BWD", flag)
        na.zero_grad(set_to_none=True)
        xc = x.clone().requires_grad_()
        RA._norm_act(na, xc).backward(gy)
        res[flag] = (xc.grad.float().clone(), {n: p.grad.float().clone() for n, p in na.named_parameters() if p.grad is not None})
    tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(res[True][0], res[False][0], **tol)
    if dtype == torch.bfloat16:      # a last-bit rounding flip here and there (the means M differ in their last fp32 bits)
        differing = (res[True][0] != res[False][0]).float().mean().item()
        assert differing < 0.05, differing
    assert res[True][1].keys() == res[False][1].keys()
    for n, g in res[True][1].items():
        torch.testing.assert_close(g, res[False][1][n], rtol=1e-4, atol=1e-4)


def test_batched_conv_weight_packs_are_bit_identical_to_the_single_packs():
    """pytc_conv3d_pack_multi (every conv-weight image of a model in one launch per step) against the per-weight pack functions,
    all six layouts, flat and tap-major plans, bf16 and fp32 images; and a stale image is rebuilt by refresh()."""
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(0)
    packs = ops.ConvPackSet()
    cases = []
    for shape in [(16, 16, 3, 3, 3), (32, 16, 3, 3, 3), (64, 64, 3, 3, 3), (128, 128, 3, 3, 3), (8, 3, 1, 3, 3), (24, 20, 3, 3, 3),
                  (32, 64, 2, 2, 2), (1, 32, 1, 1, 1)]:
        w = torch.nn.Parameter(torch.randn(*shape, device="cuda"))
        for layout in ("fwd", "dgrad", "conv", "convT", "conv_dgrad", "convT_dgrad"):
            if layout in ("fwd", "dgrad") and any(k % 2 == 0 for k in shape[2:]):
                continue
            for dt in (torch.bfloat16, torch.float32):
                cases.append((w, layout, dt, packs.get(w, layout, dt)))
    assert len(packs.rows) == len(cases)
    with torch.no_grad():
        for w in {id(c[0]): c[0] for c in cases}.values():
            w.mul_(1.5).add_(0.25)                       # version moves: every image is stale
    for w, layout, dt, img in cases:
        assert not torch.equal(img, ops._conv_pack_single(w.detach(), layout, dt)) or w.numel() < 64
    packs.refresh()
    import ctypes
    from pytorch_connectomics_amd import _native as nat
    plan = (ctypes.c_int64 * 5)()
    for w, layout, dt, img in cases:
        want = ops._conv_pack_single(w.detach(), layout, dt)
        co, ci, _so, _sc, _flip, direct = ops._conv_layout_rule(w.shape, layout)
        nat.check(nat.lib().pytc_conv3d_pack_plan(co, ci, *w.shape[2:], ops.dtype_code(dt), direct, plan), "plan")
        n = int(plan[4])                                  # elements the pack writes (the allocation may be larger)
        bits = torch.int16 if dt == torch.bfloat16 else torch.int32
        assert img.shape == want.shape and 0 < n <= img.numel()
        assert torch.equal(img[:n].view(bits), want[:n].view(bits)), (w.shape, layout, dt)
        assert packs.get(w, layout, dt) is img           # a hit, no launch
    del cases, w, img, want
    import gc
    gc.collect()
    packs.refresh()
    assert not packs.rows                                # rows of dropped weights go away


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stock_rsunet_profile_forward_and_training_step_against_the_oracle(dtype):
    """RSUNet exactly as config/profiles/arch_profiles.yaml:34-44 ships it (width [18, 36, 48, 64, 80],
    GroupNorm(4) -> GroupNorm(3, 18) on the first level, ELU, down (1,2,2) x 4, depth_2d 1) at 18 x 64 x 64: inference forward and
    one training step (loss, every parameter gradient) against autograd through the oracle (rsunet_oracle == the reference,
    tests/test_oracle_golden.py).  The HIP path runs the two widths that are no multiple of 8 with zero-padded channels (24 / 40 in
    bf16, 20 / 36 in fp32) so that every kernel takes its 16-byte form -- state_dict shapes, outputs and gradients are the
    reference's.  fp32: tight; bf16: the direction of the whole gradient and the output error against the fp32 oracle."""
    from oracle import rsunet_oracle as RO
    from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet
    kw = dict(width=[18, 36, 48, 64, 80], norm="group", num_groups=4, activation="elu", down_factors=[(1, 2, 2)] * 4, depth_2d=1,
              kernel_2d=(1, 3, 3))
    torch.manual_seed(5)
    m = RSUNet(1, 1, **kw).train()
    gn = [mod for mod in m.modules() if isinstance(mod, torch.nn.GroupNorm)]
    assert {(g.num_groups, g.num_channels) for g in gn} >= {(3, 18), (4, 36), (4, 48), (4, 64), (4, 80)}
    with torch.no_grad():
        for mod in gn:
            mod.weight.uniform_(0.7, 1.3)
            mod.bias.normal_(0, 0.2)
    x = torch.rand(2, 1, 18, 64, 64, generator=torch.Generator().manual_seed(6))
    tgt = (torch.rand(2, 1, 18, 64, 64, generator=torch.Generator().manual_seed(7)) > 0.8).float()
    params = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
    ref = RO.forward(params, x.clone(), bn_training=True, **kw)
    ref_loss = F.binary_cross_entropy_with_logits(ref, tgt)
    ref_loss.backward()
    mg = m.cuda()
    mg.compute_dtype = dtype
    out = mg(x.cuda())
    assert tuple(out.shape) == (2, 1, 18, 64, 64)
    loss = F.binary_cross_entropy_with_logits(out.float(), tgt.cuda())
    loss.backward()
    for n, p in mg.named_parameters():
        assert p.grad is not None and tuple(p.grad.shape) == tuple(params[n].shape), n
    g = torch.cat([p.grad.flatten().double().cpu() for _n, p in mg.named_parameters()])
    r = torch.cat([params[n].grad.flatten().double() for n, _p in mg.named_parameters()])
    cos = float((g * r).sum() / (g.norm() * r.norm()))
    err_out = float((out.detach().float().cpu() - ref.detach()).abs().max())
    if dtype == torch.float32:
        assert abs(float(loss) - float(ref_loss)) < 1e-4 and err_out < 2e-3, (float(loss), float(ref_loss), err_out)
        _check_grads([(n, p.grad.cpu()) for n, p in mg.named_parameters()], lambda n: params[n].grad, False)
    else:
        assert abs(float(loss) - float(ref_loss)) < 2e-2 and cos > 0.97, (float(loss), float(ref_loss), cos)
    mg.eval()
    with torch.no_grad():
        y = mg(x.cuda())
        want = RO.forward({k: v.detach() for k, v in params.items()}, x.clone(), **kw)
    d = (y.float().cpu() - want).abs()
    if dtype == torch.float32:
        assert float(d.max()) < 2e-3
    else:       # bf16 storage through ~45 conv layers: judged against the spread of the logits, mean and worst voxel
        assert float(d.mean()) < 2e-2 * float(want.std()) and float(d.max()) < 0.12 * float(want.abs().max()), \
            (float(d.mean()), float(d.max()), float(want.std()), float(want.abs().max()))
