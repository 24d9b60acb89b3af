"""GPU parity of the RSUNet HIP forward against outputs of the REFERENCE's own rsunet.py (fixtures from
tests/golden/make_golden.py) -- the parity-pinned network of this engine -- plus its building-block kernels."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CFGS = {
    "c1_group": dict(width=[8, 16], down_factors=[(2, 2, 2)], norm="group", num_groups=8, activation="relu"),
    "aniso_inst_elu_ds": dict(width=[6, 8, 12], norm="instance", activation="elu", deep_supervision=True),
    "batch_prelu_2d": dict(width=[4, 8, 8], norm="batch", activation="prelu", depth_2d=1, init=0.1),
}


def _cl(x):
    return x.permute(0, 2, 3, 4, 1).contiguous()


def _cf(y):
    return y.permute(0, 4, 1, 2, 3).contiguous()


@pytest.mark.parametrize("name", list(CFGS))
def test_rsunet_matches_reference_outputs(name, golden_dir):
    from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet
    g = np.load(golden_dir / f"rsunet_{name}.npz")
    m = RSUNet(1, 2, **CFGS[name])
    m.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd__")}, strict=True)
    m = m.cuda().eval()
    x = torch.from_numpy(g["x"]).cuda()
    with torch.no_grad():
        y = m(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y16 = m(x)
    if not isinstance(y, dict):
        y, y16 = {"output": y}, {"output": y16}
    for k, v in y.items():
        exp = torch.from_numpy(g["y__" + k])
        assert v.shape == exp.shape and v.dtype == torch.float32
        torch.testing.assert_close(v.cpu(), exp, rtol=1e-4, atol=1e-4)          # fp32 path: north-star 1e-3
        assert (torch.sigmoid(v.cpu()) - torch.sigmoid(exp)).abs().max() < 1e-3
        d16 = (torch.sigmoid(y16[k].cpu()) - torch.sigmoid(exp)).abs()   # bf16 storage budget (random Kaiming
        assert d16.max() < 0.2 and d16.mean() < 2e-2                      # weights, logits up to |4|)
    # labels bit-exact where the reference margin exceeds the tolerance
    ref = torch.from_numpy(g["y__output"])
    margin = (ref[:, 0] - ref[:, 1]).abs() > 1e-3
    assert torch.equal(y["output"].cpu().argmax(1)[margin], ref.argmax(1)[margin])


def test_rsunet_none_norm_inplace_quirk_and_builders():
    """norm='none' + in-place activation aliases the residual (reference rsunet.py:103-113,150-154)."""
    from types import SimpleNamespace as NS
    from oracle import rsunet_oracle as RO
    from pytorch_connectomics_amd.models import build_model
    torch.manual_seed(1)
    cfg = NS(model=NS(arch=NS(type="rsunet_iso"), in_channels=1, out_channels=1,
                      rsunet=NS(width=[4, 8], norm="none", activation="relu", num_groups=8),
                      loss=NS(deep_supervision=False)))
    m = build_model(cfg)
    st = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(1, 1, 8, 16, 16)
    ref = RO.forward(st, x.clone(), width=[4, 8], down_factors=[(2, 2, 2)], norm="none", activation="relu")
    with torch.no_grad():
        got = m.cuda().eval()(x.cuda()).cpu()
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,k", [(8, 16, (3, 3, 3)), (16, 8, (1, 3, 3)), (1, 8, (3, 3, 3)), (18, 36, (3, 3, 3)),
                                        (6, 2, (1, 1, 1)), (16, 16, (3, 3, 3)), (32, 64, (3, 3, 3)), (24, 16, (3, 3, 3)),
                                        (64, 32, (1, 3, 3)), (128, 32, (3, 3, 3)), (48, 80, (3, 3, 3)),
                                        (32, 16, (5, 5, 5)), (16, 3, (1, 1, 1)), (8, 8, (3, 1, 1)), (24, 16, (1, 1, 1)),
                                        (40, 16, (3, 3, 3)), (24, 24, (1, 3, 3)), (2, 20, (3, 3, 3)), (3, 5, (1, 1, 1))])
def test_conv3d_with_fused_preactivation(dt, cin, cout, k):
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(cin + cout)
    N, D, H, W = (2, 5, 9, 11) if cin < 16 or cin == 18 else (2, 6, 17, 35)      # the larger one spans several LDS tiles
    x = torch.randn(N, cin, D, H, W).to(dt).float()
    w = torch.randn(cout, cin, *k) / (cin * k[0] * k[1] * k[2]) ** 0.5
    a, b = torch.rand(N, cin) + 0.5, torch.randn(N, cin) * 0.3
    res = torch.randn(N, cout, D, H, W).to(dt).float()
    bias = torch.randn(cout)
    pre = F.elu(x * a[:, :, None, None, None] + b[:, :, None, None, None], 0.7)
    if dt == torch.bfloat16:
        pre = pre.to(dt).float()
    ref = F.conv3d(pre, w.to(dt).float(), bias, padding=tuple(v // 2 for v in k)) + res
    wp = ops.conv3d_pack_weight(w.cuda(), dt)
    y = ops.conv3d(_cl(x).cuda().to(dt), wp, c_out=cout, kernel=k, bias=bias.cuda(),
                   ab=torch.stack([a, b], 1).contiguous().cuda(), act_in=nat.ACT_ELU, act_param=0.7,
                   res=_cl(res).cuda().to(dt))
    tol = dict(rtol=1e-4, atol=1e-4) if dt == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(_cf(y.float().cpu()), ref, **tol)


def test_norm_pool_upsample_kernels():
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    from pytorch_connectomics_amd.models.architectures.rsunet import BilinearUp3d
    torch.manual_seed(0)
    x = torch.randn(2, 12, 6, 10, 14)
    xc = _cl(x).cuda()
    st = ops.channel_stats(xc)
    torch.testing.assert_close(st.sum(1)[:, 0].cpu(), x.sum((2, 3, 4)), rtol=1e-4, atol=1e-3)
    gamma, beta = torch.rand(12) + 0.5, torch.randn(12)
    for groups in (12, 4, 1):
        ab = ops.norm_finalize_groups(st, 6 * 10 * 14, gamma.cuda(), beta.cuda(), 1e-5, groups).cpu()
        got = ab[:, 0][:, :, None, None, None] * x + ab[:, 1][:, :, None, None, None]
        torch.testing.assert_close(got, F.group_norm(x, groups, gamma, beta, 1e-5), rtol=1e-4, atol=1e-4)
    ab = ops.norm_finalize_groups(st, 6 * 10 * 14, None, None, 1e-5, 12)
    y = ops.affine_act(xc, ab, nat.ACT_LEAKY, 0.2)
    torch.testing.assert_close(_cf(y.cpu()), F.leaky_relu(F.instance_norm(x), 0.2), rtol=1e-4, atol=1e-4)
    for f in ((1, 2, 2), (2, 2, 2), (2, 5, 3)):
        torch.testing.assert_close(_cf(ops.maxpool3d(xc, f).cpu()), F.max_pool3d(x, f))
    for f in ((1, 2, 2), (2, 2, 2)):
        up = BilinearUp3d(12, 12, f)
        ref = F.conv_transpose3d(x, up.weight, stride=f, padding=up.padding, groups=12)
        taps = up.weight.reshape(12, -1).t().contiguous().cuda()
        got = ops.dwconvT3d_generic(xc, taps, up.kernel_size, up.factor, up.padding)
        torch.testing.assert_close(_cf(got.cpu()), ref, rtol=1e-5, atol=1e-5)


def test_sliding_window_with_rsunet_matches_oracle(golden_dir):
    """BASELINE config 1 shape class: RSUNet [8,16] iso GroupNorm through the device sliding-window engine."""
    from oracle import rsunet_oracle as RO
    from oracle import window_oracle as WO
    from pytorch_connectomics_amd.inference.window import EagerSlidingWindowEngine
    from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet
    g = np.load(golden_dir / "rsunet_c1_group.npz")
    kw = CFGS["c1_group"]
    m = RSUNet(1, 2, **kw)
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd__")}
    m.load_state_dict(sd)
    m = m.cuda().eval()
    vol = torch.rand(1, 1, 24, 40, 40, generator=torch.Generator().manual_seed(3))
    eng = EagerSlidingWindowEngine(roi_size=(16, 32, 32), sw_batch_size=2, overlap=0.5, mode="bump",
                                   padding_mode="constant", cval=0.0)
    got = eng(vol.cuda(), m).cpu()
    ref = WO.eager_sliding_window(vol, lambda x: RO.forward(sd, x, **kw), roi=(16, 32, 32), overlap=0.5, mode="bump",
                                  sw_batch_size=2)
    assert (torch.sigmoid(got) - torch.sigmoid(ref)).abs().max() < 1e-3
