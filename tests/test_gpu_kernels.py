"""GPU parity of the network kernels (depthwise conv + statistics, GroupNorm finalize, pointwise MFMA
GEMMs with their prologue/epilogue variants) against plain PyTorch fp32 on the CPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _cl(x):   # NCDHW -> NDHWC
    return x.permute(0, 2, 3, 4, 1).contiguous()


def _cf(y):
    return y.permute(0, 4, 1, 2, 3).contiguous()


TOL = {torch.float32: dict(rtol=1e-5, atol=1e-5), torch.bfloat16: dict(rtol=2e-2, atol=2e-2)}


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,K,stride,shape", [(32, 3, 1, (9, 10, 11)), (64, 3, 2, (9, 10, 12)), (16, 5, 1, (7, 8, 9)),
                                              (8, 7, 2, (9, 9, 9)), (12, 3, 1, (5, 6, 7)), (6, 3, 2, (6, 6, 6)),
                                              (512, 3, 1, (3, 4, 3)),
                                              # shapes that take the z-march fast path (incl. ragged tiles,
                                              # two channel groups, several z chunks)
                                              (32, 3, 1, (12, 20, 24)), (64, 3, 1, (9, 17, 19)),
                                              (32, 3, 1, (45, 16, 16)), (96, 3, 1, (8, 16, 33)),
                                              # K = 5 / 7: x-blocked kernel (bf16, C % 8 == 0), ragged W, stride 2 on the gather form
                                              (32, 5, 1, (7, 9, 18)), (64, 5, 1, (6, 8, 16)), (32, 7, 1, (8, 7, 13)),
                                              (32, 5, 2, (8, 10, 12))])
def test_dwconv3d_and_stats(dev, dt, C, K, stride, shape):
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(C * 100 + K)
    N = 2
    x = torch.randn(N, C, *shape)
    w = torch.randn(C, 1, K, K, K) * 0.2
    b = torch.randn(C)
    xq = x.to(dt).float()
    ref = F.conv3d(xq, w, b, stride=stride, padding=K // 2, groups=C)
    taps = w.reshape(C, K ** 3).t().contiguous().to(dev)
    y, st = ops.dwconv3d(_cl(xq).to(dev).to(dt), taps, b.to(dev), K=K, stride=stride)
    got = _cf(y.float().cpu())
    assert got.shape == ref.shape
    torch.testing.assert_close(got, ref, **TOL[dt])
    # statistics are those of the stored (rounded) tensor
    s = st.sum(1).cpu()
    torch.testing.assert_close(s[:, 0], got.sum((2, 3, 4)), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(s[:, 1], (got * got).sum((2, 3, 4)), rtol=1e-4, atol=1e-2)
    # finalize == GroupNorm(C, C) affine
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C)
    count = float(np.prod(got.shape[2:]))
    ab = ops.groupnorm_finalize(st, count, gamma.to(dev), beta.to(dev), 1e-5).cpu()
    normed = ab[:, 0][:, :, None, None, None] * got + ab[:, 1][:, :, None, None, None]
    torch.testing.assert_close(normed, F.group_norm(got, C, gamma, beta, 1e-5), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,K,shape", [(32, 3, (4, 5, 6)), (8, 5, (3, 4, 5)), (64, 7, (3, 3, 4))])
def test_dwconv_transposed(dev, dt, C, K, shape):
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(C + K)
    N = 2
    x = torch.randn(N, C, *shape).to(dt).float()
    w = torch.randn(C, 1, K, K, K) * 0.2
    b = torch.randn(C)
    ref = F.pad(F.conv_transpose3d(x, w, b, stride=2, padding=K // 2, groups=C), (1, 0, 1, 0, 1, 0))
    taps = w.reshape(C, K ** 3).t().contiguous().to(dev)
    y, st = ops.dwconv3d(_cl(x).to(dev).to(dt), taps, b.to(dev), K=K, transposed=True)
    got = _cf(y.float().cpu())
    assert got.shape == ref.shape
    torch.testing.assert_close(got, ref, **TOL[dt])
    inner = got[:, :, 1:, 1:, 1:]
    torch.testing.assert_close(st.sum(1).cpu()[:, 0], inner.sum((2, 3, 4)), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(st.sum(1).cpu()[:, 1], (inner * inner).sum((2, 3, 4)), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,rows", [(32, 64, 1000), (64, 32, 777), (1, 32, 300), (32, 1, 513), (8, 3, 100),
                                           (12, 20, 65), (512, 1024, 90), (1024, 512, 70), (128, 256, 260)])
def test_pw_conv_plain(dev, dt, cin, cout, rows):
    """Asymmetric random weights: catches any operand / accumulator transposition."""
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(cin * 7 + cout)
    N = 2
    x = torch.randn(N, rows, cin).to(dt).float()
    w = (torch.randn(cout, cin) / cin ** 0.5)
    b = torch.randn(cout)
    wq = w.to(dt).float()
    ref = x @ wq.t() + b
    wp = ops.pw_pack_weight(w.to(dev), dt)
    y = ops.pw_conv(x.to(dev).to(dt), wp, b.to(dev), N=N, rows_per_sample=rows, c_in=cin, c_out=cout, out_dtype=dt)
    torch.testing.assert_close(y.float().cpu(), ref, **TOL[dt])
    # transposed source layout (ConvTranspose3d weight) gives the same operator
    wp2 = ops.pw_pack_weight(w.t().contiguous().to(dev), dt, transposed=True)
    assert torch.equal(wp, wp2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_pw_conv_norm_gelu_and_residual(dev, dt):
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(3)
    N, rows, C, Hd = 2, 500, 32, 64
    t = torch.randn(N, rows, C).to(dt).float()
    xres = torch.randn(N, rows, C).to(dt).float()
    a, b = torch.rand(N, C) + 0.5, torch.randn(N, C)
    w2, b2 = torch.randn(Hd, C) / C ** 0.5, torch.randn(Hd)
    w3, b3 = torch.randn(C, Hd) / Hd ** 0.5, torch.randn(C)
    ab = torch.stack([a, b], 1).contiguous().to(dev)
    h = ops.pw_conv(t.to(dev).to(dt), ops.pw_pack_weight(w2.to(dev), dt), b2.to(dev), N=N, rows_per_sample=rows,
                    c_in=C, c_out=Hd, out_dtype=dt, ab=ab, act=nat.ACT_GELU)
    tn = (t * a[:, None] + b[:, None])
    if dt == torch.bfloat16:
        tn = tn.to(dt).float()
    href = F.gelu(tn @ w2.to(dt).float().t() + b2)
    torch.testing.assert_close(h.float().cpu(), href, **TOL[dt])
    y = ops.pw_conv(h, ops.pw_pack_weight(w3.to(dev), dt), b3.to(dev), N=N, rows_per_sample=rows, c_in=Hd, c_out=C,
                    out_dtype=dt, res=xres.to(dev).to(dt), res_mode=nat.RES_ADD)
    yref = h.float().cpu() @ w3.to(dt).float().t() + b3 + xres
    torch.testing.assert_close(y.float().cpu(), yref, **TOL[dt])


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_pw_conv_strided_gather_and_upsample_epilogue(dev, dt):
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(4)
    N, C, Co = 2, 16, 32
    D, H, W = 5, 6, 7
    x = torch.randn(N, C, D, H, W).to(dt).float()
    w, b = torch.randn(Co, C, 1, 1, 1) / 4, torch.randn(Co)
    ref = F.conv3d(x, w.to(dt).float(), b, stride=2)
    Do, Ho, Wo = ref.shape[2:]
    y = ops.pw_conv(_cl(x).to(dev).to(dt), ops.pw_pack_weight(w.reshape(Co, C).to(dev), dt), b.to(dev), N=N,
                    rows_per_sample=Do * Ho * Wo, c_in=C, c_out=Co, out_dtype=dt, gather=2, grid=(D, H, W))
    torch.testing.assert_close(_cf(y.view(N, Do, Ho, Wo, Co).float().cpu()), ref, **TOL[dt])

    # up-block epilogue: pad(mlp) + pad(convT_1x1_s2(x_low)) + skip
    Cl, Cu = 32, 16
    d, h_, w_ = 3, 4, 5
    xl = torch.randn(N, Cl, d, h_, w_).to(dt).float()
    hid = torch.randn(N, 24, 2 * d, 2 * h_, 2 * w_).to(dt).float()   # hidden activations on the padded grid
    skip = torch.randn(N, Cu, 2 * d, 2 * h_, 2 * w_).to(dt).float()
    w3, b3 = torch.randn(Cu, 24, 1, 1, 1) / 5, torch.randn(Cu)
    wr, br = torch.randn(Cl, Cu, 1, 1, 1) / 6, torch.randn(Cu)      # ConvTranspose3d layout (C_in, C_out)
    mlp = F.conv3d(hid[:, :, 1:, 1:, 1:], w3.to(dt).float(), b3)
    res = F.conv_transpose3d(xl, wr.to(dt).float(), br, stride=2)
    ref = F.pad(mlp, (1, 0, 1, 0, 1, 0)) + F.pad(res, (1, 0, 1, 0, 1, 0)) + skip
    res_low = ops.pw_conv(_cl(xl).to(dev).to(dt), ops.pw_pack_weight(wr.reshape(Cl, Cu).to(dev), dt, transposed=True),
                          br.to(dev), N=N, rows_per_sample=d * h_ * w_, c_in=Cl, c_out=Cu, out_dtype=dt)
    rows = 8 * d * h_ * w_
    y = ops.pw_conv(_cl(hid).to(dev).to(dt), ops.pw_pack_weight(w3.reshape(Cu, 24).to(dev), dt), b3.to(dev), N=N,
                    rows_per_sample=rows, c_in=24, c_out=Cu, out_dtype=dt, res=_cl(skip).to(dev).to(dt),
                    res_mode=nat.RES_UPSAMPLE, grid=(2 * d, 2 * h_, 2 * w_), res_low=res_low, res_bias=br.to(dev))
    tol = dict(TOL[dt])
    if dt == torch.bfloat16:
        tol = dict(rtol=3e-2, atol=5e-2)   # res_low is itself rounded to bf16
    torch.testing.assert_close(_cf(y.view(N, 2 * d, 2 * h_, 2 * w_, Cu).float().cpu()), ref, **tol)


@pytest.mark.parametrize("cin,chid,cout,mode", [
    (32, 64, 32, "add"), (32, 64, 64, "add"), (64, 128, 32, "up"), (64, 128, 64, "add"), (64, 192, 128, "none"),
    (128, 256, 128, "add"), (128, 512, 64, "up"), (256, 512, 256, "add"), (256, 1024, 512, "add"),
    (512, 1024, 512, "add"), (512, 1024, 256, "up"), (128, 256, 256, "add"), (256, 2048, 128, "up"),
])
@pytest.mark.parametrize("hidden", ["bf16", "f16"])
def test_pw_mlp_fused_matches_reference(dev, cin, chid, cout, mode, hidden):
    """Fused norm-apply/expand/GELU/project/residual kernel vs fp32 math with rounding at the same three points (normalised
    input -> bf16, hidden activation -> bf16 or fp16, output -> bf16).  hidden = "f16": the projection weights are the fp16
    image, the hidden activation comes from the packed-fp16 polynomial GELU and feeds the f16 MFMA (the default of the models)."""
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    assert ops.pw_mlp_supported(cin, chid, cout)
    torch.manual_seed(cin + chid + cout)
    bf = torch.bfloat16
    N = 2
    if mode == "up":
        d, h_, w_ = 3, 2, 3
        grid = (2 * d, 2 * h_, 2 * w_)
        rows = grid[0] * grid[1] * grid[2]
    else:
        rows, grid = 333, (0, 0, 0)
    t = torch.randn(N, rows, cin).to(bf).float()
    a, b = torch.rand(N, cin) + 0.5, torch.randn(N, cin) * 0.5
    w2, b2 = torch.randn(chid, cin) / cin ** 0.5, torch.randn(chid) * 0.5
    w3, b3 = torch.randn(cout, chid) / chid ** 0.5, torch.randn(cout) * 0.5
    tn = (t * a[:, None] + b[:, None]).to(bf).float()
    hdt = torch.float16 if hidden == "f16" else bf
    hid = F.gelu(tn @ w2.to(bf).float().t() + b2).to(hdt).float()
    core = hid @ w3.to(hdt).float().t() + b3
    ab = torch.stack([a, b], 1).contiguous().to(dev)
    args = dict(N=N, rows_per_sample=rows, c_in=cin, c_hid=chid, c_out=cout)
    w2p, w3p = ops.pw_pack_weight_paired(w2.to(dev)), ops.pw_pack_weight_paired(w3.to(dev), f16=hidden == "f16")
    assert w3p.dtype == hdt
    if mode == "add":
        res = torch.randn(N, rows, cout).to(bf).float()
        ref = core + res
        y = ops.pw_mlp(t.to(dev).to(bf), ab, w2p, b2.to(dev), w3p, b3.to(dev), res=res.to(dev).to(bf),
                       res_mode=nat.RES_ADD, **args)
    elif mode == "none":
        ref = core
        y = ops.pw_mlp(t.to(dev).to(bf), ab, w2p, b2.to(dev), w3p, b3.to(dev), **args)
    else:
        skip = torch.randn(N, rows, cout).to(bf).float()
        res_low = torch.randn(N, d * h_ * w_, cout).to(bf).float()
        rbias = torch.randn(cout)
        c5 = core.view(N, *grid, cout)
        ref = torch.zeros_like(c5)
        ref[:, 1:, 1:, 1:] = c5[:, 1:, 1:, 1:] + rbias
        rl = res_low.view(N, d, h_, w_, cout)
        # transposed 1x1 stride-2 residual lands on padded positions 1,3,5.. (o = p-1 even)
        ref[:, 1::2, 1::2, 1::2] = c5[:, 1::2, 1::2, 1::2] + rl
        ref = ref + skip.view(N, *grid, cout)
        ref = ref.view(N, rows, cout)
        y = ops.pw_mlp(t.to(dev).to(bf), ab, w2p, b2.to(dev), w3p, b3.to(dev), res=skip.to(dev).to(bf),
                       res_mode=nat.RES_UPSAMPLE, grid=grid, res_low=res_low.to(dev).to(bf),
                       res_bias=rbias.to(dev), **args)
    torch.testing.assert_close(y.float().cpu(), ref, rtol=2e-2, atol=3e-2)


@pytest.mark.parametrize("C,chid,cout,mode", [(32, 64, 32, "add"), (64, 128, 32, "up"), (64, 128, 64, "add"), (128, 256, 128, "none"),
                                               (128, 512, 64, "add")])
def test_groupnorm_fold_into_the_expanding_conv(dev, C, chid, cout, mode):
    """pytc_groupnorm_fold_mlp (round 4): the statistics slots become per-sample expand operands W2 * diag(a_n), b2 + W2 b_n --
    (i) the affine it derives equals groupnorm_finalize's, (ii) the image is the paired bf16 image of the scaled weight, bit for
    bit, (iii) the folded bias matches fp64, and (iv) the mixer fed the RAW tensor and these operands equals the fp32 math
    W3 gelu(W2 (a t + b) + b2) + b3 (+ residual) within the fused kernel's own tolerance."""
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(C + chid)
    bf = torch.bfloat16
    N, slots = 3, 37
    if mode == "up":
        grid = (4, 6, 4)
        rows = 96
    else:
        rows, grid = 333, (0, 0, 0)
    t = (torch.randn(N, rows, C) * 1.7 + torch.randn(1, 1, C)).to(bf)
    # statistics slots whose sums are the true column sums (split unevenly over the slots)
    tf = t.float()
    s1, s2 = tf.sum(1), (tf * tf).sum(1)                                   # (N, C)
    wts = torch.rand(N, slots, 1)
    wts = wts / wts.sum(1, keepdim=True)
    stats = torch.stack([wts * s1[:, None], wts * s2[:, None]], 2).contiguous().to(dev)      # (N, slots, 2, C)
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C) * 0.3
    w2, b2 = torch.randn(chid, C) / C ** 0.5, torch.randn(chid) * 0.5
    w3, b3 = torch.randn(cout, chid) / chid ** 0.5, torch.randn(cout) * 0.5
    ab_ref = ops.groupnorm_finalize(stats, float(rows), gamma.to(dev), beta.to(dev), 1e-5)
    w2n, b2n, ab = ops.groupnorm_fold_mlp(stats, float(rows), gamma.to(dev), beta.to(dev), 1e-5, w2.to(dev), b2.to(dev), want_ab=True)
    torch.testing.assert_close(ab, ab_ref, rtol=2e-6, atol=2e-6)
    a, b = ab[:, 0].cpu(), ab[:, 1].cpu()
    for n in range(N):
        img = ops.pw_pack_weight_paired((w2 * a[n][None, :]).to(dev))
        assert torch.equal(img.view(torch.int16), w2n[n].view(torch.int16)), n
    # (iii) the bias is centred with the ROUNDED folded weights: b2 + W2 beta - bf16(W2 a) mean, in fp64
    mean = (s1 / rows).double()                                            # (N, C)
    w2r = torch.stack([(w2 * a[n][None, :]).to(bf).double() for n in range(N)])      # (N, chid, C)
    want_b = (b2.double()[None] + (w2.double() @ beta.double())[None] - torch.einsum("nok,nk->no", w2r, mean)).float()
    torch.testing.assert_close(b2n.cpu(), want_b, rtol=2e-5, atol=2e-5)
    # the mixer on raw t
    hid = F.gelu((tf * a[:, None] + b[:, None]) @ w2.t() + b2)
    core = hid @ w3.t() + b3
    w3p = ops.pw_pack_weight_paired(w3.to(dev), f16=True)
    args = dict(N=N, rows_per_sample=rows, c_in=C, c_hid=chid, c_out=cout)
    if mode == "add":
        res = torch.randn(N, rows, cout).to(bf)
        ref = core + res.float()
        y = ops.pw_mlp(t.to(dev), None, w2n, b2n, w3p, b3.to(dev), res=res.to(dev), res_mode=nat.RES_ADD, **args)
        y_aff = ops.pw_mlp(t.to(dev), ab, ops.pw_pack_weight_paired(w2.to(dev)), b2.to(dev), w3p, b3.to(dev), res=res.to(dev),
                           res_mode=nat.RES_ADD, **args)
    elif mode == "none":
        ref = core
        y = ops.pw_mlp(t.to(dev), None, w2n, b2n, w3p, b3.to(dev), **args)
        y_aff = ops.pw_mlp(t.to(dev), ab, ops.pw_pack_weight_paired(w2.to(dev)), b2.to(dev), w3p, b3.to(dev), **args)
    else:
        skip = torch.randn(N, rows, cout).to(bf)
        c5 = core.view(N, *grid, cout)
        ref = torch.zeros_like(c5)
        ref[:, 1:, 1:, 1:] = c5[:, 1:, 1:, 1:]
        ref = (ref + skip.float().view(N, *grid, cout)).view(N, rows, cout)
        kw = dict(res=skip.to(dev), res_mode=nat.RES_UPSAMPLE, grid=grid, **args)
        y = ops.pw_mlp(t.to(dev), None, w2n, b2n, w3p, b3.to(dev), **kw)
        y_aff = ops.pw_mlp(t.to(dev), ab, ops.pw_pack_weight_paired(w2.to(dev)), b2.to(dev), w3p, b3.to(dev), **kw)
    torch.testing.assert_close(y.float().cpu(), ref, rtol=2e-2, atol=3e-2)
    # and no further from the fp32 math than the affine-prologue form is (mean absolute error within 25 %)
    e_fold, e_aff = (y.float().cpu() - ref).abs().mean(), (y_aff.float().cpu() - ref).abs().mean()
    assert float(e_fold) <= 1.25 * float(e_aff) + 1e-4, (float(e_fold), float(e_aff))
    with pytest.raises(ValueError, match="norm-folded mixer operands"):
        ops.pw_mlp(t.to(dev), None, w2n[:1], b2n, w3p, b3.to(dev), **args)


@pytest.mark.parametrize("offset", [0.0, 8.0, 64.0])
def test_norm_fold_with_large_channel_offsets(dev, offset):
    """channels with |mean| >> std (a large depthwise bias).  The folded form multiplies ROUNDED weights bf16(W2 a) by the raw
    activation; with the bias folded at the unrounded product every weight's rounding error met the channel MEAN (error ~ |mean| / std
    times the affine form's).  Centred with the rounded weights (groupnorm_fold_mlp_kernel) the error meets only the spread: the folded
    mixer stays within 1.25x of the affine-prologue form's distance from the fp64 math at any offset."""
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(int(offset) + 3)
    bf = torch.bfloat16
    N, rows, C, chid, cout, slots = 2, 2000, 32, 64, 32, 11
    t = (torch.randn(N, rows, C) + offset * (torch.rand(1, 1, C) + 0.5) * torch.sign(torch.randn(1, 1, C))).to(bf)
    tf = t.double()
    s1, s2 = tf.sum(1), (tf * tf).sum(1)
    wts = torch.full((N, slots, 1), 1.0 / slots, dtype=torch.float64)
    stats = torch.stack([wts * s1[:, None], wts * s2[:, None]], 2).float().contiguous().to(dev)
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C) * 0.3
    w2, b2 = torch.randn(chid, C) / C ** 0.5, torch.randn(chid) * 0.5
    w3, b3 = torch.randn(cout, chid) / chid ** 0.5, torch.randn(cout) * 0.5
    w2n, b2n, ab = ops.groupnorm_fold_mlp(stats, float(rows), gamma.to(dev), beta.to(dev), 1e-5, w2.to(dev), b2.to(dev), want_ab=True)
    mean = s1 / rows
    var = s2 / rows - mean * mean
    a = gamma.double()[None] / torch.sqrt(var + 1e-5)
    b = beta.double()[None] - mean * a
    hid = F.gelu((tf * a[:, None] + b[:, None]) @ w2.double().t() + b2.double())
    ref = (hid @ w3.double().t() + b3.double()).float()
    w3p = ops.pw_pack_weight_paired(w3.to(dev), f16=True)
    args = dict(N=N, rows_per_sample=rows, c_in=C, c_hid=chid, c_out=cout)
    y = ops.pw_mlp(t.to(dev), None, w2n, b2n, w3p, b3.to(dev), **args).float().cpu()
    y_aff = ops.pw_mlp(t.to(dev), ab, ops.pw_pack_weight_paired(w2.to(dev)), b2.to(dev), w3p, b3.to(dev), **args).float().cpu()
    e_fold, e_aff = float((y - ref).abs().mean()), float((y_aff - ref).abs().mean())
    assert e_fold <= 1.25 * e_aff + 1e-4, (offset, e_fold, e_aff)


@pytest.mark.parametrize("cin,chid,cout,mode,N,rows", [(256, 512, 256, "add", 3, 1000), (512, 1024, 512, "add", 8, 343),
                                                         (512, 1024, 256, "up", 2, 96), (128, 256, 128, "none", 2, 2197),
                                                         (256, 512, 512, "add", 1, 17), (128, 512, 128, "add", 5, 4096)])
@pytest.mark.parametrize("rows_knob", [0, 64, 128])
def test_deep_level_gemm_pair_is_bit_identical_to_the_fused_mixer(dev, cin, chid, cout, mode, N, rows, rows_knob):
    """ops.pw_gemm x 2 (round 4: the deep levels' convs as LDS-tiled GEMM launches) against ops.pw_mlp with the fp16 projection: the
    same operations in the same order -- bias-initialised accumulators, k ascending in steps of 32, GroupNorm affine in fp32 rounded
    to bf16, packed-fp16 GELU, fp16 hidden, f16 MFMA projection, the shared residual epilogue -- hence equal BITS, for both workgroup
    heights, rows that are no multiple of the tile and tiles that straddle samples."""
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(cin + rows)
    bf = torch.bfloat16
    grid = (4, 6, 4) if mode == "up" else (0, 0, 0)
    t = torch.randn(N, rows, cin, device=dev).to(bf)
    ab = torch.stack([torch.rand(N, cin, device=dev) + 0.5, torch.randn(N, cin, device=dev) * 0.5], 1).contiguous()
    w2, b2 = torch.randn(chid, cin, device=dev) / cin ** 0.5, torch.randn(chid, device=dev) * 0.5
    w3, b3 = torch.randn(cout, chid, device=dev) / chid ** 0.5, torch.randn(cout, device=dev) * 0.5
    res = torch.randn(N, rows, cout, device=dev).to(bf)
    kw = dict(N=N, rows_per_sample=rows)
    ekw = {}
    if mode == "add":
        ekw = dict(res=res, res_mode=nat.RES_ADD)
    elif mode == "up":
        low = torch.randn(N, rows // 8, cout, device=dev).to(bf)
        ekw = dict(res=res, res_mode=nat.RES_UPSAMPLE, grid=grid, res_low=low, res_bias=b3)
    want = ops.pw_mlp(t, ab, ops.pw_pack_weight_paired(w2), b2, ops.pw_pack_weight_paired(w3, f16=True), b3, c_in=cin, c_hid=chid,
                      c_out=cout, **kw, **ekw)
    ops.set_tuning("pw_gemm_rows", rows_knob)
    try:
        h = ops.pw_gemm(t, w2.to(bf).contiguous(), b2, ab=ab, gelu=True, **kw)
        got = ops.pw_gemm(h, w3.to(torch.float16).contiguous(), b3, **kw, **ekw)
    finally:
        ops.set_tuning("pw_gemm_rows", 0)
    assert h.dtype == torch.float16 and got.dtype == bf and got.shape == want.shape
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())


@pytest.mark.parametrize("cin,chid,cout,mode,N,grid", [(64, 128, 64, "add", 3, (19, 19, 19)), (128, 256, 64, "up", 5, (14, 14, 14)),
                                                       (128, 256, 128, "add", 2, (13, 11, 9)), (64, 128, 32, "up", 3, (18, 16, 22)),
                                                       (64, 128, 64, "none", 1, (40, 40, 40)), (128, 256, 64, "up", 1, (2, 2, 2)),
                                                       (128, 256, 128, "add", 8, (28, 28, 28))])
@pytest.mark.parametrize("folded", [True, False])
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
def test_lds_resident_mixer_is_bit_identical(dev, cin, chid, cout, mode, N, grid, folded, variant):
    """ops.pw_mlp(lds=True) -- the persistent mixer with both weight images in LDS (round 4, pw_mlp_lds_kernels.hip) -- against the
    streaming kernel: same MFMA order, GELU and epilogue, hence equal BITS; ragged last tiles, shares of the (sample, tile) sequence
    that cross samples (per-sample images re-staged), fewer tiles than workgroups, every launch variant."""
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    assert ops.pw_mlp_lds_supported(cin, chid, cout) and not ops.pw_mlp_lds_supported(32, 64, 32)
    torch.manual_seed(cin + grid[0])
    bf = torch.bfloat16
    rows = grid[0] * grid[1] * grid[2]
    t = torch.randn(N, rows, cin, device=dev).to(bf)
    w2f = torch.randn(chid, cin, device=dev) / cin ** 0.5
    w3 = ops.pw_pack_weight_paired(torch.randn(cout, chid, device=dev) / chid ** 0.5, f16=True)
    b3 = torch.randn(cout, device=dev) * 0.5
    if folded:
        ab = None
        w2 = torch.stack([ops.pw_pack_weight_paired(w2f * (1 + 0.25 * n)) for n in range(N)])
        b2 = torch.randn(N, chid, device=dev) * 0.5
    else:
        ab = torch.stack([torch.rand(N, cin, device=dev) + 0.5, torch.randn(N, cin, device=dev) * 0.5], 1).contiguous()
        w2, b2 = ops.pw_pack_weight_paired(w2f), torch.randn(chid, device=dev) * 0.5
    kw = dict(N=N, rows_per_sample=rows, c_in=cin, c_hid=chid, c_out=cout)
    res = torch.randn(N, rows, cout, device=dev).to(bf)
    if mode == "add":
        kw.update(res=res, res_mode=nat.RES_ADD)
    elif mode == "up":
        low = torch.randn(N, rows // 8, cout, device=dev).to(bf)
        kw.update(res=res, res_mode=nat.RES_UPSAMPLE, grid=grid, res_low=low, res_bias=b3)
    want = ops.pw_mlp(t, ab, w2, b2, w3, b3, **kw)
    got = torch.full_like(want, float("nan"))
    ops.set_tuning("mlp_lds_variant", variant)
    try:
        ops.pw_mlp(t, ab, w2, b2, w3, b3, y=got, lds=True, **kw)
    finally:
        ops.set_tuning("mlp_lds_variant", 0)
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())


@pytest.mark.parametrize("N,rows,chid,n_head", [(1, 1, 64, 1), (2, 63, 64, 3), (3, 1000, 96, 16), (2, 4097, 128, 5), (5, 777, 64, 2)])
@pytest.mark.parametrize("wgs", [0, 1, 3])
def test_dma_prefetch_mixer_is_bit_identical(dev, N, rows, chid, n_head, wgs):
    """pw_mlp_dma_kernel (round 5): the level-0 mixer whose operand rows land in LDS by `global_load_lds` one tile ahead of the tile a
    wave computes, results stored one tile late -- plain, residual-add, recomputed-stem-residual and fused-head forms -- against the
    one-tile-per-wave kernel it replaces above `mlp_dma_rows` rows: same MFMA order, GELU and epilogue, hence equal BITS; single-row and
    ragged samples, fewer tiles than waves (wgs 0 = the launch's own grid), waves that walk many tiles (1 workgroup per sample)."""
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    bf = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(rows)
    t = torch.randn(N, rows, 32, generator=g).to(bf).to(dev)
    res = torch.randn(N, rows, 32, generator=g).to(bf).to(dev)
    w2n = torch.stack([ops.pw_pack_weight_paired((torch.randn(chid, 32, generator=g) / 32 ** 0.5).to(dev)) for _ in range(N)])
    b2n = torch.randn(N, chid, generator=g).to(dev)
    w3 = ops.pw_pack_weight_paired((torch.randn(32, chid, generator=g) / chid ** 0.5).to(dev), f16=True)
    b3 = torch.randn(32, generator=g).to(dev)
    x0 = torch.randn(N, rows, generator=g).to(dev)
    sw, sb = torch.randn(32, generator=g).to(dev), torch.randn(32, generator=g).to(dev)
    hw = ops.pack_head_fragment((torch.randn(n_head, 32, generator=g) / 32 ** 0.5).to(dev))
    hb = torch.randn(n_head, generator=g).to(dev)
    kw = dict(N=N, rows_per_sample=rows, c_in=32, c_hid=chid, c_out=32)

    def everything():
        out = [ops.pw_mlp(t, None, w2n, b2n, w3, b3, **kw), ops.pw_mlp(t, None, w2n, b2n, w3, b3, res=res, res_mode=nat.RES_ADD, **kw),
               ops.pw_mlp_stemres(t, None, w2n, b2n, w3, b3, x0, sw, sb, **kw)]
        for add in (False, True):
            for store_y in (False, True):
                out += list(ops.pw_mlp_head(t, None, w2n, b2n, w3, b3, hw, hb if add else None, res=res if add else None, store_y=store_y, **kw))
        torch.cuda.synchronize()
        return out

    ops.set_tuning("mlp_dma_rows", 0)
    try:
        ops.set_tuning("mlp_dma", 0)
        want = everything()
        ops.set_tuning("mlp_dma", 1)
        ops.set_tuning("mlp_dma_wgs", wgs)
        got = everything()
    finally:
        ops.set_tuning("mlp_dma", 1)
        ops.set_tuning("mlp_dma_wgs", 0)
        ops.set_tuning("mlp_dma_rows", 1 << 20)
    assert len(want) == len(got) == 11
    for i, (a, b) in enumerate(zip(want, got)):
        assert (a is None) == (b is None), i
        if a is not None:
            assert torch.equal(a, b), (i, float((a.float() - b.float()).abs().max()))


@pytest.mark.parametrize("cin,chid,cout,mode,N,grid", [(256, 64, 128, "none", 1, (5, 5, 5)),       # first: the narrow instance of <8, 8> opts into > 64 KB of LDS
                                                       (128, 1024, 128, "add", 2, (7, 7, 7)), (128, 1024, 128, "none", 1, (3, 3, 3)),
                                                       (256, 2048, 128, "up", 2, (6, 6, 6)), (128, 512, 64, "up", 3, (4, 4, 4)),
                                                       (64, 512, 128, "none", 2, (9, 9, 9)), (128, 96, 128, "add", 2, (5, 5, 5)),
                                                       (128, 64, 64, "add", 1, (4, 4, 4)), (64, 128, 32, "up", 2, (6, 6, 6)),
                                                       (64, 256, 64, "add", 2, (7, 7, 7)), (128, 256, 64, "up", 2, (14, 14, 14))])
@pytest.mark.parametrize("folded", [True, False])
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
def test_chunk_streamed_mixer_is_bit_identical(dev, cin, chid, cout, mode, N, grid, folded, variant):
    """ops.pw_mlp(chunked=True) -- the mixer whose workgroups stream both weight images through an LDS ring one 32-wide hidden chunk at a
    time by LDS-DMA (round 5, pw_mlp_chunk_kernels.hip: MedNeXt-L's wide hidden layers) -- against the streaming kernel: same MFMA order,
    GELU and epilogue, hence equal BITS; one / two / three chunks (ring start-up and drain), ragged last workgroups, waves without rows,
    every launch variant (8 / 12 / 16 waves: 3, 2, 1 or no DMA pieces per wave and chunk)."""
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    assert ops.pw_mlp_chunk_supported(cin, chid, cout) and not ops.pw_mlp_chunk_supported(32, 64, 32)
    bf = torch.bfloat16
    rows = grid[0] * grid[1] * grid[2]
    g = torch.Generator(device="cpu").manual_seed(cin + rows)
    t = torch.randn(N, rows, cin, generator=g).to(bf).to(dev)
    w3 = ops.pw_pack_weight_paired((torch.randn(cout, chid, generator=g) / chid ** 0.5).to(dev), f16=True)
    b3 = (torch.randn(cout, generator=g) * 0.5).to(dev)
    if folded:
        ab = None
        w2 = torch.stack([ops.pw_pack_weight_paired((torch.randn(chid, cin, generator=g) / cin ** 0.5).to(dev)) for _ in range(N)])
        b2 = (torch.randn(N, chid, generator=g) * 0.5).to(dev)
    else:
        ab = torch.stack([torch.rand(N, cin, generator=g) + 0.5, torch.randn(N, cin, generator=g) * 0.5], 1).contiguous().to(dev)
        w2, b2 = ops.pw_pack_weight_paired((torch.randn(chid, cin, generator=g) / cin ** 0.5).to(dev)), (torch.randn(chid, generator=g) * 0.5).to(dev)
    kw = dict(N=N, rows_per_sample=rows, c_in=cin, c_hid=chid, c_out=cout)
    res = torch.randn(N, rows, cout, generator=g).to(bf).to(dev)
    if mode == "add":
        kw.update(res=res, res_mode=nat.RES_ADD)
    elif mode == "up":
        low = torch.randn(N, rows // 8, cout, generator=g).to(bf).to(dev)
        kw.update(res=res, res_mode=nat.RES_UPSAMPLE, grid=grid, res_low=low, res_bias=b3)
    want = ops.pw_mlp(t, ab, w2, b2, w3, b3, **kw)
    got = torch.full_like(want, float("nan"))
    ops.set_tuning("mlp_chunk_variant", variant)
    try:
        ops.pw_mlp(t, ab, w2, b2, w3, b3, y=got, chunked=True, **kw)
    finally:
        ops.set_tuning("mlp_chunk_variant", 0)
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())


def test_chunk_streamed_mixer_refuses_what_it_does_not_cover(dev):
    from pytorch_connectomics_amd import hip_ops as ops
    t = torch.zeros(1, 64, 32, device=dev, dtype=torch.bfloat16)
    ab = torch.zeros(1, 2, 32, device=dev)
    z = torch.zeros(64, device=dev)
    w = torch.zeros(64, 32, device=dev)
    with pytest.raises(RuntimeError, match="no chunk-streamed kernel"):
        ops.pw_mlp(t, ab, ops.pw_pack_weight_paired(w), z, ops.pw_pack_weight_paired(w.t().contiguous(), f16=True), z[:32], N=1,
                   rows_per_sample=64, c_in=32, c_hid=64, c_out=32, chunked=True)
    t = torch.zeros(1, 64, 128, device=dev, dtype=torch.bfloat16)
    ab = torch.zeros(1, 2, 128, device=dev)
    w2 = torch.zeros(256, 128, device=dev)
    with pytest.raises(RuntimeError, match="fp16"):                       # bf16 projection image
        ops.pw_mlp(t, ab, ops.pw_pack_weight_paired(w2), torch.zeros(256, device=dev), ops.pw_pack_weight_paired(w2.t().contiguous()),
                   torch.zeros(128, device=dev), N=1, rows_per_sample=64, c_in=128, c_hid=256, c_out=128, chunked=True)


@pytest.mark.parametrize("N,rows,chid,add", [(1, 1, 64, True), (2, 63, 96, False), (3, 1000, 128, True), (2, 4097, 96, True)])
def test_mixer_with_a_projection_in_its_epilogue(dev, N, rows, chid, add):
    """pytc_pw_mlp_proj_fwd (round 5): the 32-channel mixer with a 32 -> 32 conv of its output in the epilogue (merged task heads' input
    projection): y (store_y) carries the bits of the plain mixer; z = bf16(W bf16(y) + b) equals the conv launched on y up to where the
    bias enters the fp32 sum (accumulators start from it here): at most one bf16 ulp, on a small fraction of the outputs."""
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    bf = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(rows + chid)
    t = torch.randn(N, rows, 32, generator=g).to(bf).to(dev)
    res = torch.randn(N, rows, 32, generator=g).to(bf).to(dev) if add else None
    w2n = torch.stack([ops.pw_pack_weight_paired((torch.randn(chid, 32, generator=g) / 32 ** 0.5).to(dev)) for _ in range(N)])
    b2n = torch.randn(N, chid, generator=g).to(dev)
    w3 = ops.pw_pack_weight_paired((torch.randn(32, chid, generator=g) / chid ** 0.5).to(dev), f16=True)
    b3 = torch.randn(32, generator=g).to(dev)
    wp = (torch.randn(32, 32, generator=g) / 32 ** 0.5).to(dev)
    bp = torch.randn(32, generator=g).to(dev)
    wpp = ops.pw_pack_weight_paired(wp)
    kw = dict(N=N, rows_per_sample=rows, c_in=32, c_hid=chid, c_out=32)
    assert ops.pw_mlp_proj_supported(32, chid, 32, 32) and not ops.pw_mlp_proj_supported(64, 128, 64, 32)
    y_ref = ops.pw_mlp(t, None, w2n, b2n, w3, b3, res=res, res_mode=nat.RES_ADD if add else nat.RES_NONE, **kw)
    z_ref = ops.pw_conv(y_ref, wpp, bp, N=N, rows_per_sample=rows, c_in=32, c_out=32, out_dtype=bf, w_paired=True)
    y, z = ops.pw_mlp_proj(t, w2n, b2n, w3, b3, wpp, bp, res=res, store_y=True, **kw)
    none, z2 = ops.pw_mlp_proj(t, w2n, b2n, w3, b3, wpp, bp, res=res, store_y=False, **kw)
    assert none is None and torch.equal(y, y_ref) and torch.equal(z, z2)
    zf, rf = z.float(), z_ref.float()
    diff = (zf - rf).abs()
    assert float((diff > 0).float().mean()) < 0.02, float((diff > 0).float().mean())
    assert bool((diff <= rf.abs() * 2.0 ** -7 + 1e-30).all()), float((diff / rf.abs().clamp_min(1e-6)).max())
    # and against fp64 math on the rounded y
    want = (y_ref.double() @ wp.to(bf).double().t() + bp.double()).float()
    torch.testing.assert_close(zf, want, rtol=2.0 ** -7, atol=1e-3)


def test_lds_resident_mixer_refuses_what_it_does_not_cover(dev):
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    t = torch.zeros(1, 64, 32, device=dev, dtype=torch.bfloat16)
    ab = torch.zeros(1, 2, 32, device=dev)
    z = torch.zeros(64, device=dev)
    w = torch.zeros(64, 32, device=dev)
    with pytest.raises(RuntimeError, match="no LDS-resident kernel"):
        ops.pw_mlp(t, ab, ops.pw_pack_weight_paired(w), z, ops.pw_pack_weight_paired(w.t().contiguous(), f16=True), z[:32], N=1,
                   rows_per_sample=64, c_in=32, c_hid=64, c_out=32, lds=True)
    t = torch.zeros(1, 64, 64, device=dev, dtype=torch.bfloat16)
    ab = torch.zeros(1, 2, 64, device=dev)
    w2 = torch.zeros(128, 64, device=dev)
    with pytest.raises(RuntimeError, match="must be fp16"):
        ops.pw_mlp(t, ab, ops.pw_pack_weight_paired(w2), torch.zeros(128, device=dev), ops.pw_pack_weight_paired(w2.t().contiguous()),
                   z, N=1, rows_per_sample=64, c_in=64, c_hid=128, c_out=64, lds=True)


def test_packed_fp16_gelu_accuracy(dev):
    """gelu_h2 (csrc/pytc_common.h) through the fused mixer: identity-like first GEMM, one-hot projection, so the output is
    bf16(gelu_h2(x)) for a dense sweep of x.  Error budget: polynomial fit 1e-4 + fp16 evaluation, then the bf16 output
    rounding -> |y - gelu(x)| <= 1.2e-3 + 2^-8 |gelu(x)|; the result is more accurate than the bf16 hidden activation it
    replaces wherever |gelu| > 0.1."""
    from pytorch_connectomics_amd import hip_ops as ops
    C = 32
    xs = torch.linspace(-8, 8, 64 * 512).view(1, -1, 1)
    t = (xs * torch.ones(1, 1, C)).to(torch.bfloat16)                 # every channel carries x
    eye = torch.eye(C)
    ab = torch.stack([torch.ones(1, C), torch.zeros(1, C)], 1).contiguous().to(dev)
    z = torch.zeros(C, device=dev)
    y = ops.pw_mlp(t.to(dev), ab, ops.pw_pack_weight_paired(eye.to(dev)), z, ops.pw_pack_weight_paired(eye.to(dev), f16=True), z,
                   N=1, rows_per_sample=t.shape[1], c_in=C, c_hid=C, c_out=C).float().cpu()
    want = F.gelu(t.float())
    err = (y - want).abs()
    assert float((err - (1.2e-3 + 2.0 ** -8 * want.abs())).max()) <= 0, float(err.max())
    assert float(err.mean()) < 1e-3
    assert torch.equal(y[..., 0], y[..., C - 1])                        # all channels evaluate the same function


def test_gelu_accuracy(dev):
    """The in-kernel erf GELU (A&S 7.1.26) against torch's exact erf GELU over the useful range."""
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    x = torch.linspace(-9, 9, 16 * 4096).view(1, -1, 16)
    eye = torch.eye(16)
    y = ops.pw_conv(x.to(dev), ops.pw_pack_weight(eye.to(dev), torch.float32), None, N=1, rows_per_sample=x.shape[1],
                    c_in=16, c_out=16, out_dtype=torch.float32, act=nat.ACT_GELU).cpu()
    torch.testing.assert_close(y, F.gelu(x), rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("ci,co,mode", [(32, 64, "ab"), (64, 32, "gelu_add"), (32, 128, "gelu_bwd"), (128, 64, "plain"),
                                        (256, 128, "ab"), (512, 1024, "plain"), (1024, 512, "gelu_add"), (128, 32, "up")])
def test_pw_conv_paired_row_kernel_matches_generic(ci, co, mode):
    """bf16 1x1 conv on the paired-row weight image (16-byte stores) against the generic kernel, all fused modes."""
    from pytorch_connectomics_amd import _native as nat, hip_ops as ops
    torch.manual_seed(ci + co)
    N, grid = 2, (6, 8, 10)
    rows = grid[0] * grid[1] * grid[2]
    x = torch.randn(N, rows, ci).bfloat16().cuda()
    w = (torch.randn(co, ci) / ci ** 0.5).cuda()
    b = torch.randn(co).cuda()
    kw = {}
    if mode == "ab":
        kw["ab"] = torch.stack([torch.rand(N, ci) + 0.5, torch.randn(N, ci)], 1).contiguous().cuda()
    if mode == "gelu_add":
        kw.update(pre_act=nat.ACT_GELU, res=torch.randn(N, rows, co).bfloat16().cuda(), res_mode=nat.RES_ADD)
    if mode == "gelu_bwd":
        kw.update(res=torch.randn(N, rows, co).bfloat16().cuda(), res_mode=nat.RES_GELU_BWD)
    if mode == "up":
        low = tuple(g // 2 for g in grid)
        kw.update(res=torch.randn(N, rows, co).bfloat16().cuda(), res_mode=nat.RES_UPSAMPLE, grid=grid,
                  res_low=torch.randn(N, low[0] * low[1] * low[2], co).bfloat16().cuda(), res_bias=torch.randn(co).cuda())
    assert ops.pw_conv_paired_supported(c_in=ci, c_out=co, in_dtype=torch.bfloat16, out_dtype=torch.bfloat16)
    common = dict(N=N, rows_per_sample=rows, c_in=ci, c_out=co, out_dtype=torch.bfloat16, **kw)
    y0 = ops.pw_conv(x, ops.pw_pack_weight(w, torch.bfloat16), b, **common)
    y1 = ops.pw_conv(x, ops.pw_pack_weight_paired(w), b, w_paired=True, **common)
    d = (y0.float() - y1.float()).abs()
    assert float(d.max()) <= 2 ** -6 * max(1.0, float(y0.float().abs().max()))      # one bf16 ulp at most
    # the paired-row kernel applies the sigmoid-form GELU (|err| <= 2.5e-5) where the generic one uses erf: a few more
    # outputs sit on the other side of a bf16 rounding boundary
    assert float((d > 0).float().mean()) < (0.06 if mode == "gelu_add" else 0.02)
    assert not ops.pw_conv_paired_supported(c_in=24, c_out=64, in_dtype=torch.bfloat16, out_dtype=torch.bfloat16)
    assert not ops.pw_conv_paired_supported(c_in=32, c_out=64, in_dtype=torch.float32, out_dtype=torch.float32,
                                            w_dtype=torch.float32)


def test_pw_conv_paired_row_kernel_strided_gather():
    """Down-block residual conv (1x1, stride 2): paired-row kernel with the gather against the generic kernel."""
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(11)
    N, grid, ci, co = 2, (6, 9, 10), 32, 64
    og = tuple((g - 1) // 2 + 1 for g in grid)
    rows = og[0] * og[1] * og[2]
    x = torch.randn(N, grid[0] * grid[1] * grid[2], ci).bfloat16().cuda()
    w, b = (torch.randn(co, ci) / ci ** 0.5).cuda(), torch.randn(co).cuda()
    kw = dict(N=N, rows_per_sample=rows, c_in=ci, c_out=co, out_dtype=torch.bfloat16, gather=2, grid=grid)
    assert ops.pw_conv_paired_supported(c_in=ci, c_out=co, in_dtype=torch.bfloat16, out_dtype=torch.bfloat16, gather=2)
    y0 = ops.pw_conv(x, ops.pw_pack_weight(w, torch.bfloat16), b, **kw)
    y1 = ops.pw_conv(x, ops.pw_pack_weight_paired(w), b, w_paired=True, **kw)
    ref = x.float().view(N, *grid, ci)[:, ::2, ::2, ::2].reshape(N, rows, ci) @ w.bfloat16().float().t() + b
    torch.testing.assert_close(y1.float(), ref.bfloat16().float(), rtol=2e-2, atol=2e-2)
    assert float((y0.float() - y1.float()).abs().max()) <= 2 ** -6 * max(1.0, float(y0.float().abs().max()))


@pytest.mark.parametrize("kind,in_dt,out_dt", [("stem", torch.float32, torch.bfloat16), ("stem", torch.bfloat16, torch.bfloat16),
                                               ("head", torch.bfloat16, torch.float32), ("head", torch.bfloat16, torch.bfloat16)])
def test_pw_conv_thin_kernels(kind, in_dt, out_dt):
    """C_in == 1 (stem) / C_out == 1 (head) streaming kernels against the MFMA kernel they bypass (`pw_thin` knob)."""
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(5)
    N, rows = 2, 4099
    ci, co = (1, 32) if kind == "stem" else (32, 1)
    x = torch.randn(N, rows, ci).to(in_dt).cuda()
    w, b = torch.randn(co, ci).cuda(), torch.randn(co).cuda()
    wp = ops.pw_pack_weight(w, torch.bfloat16)
    kw = dict(N=N, rows_per_sample=rows, c_in=ci, c_out=co, out_dtype=out_dt)
    y1 = ops.pw_conv(x, wp, b, **kw)
    ops.set_tuning("pw_thin", 0)
    try:
        y0 = ops.pw_conv(x, wp, b, **kw)
    finally:
        ops.set_tuning("pw_thin", 1)
    if kind == "stem":
        assert torch.equal(y0, y1)
    else:
        torch.testing.assert_close(y1.float(), y0.float(), rtol=1e-2 if out_dt == torch.bfloat16 else 1e-5, atol=1e-2 if out_dt == torch.bfloat16 else 1e-5)
    ref = x.float().bfloat16().float() @ w.bfloat16().float().t() + b
    torch.testing.assert_close(y1.float(), ref if out_dt == torch.float32 else ref.bfloat16().float(), rtol=2e-2, atol=2e-2)


def test_dwconv_march_packed_f16_error_budget(dev):
    """The z-march depthwise conv accumulates the nine in-plane taps of a z step in packed f16 and the z direction in fp32
    (`dwconv_march_h16`, default on for bf16 storage).  Budget, against an fp64 convolution of the same bf16 operands: the mean
    error stays within 15 % of what rounding the EXACT result to bf16 costs on its own, the worst element within 1.5x the worst
    element of the fp32-tap form (knob off), which is reproduced within that noise.  Large activations are clamped into the f16
    range at staging instead of becoming inf."""
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops

    def knob(v):
        nat.check(nat.lib().pytc_set_tuning(b"dwconv_march_h16", int(v)), "set_tuning")

    ops.set_tuning("dwconv_mfma", 0)                   # the VALU forms (gradient entries; forward when the matrix-core form is off)
    torch.manual_seed(11)
    N, D, C = 1, 24, 64
    x = (torch.randn(N, D, D, D, C, device=dev) * 2).to(torch.bfloat16)
    taps = torch.randn(27, C, device=dev) * 0.3
    b = torch.randn(C, device=dev) * 0.1
    w64 = taps.t().reshape(C, 1, 3, 3, 3).double().cpu()
    ref = F.conv3d(x.double().cpu().permute(0, 4, 1, 2, 3), w64, b.double().cpu(), padding=1, groups=C).permute(0, 2, 3, 4, 1)
    floor = (ref.float().to(torch.bfloat16).double() - ref).abs().mean()
    out = {}
    try:
        for h in (0, 1):
            knob(h)
            y, st = ops.dwconv3d(x, taps, b, K=3)
            out[h] = y.double().cpu()
            err = (out[h] - ref).abs()
            assert float(err.mean()) < (1.02 if h == 0 else 1.15) * float(floor), (h, float(err.mean()), float(floor))
            out[("max", h)] = float(err.max())
            s = st.sum(1).cpu().double()                                  # statistics of the stored tensor, either way
            torch.testing.assert_close(s[:, 0], out[h].sum((1, 2, 3)), rtol=1e-4, atol=1e-2)
        assert float((out[1] - out[0]).abs().mean()) < 0.6 * float(floor)
        assert out[("max", 1)] <= 1.5 * out[("max", 0)]
        knob(1)
        big = x.clone()
        big[0, 5, 5, 5, :] = 3.0e5                                       # beyond f16: saturates at the largest f16, finite everywhere
        big[0, 9, 9, 9, :] = -3.0e5
        yb, _ = ops.dwconv3d(big, taps, b, K=3)
        assert bool(torch.isfinite(yb.float()).all())
        # the result is the convolution of the input clamped to +-6e4 at staging (one v_med3_f32 per element since round 4)
        sat = big.double().cpu().clamp(-60000.0, 60000.0)
        refb = F.conv3d(sat.permute(0, 4, 1, 2, 3), w64, b.double().cpu(), padding=1, groups=C).permute(0, 2, 3, 4, 1)
        errb = (yb.double().cpu() - refb).abs()
        assert float((errb / (refb.abs() + 1.0)).max()) < 2e-2, float((errb / (refb.abs() + 1.0)).max())
    finally:
        knob(1)
        ops.set_tuning("dwconv_mfma", 1)


@pytest.mark.parametrize("N,shape,C", [(2, (9, 20, 31), 32), (1, (16, 16, 32), 64), (3, (7, 5, 6), 32), (1, (30, 18, 17), 64), (1, (44, 32, 32), 32)])
def test_dwconv_stride2_march_is_bit_identical_to_the_gather_kernel(dev, N, shape, C):
    """The down blocks' stride-2 depthwise conv at C = 32 / 64 (round 4, csrc/dwconv_s2_kernels.hip: z-march over an LDS ring of input
    planes, taps in registers) against the gather kernel (knob dwconv_s2_march = 0): fp32 FMAs in the same (kz, ky, kx) order, hence equal
    BITS; statistics equal to summation order; odd and even extents, ragged tiles, several z chunks, batch-invariant results."""
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(C + shape[1])
    x = torch.randn(N, *shape, C, device=dev).to(torch.bfloat16)
    taps = torch.randn(27, C, device=dev) * 0.3
    b = torch.randn(C, device=dev) * 0.2
    assert nat.lib().pytc_dwconv3d_kernel_variant(N, *shape, C, 3, 2, nat.BF16, 0) == 8
    try:
        y1, st1 = ops.dwconv3d(x, taps, b, K=3, stride=2)
        ops.set_tuning("dwconv_s2_march", 0)
        assert nat.lib().pytc_dwconv3d_kernel_variant(N, *shape, C, 3, 2, nat.BF16, 0) == 1
        y0, st0 = ops.dwconv3d(x, taps, b, K=3, stride=2)
    finally:
        ops.set_tuning("dwconv_s2_march", 1)
    assert y1.shape == y0.shape == (N, (shape[0] - 1) // 2 + 1, (shape[1] - 1) // 2 + 1, (shape[2] - 1) // 2 + 1, C)
    assert torch.equal(y1, y0), float((y1.float() - y0.float()).abs().max())
    torch.testing.assert_close(st1.sum(1), st0.sum(1), rtol=1e-5, atol=1e-3)
    yf = y1.float()
    torch.testing.assert_close(st1.sum(1)[:, 0], yf.sum((1, 2, 3)), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(st1.sum(1)[:, 1], (yf * yf).sum((1, 2, 3)), rtol=1e-4, atol=1e-2)
    if N > 1:
        y_one, st_one = ops.dwconv3d(x[1:2].contiguous(), taps, b, K=3, stride=2)
        assert torch.equal(y_one[0], y1[1]) and torch.equal(st_one[0], st1[1])


@pytest.mark.parametrize("N,shape,C", [(2, (5, 9, 11), 64), (1, (8, 8, 16), 128), (3, (3, 4, 7), 128), (1, (14, 14, 14), 64)])
def test_dwconv_transposed_tile_form_is_bit_identical_to_the_cell_kernel(dev, N, shape, C):
    """The up blocks' transposed depthwise conv at C = 64 / 128 (round 4, csrc/dwconvT_tile_kernels.hip: one tile of input cells per
    workgroup, taps in registers, no load after the first store) against the cell kernel (knob dwconvT_tile = 0): the same fp32 FMAs in
    the same order, hence equal BITS; statistics equal to summation order; ragged tiles, several samples, the statistics-only mode."""
    from pytorch_connectomics_amd import _native as nat
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(C + shape[2])
    x = torch.randn(N, *shape, C, device=dev).to(torch.bfloat16)
    taps = torch.randn(27, C, device=dev) * 0.3
    b = torch.randn(C, device=dev) * 0.2
    assert nat.lib().pytc_dwconv3d_kernel_variant(N, *shape, C, 3, 2, nat.BF16, 1) == 7
    try:
        y1, st1 = ops.dwconv3d(x, taps, b, K=3, transposed=True)
        y_none, st_only = ops.dwconv3d(x, taps, b, K=3, transposed=True, store=False)
        ops.set_tuning("dwconvT_tile", 0)
        assert nat.lib().pytc_dwconv3d_kernel_variant(N, *shape, C, 3, 2, nat.BF16, 1) == 4
        y0, st0 = ops.dwconv3d(x, taps, b, K=3, transposed=True)
    finally:
        ops.set_tuning("dwconvT_tile", 1)
    assert y1.shape == (N, 2 * shape[0], 2 * shape[1], 2 * shape[2], C) and y_none is None
    assert torch.equal(y1, y0), float((y1.float() - y0.float()).abs().max())
    torch.testing.assert_close(st1.sum(1), st0.sum(1), rtol=1e-5, atol=1e-3)
    assert torch.equal(st_only, st1)
    yf = y1.float()
    torch.testing.assert_close(st1.sum(1)[:, 0], yf.sum((1, 2, 3)), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(st1.sum(1)[:, 1], (yf * yf).sum((1, 2, 3)), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("N,shape,C", [(1, (9, 20, 31), 32), (2, (16, 32, 28), 64), (1, (30, 17, 16), 96), (1, (24, 24, 24), 128),
                                       (3, (8, 16, 16), 32), (2, (14, 14, 14), 256), (1, (9, 10, 13), 64)])
def test_dwconv_matrix_core_form(dev, N, shape, C):
    """dwconv3d, bf16, K = 3, stride 1 on v_mfma_f32_4x4x4_16b_bf16 (one channel per block of the instruction; round 4,
    csrc/dwconv_mfma_kernels.hip) against an fp64 convolution of the same bf16 operands.  hi + lo weight instructions (variant bit 0):
    fp32-weight accuracy -- the mean error is the bf16 rounding of the exact result; hi only (default): bf16 weights, what
    torch.autocast gives the reference's Conv3d -- within 1.8x of that floor (27 taps of 2^-9 relative weight error each).  Activations are used as stored: no range clamp (3e5 is
    exact), fp32 accumulation.  Planes in flight (bit 1) do not change a bit; statistics are those of the stored tensor; ragged
    footprints, several channel groups, several z chunks, batch-invariant results."""
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(C + shape[0])
    x = (torch.randn(N, *shape, C, device=dev) * 2).to(torch.bfloat16)
    taps = torch.randn(27, C, device=dev) * 0.3
    b = torch.randn(C, device=dev) * 0.1
    w64 = taps.t().reshape(C, 1, 3, 3, 3).double().cpu()

    def conv64(t, w, bias):
        return F.conv3d(t.double().cpu().permute(0, 4, 1, 2, 3), w, bias, padding=1, groups=C).permute(0, 2, 3, 4, 1)

    ref, mag = conv64(x, w64, b.double().cpu()), conv64(x.abs(), w64.abs(), b.double().cpu().abs())
    floor = float((ref.float().to(torch.bfloat16).double() - ref).abs().mean())
    spiky = x.clone()
    spiky[0, 3, 5, 5, :] = 3.0e5                                          # beyond f16, exact in bf16
    spiky[0, 4, 9, 9, :] = -3.0e5
    ref_s, mag_s = conv64(spiky, w64, b.double().cpu()), conv64(spiky.abs(), w64.abs(), b.double().cpu().abs())
    out = {}
    try:
        from pytorch_connectomics_amd import _native as nat
        assert nat.lib().pytc_dwconv3d_kernel_variant(N, *shape, C, 3, 1, nat.BF16, 0) == 6
        for v in (0, 1, 2, 3):
            ops.set_tuning("dwconv_mfma_variant", v)
            y, st = ops.dwconv3d(x, taps, b, K=3)
            out[v] = y.double().cpu()
            err = (out[v] - ref).abs()
            assert float(err.mean()) < (1.02 if v & 1 else 1.8) * floor, (v, float(err.mean()), floor)
            assert float((err / (mag + 1.0)).max()) < (2.0 ** -8 if v & 1 else 2.0 ** -7), (v, float((err / (mag + 1.0)).max()))
            s = st.sum(1).cpu().double()
            torch.testing.assert_close(s[:, 0], out[v].sum((1, 2, 3)), rtol=1e-4, atol=1e-2)
            torch.testing.assert_close(s[:, 1], (out[v] * out[v]).sum((1, 2, 3)), rtol=1e-4, atol=1e-2)
            ys, _ = ops.dwconv3d(spiky, taps, b, K=3)                     # no clamp, no overflow: error relative to sum |w| |x|
            assert bool(torch.isfinite(ys.float()).all())
            errs = (ys.double().cpu() - ref_s).abs()
            assert float((errs / (mag_s + 1.0)).max()) < (2.0 ** -8 if v & 1 else 2.0 ** -7), (v, float((errs / (mag_s + 1.0)).max()))
        assert torch.equal(out[0], out[2]) and torch.equal(out[1], out[3])
        ops.set_tuning("dwconv_mfma_variant", 0)
        if N > 1:                                                         # a sample's result and statistics do not depend on its batch
            y1, st1 = ops.dwconv3d(x[1:2].contiguous(), taps, b, K=3)
            yN, stN = ops.dwconv3d(x, taps, b, K=3)
            assert torch.equal(y1[0], yN[1]) and torch.equal(st1[0], stN[1])
        ops.set_tuning("dwconv_mfma", 0)                                  # and the VALU z-march agrees to the rounding of either
        yv, _ = ops.dwconv3d(x, taps, b, K=3)
        dv = (yv.double().cpu() - out[1]).abs()
        assert float(dv.mean()) < 0.6 * floor
    finally:
        ops.set_tuning("dwconv_mfma_variant", 0)
        ops.set_tuning("dwconv_mfma", 1)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,shape", [(64, 1, (3, 5, 4)), (8, 3, (2, 2, 3)), (32, 4, (4, 3, 2)), (16, 2, (1, 1, 1))])
def test_conv_transpose_thin_matches_torch(dev, dt, cin, cout, shape):
    """ConvTranspose3d(k 3, s 2, p 1, output_padding 1) with few output channels (one thread per output voxel, 1..8 taps each)
    against torch, and against the MFMA gather kernel it replaces for such layers."""
    from pytorch_connectomics_amd import hip_ops as ops
    torch.manual_seed(cin * 10 + cout)
    N = 2
    x = torch.randn(N, cin, *shape)
    w = torch.randn(cin, cout, 3, 3, 3) * 0.2
    b = torch.randn(cout)
    xq = x.to(dt).float()
    ref = F.conv_transpose3d(xq, w, b, stride=2, padding=1, output_padding=1)
    assert ops.convT3d_thin_supported(cin, cout) and not ops.convT3d_thin_supported(cin, 5) and not ops.convT3d_thin_supported(12, cout)
    got = ops.convT3d_thin(_cl(xq).to(dev).to(dt), w.to(dev), b.to(dev))
    assert tuple(got.shape) == (N, 2 * shape[0], 2 * shape[1], 2 * shape[2], cout) and got.dtype == dt
    torch.testing.assert_close(_cf(got.float().cpu()), ref, **(dict(rtol=1e-4, atol=1e-4) if dt == torch.float32 else dict(rtol=2e-2, atol=3e-2)))
    mf = ops.conv3d_strided(_cl(xq).to(dev).to(dt), ops.conv3d_pack_weight_direct(w.to(dev), dt, layout="convT"), c_out=cout,
                            kernel=(3, 3, 3), stride=(2, 2, 2), pad=(1, 1, 1), out_dims=tuple(2 * s for s in shape), transposed=True,
                            bias=b.to(dev))
    torch.testing.assert_close(got.float(), mf.float(), **(dict(rtol=1e-4, atol=1e-4) if dt == torch.float32 else dict(rtol=2e-2, atol=3e-2)))
