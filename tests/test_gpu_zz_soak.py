"""Soak test of the two-stream window pipeline.

Round 3 found one kernel (the K = 3 x-block depthwise conv: compiler-generated packed-fp32 FMAs with a high-dword select) whose
results changed when an MFMA kernel of another HIP stream shared the GPU; the fix is a build rule (csrc/build.py: no SLP
vectorisation in the files that contained such forms + an ISA grep) whose mechanism was never pinned down, and two streams are the
DEFAULT and the benched path.  The short guard in test_gpu_window.py runs 4 passes on a reduced volume with MedNeXt-S only; this file is the
long form: every architecture of the path, at its real window, whole volumes, 15 passes on one stream and 15 each on two and three, every pass compared bit
for bit with the first one-stream pass -- and the training step with its weight-gradient side stream off / on.
Runs last (file name) and takes about a minute of GPU time."""
from types import SimpleNamespace as NS

import pytest
import torch

pytestmark = pytest.mark.gpu

PASSES = 30


def _engine(roi, swb):
    from pytorch_connectomics_amd.inference.window import EagerSlidingWindowEngine
    return EagerSlidingWindowEngine(roi_size=roi, sw_batch_size=swb, overlap=0.5, mode="bump", padding_mode="constant", cval=0.0)


def _mednext(size, out_channels, heads=None, primary=None):
    from pytorch_connectomics_amd.models import build_model
    cfg = NS(model=NS(arch=NS(type="mednext"), in_channels=1, out_channels=out_channels, mednext=NS(size=size, kernel_size=3),
                      loss=NS(deep_supervision=False), heads=heads, primary_head=primary))
    torch.manual_seed(0)
    m = build_model(cfg).cuda().eval()
    m.model.compute_dtype = torch.bfloat16
    return m


def _rsunet_stock():
    from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet
    torch.manual_seed(0)
    m = RSUNet(1, 1, width=[18, 36, 48, 64, 80], norm="group", num_groups=4, activation="elu", down_factors=[(1, 2, 2)] * 4, depth_2d=1,
               kernel_2d=(1, 3, 3)).cuda().eval()
    m.compute_dtype = torch.bfloat16
    return m


CASES = {
    # Lucchi++ test volume, the headline configuration of bench.py: 468 windows, 59 batches per pass
    "mednext_s_112_lucchi": (lambda: _mednext("S", 1), (165, 1024, 768), (112, 112, 112), 8),
    # MitoEM configuration: MedNeXt-L, three heads, 160^3 windows (12 windows, 6 batches per pass)
    "mednext_l_160_three_heads": (lambda: _mednext("L", 7, heads={"aff_r1": {"out_channels": 3, "num_blocks": 1, "hidden_channels": 8},
                                                                     "aff_r5": {"out_channels": 3, "num_blocks": 1, "hidden_channels": 8},
                                                                     "sdt": {"out_channels": 1, "num_blocks": 1, "hidden_channels": 8}},
                                                   primary="aff_r1"), (160, 320, 400), (160, 160, 160), 2),
    # the reference's stock RSUNet profile on an anisotropic volume (channel-padded dense-conv path)
    "rsunet_stock_18x256": (_rsunet_stock, (36, 512, 512), (18, 256, 256), 2),
}


@pytest.mark.parametrize("name", list(CASES))
def test_two_window_streams_are_bit_identical_over_many_whole_volume_passes(name):
    make, shape, roi, swb = CASES[name]
    model = make()
    eng = _engine(roi, swb)
    vol = torch.rand((1, 1) + shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7))
    n_batches = (len(eng.plan(shape)[1]) - 1 + swb - 1) // swb
    assert n_batches >= 4, n_batches
    with torch.no_grad():
        eng.pipeline_streams = 1
        ref = eng(vol, model).clone()
        torch.cuda.synchronize()
        assert eng.last_stats["streams"] == 1
        for streams in (2, 1, 3):                       # 3 = the engine's default since round 4
            eng.pipeline_streams = streams
            for i in range(PASSES // 2 if streams == 1 else PASSES // 2 + (PASSES % 2)):
                y = eng(vol, model)
                torch.cuda.synchronize()
                assert eng.last_stats["streams"] == streams
                if not torch.equal(y, ref):
                    bad = (y != ref)
                    raise AssertionError(f"{name}: pass {i} on {streams} stream(s) differs from the first one-stream pass in "
                                         f"{int(bad.sum())} voxels (max |d| {float((y - ref).abs().max()):.3e})")
                del y


def test_training_step_is_bit_identical_with_the_weight_gradient_side_stream():
    """MedNeXt-S training step at 2 x 64^3, bf16: loss and every parameter gradient with the weight-gradient kernels on a side
    stream (SIDE_STREAM_WGRAD) equal the single-stream step bit for bit, 10 steps each."""
    from pytorch_connectomics_amd.training import autograd as AG
    from pytorch_connectomics_amd.training.fused import bce_dice_loss
    from pytorch_connectomics_amd.models import build_model
    cfg = NS(model=NS(arch=NS(type="mednext"), in_channels=1, out_channels=1, mednext=NS(size="S", kernel_size=3),
                      loss=NS(deep_supervision=False), heads=None))
    torch.manual_seed(0)
    m = build_model(cfg).cuda().train()
    m.model.compute_dtype = torch.bfloat16
    x = torch.rand(2, 1, 64, 64, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    y = (torch.rand(2, 1, 64, 64, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)) > 0.85).float()

    def step():
        m.zero_grad(set_to_none=True)
        loss, _ = bce_dice_loss(m(x), y)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), [None if p.grad is None else p.grad.detach().clone() for p in m.parameters()]

    prev = AG.SIDE_STREAM_WGRAD
    try:
        AG.SIDE_STREAM_WGRAD = False
        loss0, g0 = step()
        for side in (True, False, True):
            AG.SIDE_STREAM_WGRAD = side
            for i in range(5):
                loss, g = step()
                assert torch.equal(loss, loss0), (side, i)
                for a, b, (n, _p) in zip(g, g0, m.named_parameters()):
                    assert (a is None) == (b is None), n              # parameters outside the graph (upstream's dummy tensor)
                    assert a is None or torch.equal(a, b), f"side stream {side}, step {i}: gradient of {n} differs"
        assert sum(a is not None for a in g0) > 100
    finally:
        AG.SIDE_STREAM_WGRAD = prev
