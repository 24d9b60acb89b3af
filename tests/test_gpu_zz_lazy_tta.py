"""GPU parity of lazy sliding-window inference with per-window test-time augmentation and a mask volume against the REFERENCE's
lazy loop (tests/golden/lazy_tta.npz, make_golden.py --lazy_tta): the real gather / activation / ensemble / blend kernels under the
orchestration that tests/test_host_lazy_tta.py checks on the CPU with stand-ins.  (File name: collected last on purpose -- the newest
device path of the round runs after everything else.)"""
import numpy as np
import pytest
import torch

from lazy_tta_cases import LAZY_TTA_CASES, lazy_tta_cfg

pytestmark = pytest.mark.gpu


def _net_lazy(x):
    ramp = torch.linspace(0, 1, x.shape[-1], device=x.device).view(1, 1, 1, 1, -1)
    return torch.cat([2 * x - 1 + ramp, 0.5 * x + x.mean(dim=(2, 3, 4), keepdim=True)], 1)


@pytest.mark.parametrize("name", list(LAZY_TTA_CASES))
def test_lazy_tta_and_mask_match_the_reference_loop_on_device(name, golden_dir):
    from pytorch_connectomics_amd.inference.lazy import lazy_predict_region, lazy_predict_volume
    g = np.load(golden_dir / "lazy_tta.npz")
    case = LAZY_TTA_CASES[name]
    cfg = lazy_tta_cfg(**case["cfg"])
    kw = dict(mask_path=g["mask"] if case.get("mask") else None, device="cuda")
    if case.get("region") is None:
        y = lazy_predict_volume(cfg, _net_lazy, g["vol"], **kw)
    else:
        y = lazy_predict_region(cfg, _net_lazy, g["vol"], region_start=case["region"][0], region_stop=case["region"][1], **kw)
    want = g[f"{name}__y"]
    assert y.is_cuda and tuple(y.shape) == want.shape
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=2e-5, atol=2e-5)


def test_chunked_runner_passes_the_mask_and_views_on(tmp_path, golden_dir):
    """run_chunked_prediction_inference with the reference's keywords (image_path=, mask_path=): chunk by chunk == whole volume."""
    from types import SimpleNamespace as NS
    from pytorch_connectomics_amd.inference.chunked import run_chunked_prediction_inference
    from pytorch_connectomics_amd.inference.lazy import lazy_predict_volume
    g = np.load(golden_dir / "lazy_tta.npz")
    cfg = lazy_tta_cfg(roi=(8, 12, 16), flips=[[1]], acts=[{"channels": "0", "activation": "sigmoid"}], padding_mode="constant")
    cfg.inference.chunking = NS(enabled=True, chunk_size=[9, 16, 20], halo=[2, 3, 4], axes="all", shard_id=None, num_shards=None)
    full = lazy_predict_volume(cfg, _net_lazy, g["vol"], mask_path=g["mask"], device="cuda")
    seen = []
    out = run_chunked_prediction_inference(cfg, _net_lazy, image_path=g["vol"], mask_path=g["mask"], output_path=tmp_path / "p.npy",
                                           device="cuda", qc_streaming_callback=NS(update=lambda a, z_offset, z_axis: seen.append((a.shape, z_offset, z_axis))))
    np.testing.assert_array_equal(out, full[0].cpu().numpy())
    assert seen == [(out.shape, 0, 1)]


_LR = ["1-0-0", "0-1-0", "0-0-1", "3-0-0", "0-3-0", "0-0-3"]
WHOLE_AFF_CASES = {
    "whole_aff6_flip8_mean_deepem": ("all", None, "mean", 6, _LR, "deepem", None, "x"),
    "whole_aff3_rot_min_banis": ([[0]], [[1, 2]], "min", 3, ["1-0-0", "0-1-0", "0-0-1"], "banis", None, "x_square"),
    "whole_aff6_select_max": ([[1], [2], [1, 2]], None, "max", 6, _LR, "deepem", [3, 0, 4], "x"),
}


@pytest.mark.parametrize("name", list(WHOLE_AFF_CASES))
def test_whole_volume_affinity_tta_matches_reference(name, golden_dir):
    """`patch_first_local: false` with directional-affinity outputs (refused before this round's second session): every view is a
    whole-volume sliding pass, the inverse view re-anchors the affinity channels, the ensemble counts per-voxel validity --
    against the reference's InferenceManager.predict_with_tta (tests/golden/tta_affinity_whole.npz)."""
    from types import SimpleNamespace as NS
    from pytorch_connectomics_amd.inference import InferenceManager
    from test_gpu_tta import _cfg, _net_aff
    flip, rot, mode, n_out, offsets, amode, select, xkey = WHOLE_AFF_CASES[name]
    g = np.load(golden_dir / "tta_affinity_whole.npz")
    tta_ns = NS(enabled=True, flip_axes=flip, rotation90_axes=rot, rotate90_k=None, ensemble_mode=mode, patch_first_local=False,
                distributed_sharding=False, apply_mask=True)
    cfg = _cfg(tta_ns, [{"channels": ":", "activation": "sigmoid"}], select)
    cfg.model.out_channels = n_out
    cfg.data.label_transform = NS(stack_outputs=True, targets=[{"name": "affinity", "kwargs": {"offsets": offsets, "affinity_mode": amode}}])
    mgr = InferenceManager(cfg=cfg, model=torch.nn.Identity(), forward_fn=lambda t: _net_aff(t, n_out))
    y = mgr.predict_with_tta(torch.from_numpy(g[xkey]).cuda()).cpu().numpy()
    want = g[f"{name}__y"]
    assert y.shape == want.shape
    np.testing.assert_allclose(y, want, rtol=2e-5, atol=2e-5)
