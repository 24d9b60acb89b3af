"""The sliding engine + TTA predictor ORCHESTRATION on CPU: the fixtures of tests/test_gpu_tta.py (outputs of the reference's
InferenceManager / TTAPredictor, tests/golden/make_golden.py --tta / --tta_affinity) replayed through this package's real engine and
predictor code with the device kernels replaced by the torch stand-ins of tests/test_host_lazy_tta.py (view-coded gather / blend,
affinity channel maps, per-shift weights).  What the GPU tests pin on the HIP kernels, this pins on the host logic around them --
the window plan, view codes, channel maps, partial-channel normalisation, ensemble bookkeeping -- in the `-m "not gpu"` suite."""
import sys
from pathlib import Path
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import test_gpu_tta as G          # noqa: E402  (case tables and closed-form networks only; its tests stay GPU-marked)
from test_host_lazy_tta import _Ops  # noqa: E402


@pytest.fixture()
def standin_engine(monkeypatch):
    import pytorch_connectomics_amd.inference.tta as tta
    import pytorch_connectomics_amd.inference.tta_ensemble as ens
    import pytorch_connectomics_amd.inference.window as window
    for mod in (tta, ens, window):
        monkeypatch.setattr(mod, "ops", _Ops)
    monkeypatch.setattr(window.EagerSlidingWindowEngine, "_check_inputs", lambda self, inputs: torch.device("cpu"))
    monkeypatch.setattr(window.EagerSlidingWindowEngine, "_lanes", lambda self, dev, n, network=None: [])
    from pytorch_connectomics_amd.inference import InferenceManager
    return InferenceManager


@pytest.mark.parametrize("name", list(G.CASES))
def test_patch_first_tta_orchestration_matches_reference(name, golden_dir, standin_engine):
    g = np.load(golden_dir / "tta.npz")
    tta_ns, acts, select = G.CASES[name]
    x = torch.from_numpy(g["x_square"] if name.startswith("rot") else g["x"])
    mgr = standin_engine(cfg=G._cfg(tta_ns, acts, select), model=torch.nn.Identity(), forward_fn=G._net_asym)
    np.testing.assert_allclose(mgr.predict_with_tta(x).numpy(), g[f"{name}__y"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("name", list(G.WHOLE))
def test_whole_volume_tta_orchestration_matches_reference(name, golden_dir, standin_engine):
    g = np.load(golden_dir / "tta.npz")
    tta_ns, acts, select = G.WHOLE[name]
    cfg = G._cfg(tta_ns, acts, select)
    cfg.inference.sliding_window.window_size = [8, 12, 16]
    mgr = standin_engine(cfg=cfg, model=torch.nn.Identity(), forward_fn=G._net_asym)
    np.testing.assert_allclose(mgr.predict_with_tta(torch.from_numpy(g["x"])).numpy(), g[f"{name}__y"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("name", list(G.AFF_CASES))
def test_affinity_tta_orchestration_matches_reference(name, golden_dir, standin_engine):
    flip, rot, mode, n_out, offsets, amode, select, xkey = G.AFF_CASES[name]
    g = np.load(golden_dir / "tta_affinity.npz")
    tta_ns = NS(enabled=True, flip_axes=flip, rotation90_axes=rot, rotate90_k=None, ensemble_mode=mode, patch_first_local=True,
                distributed_sharding=False, apply_mask=True)
    cfg = G._cfg(tta_ns, [{"channels": ":", "activation": "sigmoid"}], select)
    cfg.model.out_channels = n_out
    cfg.data.label_transform = NS(stack_outputs=True, targets=[{"name": "affinity", "kwargs": {"offsets": offsets, "affinity_mode": amode}}])
    mgr = standin_engine(cfg=cfg, model=torch.nn.Identity(), forward_fn=lambda t: G._net_aff(t, n_out))
    np.testing.assert_allclose(mgr.predict_with_tta(torch.from_numpy(g[xkey])).numpy(), g[f"{name}__y"], rtol=2e-5, atol=2e-5)


def test_predictor_without_a_sliding_engine_calls_the_network_on_whole_views(standin_engine):
    """`sliding_inferer=None`: every view goes to the network whole (reference `_run_network` without an inferer, tta.py:415-433), so an
    odd quarter turn of unequal axes is legal and a network may change the spatial shape -- then the mask check speaks
    (reference tests/unit/test_inference_tta_masking.py:143-174)."""
    from pytorch_connectomics_amd.inference.tta import TTAPredictor
    tta_ns = NS(enabled=True, flip_axes=None, rotation90_axes=[[1, 2]], rotate90_k=[0, 1], ensemble_mode="mean", patch_first_local=False,
                distributed_sharding=False, apply_mask=True)
    cfg = G._cfg(tta_ns, None, None)
    cfg.inference.sliding_window = None
    seen = []

    def net(x):
        seen.append(tuple(x.shape[2:]))
        return torch.cat([x, 2 * x, x + 1], 1)
    x = torch.rand(1, 1, 4, 5, 7)
    y = TTAPredictor(cfg=cfg, sliding_inferer=None, forward_fn=net).predict(x)
    assert seen == [(4, 5, 7), (4, 7, 5)] and tuple(y.shape) == (1, 3, 4, 5, 7)
    torch.testing.assert_close(y, torch.cat([x, 2 * x, x + 1], 1))          # a pointwise network: every inverted view is the same map
    cfg.inference.test_time_augmentation = NS(enabled=False, apply_mask=True)
    wider = TTAPredictor(cfg=cfg, sliding_inferer=None, forward_fn=lambda t: torch.nn.functional.pad(t, (1, 1)))
    assert tuple(wider.predict(x).shape) == (1, 1, 4, 5, 9)
    with pytest.raises(ValueError, match="Mask spatial shape must exactly match"):
        wider.predict(x, mask=torch.zeros(1, 1, 4, 5, 7))
    assert torch.all(wider.predict(x, mask=torch.zeros(1, 1, 4, 5, 7), mask_align_to_image=True) == 0)


def test_a_callers_own_sliding_inferer_is_called_per_whole_view(standin_engine, golden_dir):
    """Drop-in boundary (b)-2: the predictor takes ANY `inferer(inputs=, network=)` callable.  Without patch-first-local TTA every view
    goes through it (reference `_run_network`, tta.py:415-433); patch-first-local TTA never calls it and runs the window loop the
    configuration describes (reference tests/unit/test_inference_tta_masking.py:372-410) -- both give the engine's own answer."""
    from pytorch_connectomics_amd.inference.tta import TTAPredictor
    from pytorch_connectomics_amd.inference.window import build_sliding_inferer

    class Tracking:
        def __init__(self, inferer):
            self.inferer, self.calls = inferer, 0

        def __call__(self, *args, **kwargs):
            self.calls += 1
            return self.inferer(*args, **kwargs)
    g = np.load(golden_dir / "tta.npz")
    x = torch.from_numpy(g["x"])
    tta_ns, acts, select = G.CASES["flipz_select"]
    results = {}
    for patch_first in (False, True):
        cfg = G._cfg(NS(**{**vars(tta_ns), "patch_first_local": patch_first}), acts, select)
        own = TTAPredictor(cfg=cfg, sliding_inferer=build_sliding_inferer(cfg), forward_fn=G._net_asym).predict(x)
        wrapped = Tracking(build_sliding_inferer(cfg))
        got = TTAPredictor(cfg=cfg, sliding_inferer=wrapped, forward_fn=G._net_asym).predict(x)
        assert wrapped.calls == (0 if patch_first else 3)                     # identity, [0], [1, 2]
        torch.testing.assert_close(got, own, rtol=1e-6, atol=1e-6)
        results[patch_first] = got
    # (the two modes differ for this position-dependent network; the reference fixture is the patch-first one)
    np.testing.assert_allclose(results[True].numpy(), g["flipz_select__y"], rtol=2e-5, atol=2e-5)
