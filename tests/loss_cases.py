"""Shared by tests/golden/make_golden.py --loss_orchestration (the REFERENCE's LossOrchestrator runs them) and
tests/test_host_training_module.py: loss-term lists over the torch-only losses with coefficients, pred / target / mask slices,
pos_weight spellings, batch masks, deep supervision."""
from types import SimpleNamespace as NS

import torch

CASES = [
    # (label, terms, deep supervision, batch mask)
    ("mse_default_is_class_balanced", [{"function": "WeightedMSELoss", "weight": 1.0, "target_slice": "0:3"}], False, False),
    ("mse_masked_ds", [{"function": "WeightedMSELoss", "weight": 0.5, "target_slice": "0:3"}], True, True),
    ("mae_numeric_pos_weight", [{"function": "WeightedMAELoss", "weight": 1.0, "target_slice": "0:3", "pos_weight": 2.5}], False, True),
    ("smoothl1_beta_auto", [{"function": "SmoothL1Loss", "weight": 2.0, "target_slice": "0:3", "pos_weight": "auto", "kwargs": {"beta": 0.5}}], True, False),
    ("bce_auto_pos_weight_masked", [{"function": "WeightedBCEWithLogitsLoss", "weight": 1.0, "target_slice": "0:3", "pos_weight": "auto"}], False, True),
    ("bce_numeric_pos_weight_slices", [{"function": "WeightedBCEWithLogitsLoss", "weight": 1.0, "pred_slice": "0:2", "target_slice": "1:3", "pos_weight": 2.5}], True, True),
    ("term_mask_replaces_balancing", [{"function": "WeightedMSELoss", "weight": 1.0, "target_slice": "0:3", "mask_slice": "3:4"}], False, True),
    ("term_mask_on_bce", [{"function": "WeightedBCEWithLogitsLoss", "weight": 1.0, "target_slice": "0:3", "mask": "3:4"}], True, False),
    ("per_channel_bce_keeps_its_own_balancing", [{"function": "PerChannelBCEWithLogitsLoss", "weight": 1.0, "target_slice": "0:3"}], False, True),
    ("torch_bce_and_mse_see_masks_through_inputs", [{"function": "BCEWithLogitsLoss", "weight": 1.0, "pred": "0:1", "target": "0:1"},
                                                   {"function": "MSELoss", "coefficient": 0.5, "pred_slice": "1:3", "target_slice": "1:3"}], True, True),
    ("term_skipped_on_the_ds_scales", [{"function": "WeightedBCEWithLogitsLoss", "weight": 1.0, "target_slice": "0:3"},
                                       {"function": "WeightedMSELoss", "weight": 2.0, "target_slice": "0:3", "apply_deep_supervision": False}], True, False),
    ("three_terms_mixed", [{"function": "WeightedBCEWithLogitsLoss", "weight": 1.0, "pred_slice": "0:1", "target_slice": "0:1"},
                           {"function": "WeightedMSELoss", "weight": 0.5, "pred_slice": "1:3", "target_slice": "1:3", "kwargs": {"tanh": True}},
                           {"function": "SmoothL1Loss", "weight": 2.0, "pred_slice": "2:3", "target_slice": "2:3", "mask_slice": "3:4"}], True, True),
]


def loss_cfg(terms, ds):
    return NS(model=NS(loss=NS(deep_supervision=ds, deep_supervision_weights=[1.0, 0.5, 0.25, 0.125, 0.0625], deep_supervision_clamp_min=-20.0,
                               deep_supervision_clamp_max=20.0, losses=terms, loss_balancing=None, fused=False),
                       primary_head=None, heads=None, out_channels=3), data=NS(label_transform=None))


def loss_tensors(index: int):
    """outputs {output, ds_1} (3 channels, logits up to +-25: beyond the clamp), labels (4 channels: binary, binary, real-valued, a
    term-mask candidate), batch mask."""
    g = torch.Generator().manual_seed(1000 + index)
    outs = {"output": torch.randn(2, 3, 8, 8, 8, generator=g) * 8, "ds_1": torch.randn(2, 3, 4, 4, 4, generator=g) * 8}
    lab = (torch.rand(2, 4, 8, 8, 8, generator=g) > 0.7).float()
    lab[:, 2] = torch.rand(2, 8, 8, 8, generator=g) * 2 - 1
    lab[:, 3] = (torch.rand(2, 8, 8, 8, generator=g) > 0.3).float()
    return outs, lab, (torch.rand(2, 1, 8, 8, 8, generator=g) > 0.3).float()
