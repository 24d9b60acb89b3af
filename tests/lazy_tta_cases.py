"""Shared by tests/golden/make_golden.py --lazy_tta (which runs the REFERENCE's lazy loop on them) and the tests that read
tests/golden/lazy_tta.npz: lazy sliding-window inference with per-window test-time augmentation and / or a mask volume."""
from types import SimpleNamespace as NS

import torch

SIG_TANH = [{"channels": "0", "activation": "sigmoid"}, {"channels": "1", "activation": "tanh"}]


def lazy_tta_cfg(*, roi, flips=None, rot=None, rot_k=None, mode="mean", acts=None, select=None, overlap=0.5, blending="bump",
                 padding_mode="reflect", swb=3, tta=True, apply_mask=True):
    return NS(
        model=NS(primary_head=None, heads=None, out_channels=2, output_size=list(roi)),
        system=NS(num_workers=0),
        data=NS(train=NS(do_2d=False), val=NS(do_2d=False), dataloader=NS(batch_size=1, use_lazy_zarr=False, use_lazy_h5=False),
                label_transform=None),
        inference=NS(
            sliding_window=NS(window_size=list(roi), sw_batch_size=swb, overlap=overlap, blending=blending, padding_mode=padding_mode,
                              cval=0.0, keep_input_on_cpu=False, sw_device=None, output_device=None, border_mask=[],
                              distributed_sharding=False, snap_to_edge=False, target_context=[], distributed_reduce_chunk_mb=128),
            model=NS(head=None, select_channel=select, output_dtype=None, channel_activations=acts, crop_pad=None),
            test_time_augmentation=NS(enabled=tta, distributed_sharding=False, flip_axes=flips, rotation90_axes=rot, rotate90_k=rot_k,
                                      ensemble_mode=mode, patch_first_local=True, apply_mask=apply_mask, empty_cache_interval=0)))


LAZY_TTA_CASES = {
    "flips8_mean": dict(cfg=dict(roi=(8, 12, 16), flips="all", acts=SIG_TANH)),
    "flips_min": dict(cfg=dict(roi=(8, 12, 16), flips=[[0], [1, 2]], mode="min", acts=SIG_TANH)),
    "flips_mixed_modes_select": dict(cfg=dict(roi=(8, 12, 16), flips=[[2], [0, 1]], mode=[["0", "max"], ["1", "mean"]], acts=SIG_TANH)),
    "rot_yx_mean": dict(cfg=dict(roi=(8, 12, 12), flips=[[0]], rot=[[1, 2]], acts=SIG_TANH)),
    "mask_only": dict(cfg=dict(roi=(8, 12, 16), tta=False, acts=SIG_TANH), mask=True),
    "mask_not_applied": dict(cfg=dict(roi=(8, 12, 16), tta=False, acts=SIG_TANH, apply_mask=False), mask=True),
    "region_flips_mask": dict(cfg=dict(roi=(8, 12, 16), flips=[[1]], acts=SIG_TANH, padding_mode="constant"), mask=True,
                              region=((3, 5, 7), (17, 22, 30))),
    "select_channel_tta": dict(cfg=dict(roi=(8, 8, 8), flips=[[0, 2]], select=[1], acts=SIG_TANH, blending="constant", overlap=0.25)),
}


# ---- mask application (tests/golden/mask_application.npz, make_golden.py --masks)
def mask_cases():
    """Inputs of the mask-application fixture (shared with tests/test_host_tta_utils.py): (label, prediction shape, mask builder,
    align, activation types, apply_mask)."""
    g = torch.Generator().manual_seed(23)
    def rnd(*shape):
        return torch.rand(*shape, generator=g)
    return [
        ("binary_same_shape", (1, 2, 4, 5, 6), lambda: (rnd(1, 1, 4, 5, 6) > 0.4).float(), False, None, True),
        ("uint8_255", (1, 2, 4, 5, 6), lambda: ((rnd(1, 1, 4, 5, 6) > 0.4) * 255).to(torch.uint8), False, None, True),
        ("real_valued_and_negative", (1, 2, 4, 5, 6), lambda: rnd(1, 1, 4, 5, 6) - 0.5, False, None, True),
        ("per_channel", (2, 3, 4, 5, 6), lambda: (rnd(2, 3, 4, 5, 6) > 0.5).float(), False, None, True),
        ("no_channel_axis", (2, 3, 4, 5, 6), lambda: (rnd(2, 4, 5, 6) > 0.5).float(), False, None, True),
        ("spatial_only_broadcast_batch", (2, 3, 4, 5, 6), lambda: (rnd(4, 5, 6) > 0.5).float(), False, None, True),
        ("numpy_nested_singletons", (1, 2, 4, 5, 6), lambda: [[(rnd(1, 4, 5, 6) > 0.5).float().numpy()]], False, None, True),
        ("list_of_two_stacks", (2, 1, 4, 5, 6), lambda: [(rnd(1, 4, 5, 6) > 0.5).float(), (rnd(1, 4, 5, 6) > 0.5).float()], False, None, True),
        ("tanh_channel_fills_minus_one", (1, 3, 4, 5, 6), lambda: (rnd(1, 1, 4, 5, 6) > 0.5).float(), False, ["sigmoid", "tanh", None], True),
        ("types_of_other_width_ignored", (1, 3, 4, 5, 6), lambda: (rnd(1, 1, 4, 5, 6) > 0.5).float(), False, ["tanh", "tanh"], True),
        ("align_crop_and_pad", (1, 2, 4, 6, 5), lambda: (rnd(1, 1, 7, 3, 5) > 0.3).float(), True, None, True),
        ("depth1_mask_for_2d_prediction", (2, 2, 5, 6), lambda: (rnd(2, 1, 1, 5, 6) > 0.5).float(), False, None, True),
        ("apply_mask_off", (1, 2, 4, 5, 6), lambda: torch.zeros(1, 1, 4, 5, 6), False, None, False),
        ("error_shape", (1, 2, 4, 5, 6), lambda: torch.ones(1, 1, 4, 5, 7), False, None, True),
        ("error_channels", (1, 3, 4, 5, 6), lambda: torch.ones(1, 2, 4, 5, 6), False, None, True),
        ("error_batch", (3, 2, 4, 5, 6), lambda: torch.ones(2, 1, 4, 5, 6), False, None, True),
        ("error_rank", (1, 2, 4, 5, 6), lambda: torch.ones(5, 6), False, None, True),
        ("error_ragged_list", (2, 1, 4, 5, 6), lambda: [torch.ones(1, 4, 5, 6), torch.ones(1, 4, 5, 5)], False, None, True),
        ("unsupported_payload_is_skipped", (1, 2, 4, 5, 6), lambda: "not a mask", False, None, True),
    ]
