"""Shared by tests/golden/make_golden.py --lazy_tta (which runs the REFERENCE's lazy loop on them) and the tests that read
tests/golden/lazy_tta.npz: lazy sliding-window inference with per-window test-time augmentation and / or a mask volume."""
from types import SimpleNamespace as NS

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
