"""Entry point mirroring the reference's scripts/main.py + runtime/cli.py + runtime/dispatch.py for the hot path:

    python -m pytorch_connectomics_amd.main --config X.yaml [--mode train|test] [--checkpoint C.ckpt]
                                            [--fast-dev-run] [key.sub=value ...]

test : build_model(cfg) -> load checkpoint -> volume to HBM -> InferenceManager.predict_with_tta (or chunked
       inference when inference.chunking.enabled) -> optional binary Jaccard -> <save_path>/results/*_prediction.h5 (the CZYX raw-prediction artifact)
train: training/module.py's Lightning-free harness (HIP forward + backward, fused loss, fused clip + AdamW), one process
       per GPU under torch.distributed.run (DDP over RCCL); writes checkpoints/last.ckpt in the Lightning layout.
Volumes: .npy / .npz (first array) / random://<name>[?shape=Z,Y,X] / .h5 (dataset `main`; h5py or the in-repo libhdf5 shim).
"""
from __future__ import annotations

import argparse
import json
from types import SimpleNamespace as NS
import logging
import os
import sys
import time
from pathlib import Path
from typing import Optional, Sequence

# RCCL / cross-process device memory on this driver stack needs dmabuf IPC (already exported on the target image)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

from .config import load_config

logger = logging.getLogger("pytorch_connectomics_amd")


def parse_args(argv: Optional[Sequence[str]] = None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="PyTorch Connectomics hot path on MI355X")
    p.add_argument("--config", required=True, help="YAML config (reference tutorial format)")
    p.add_argument("--mode", default="train", choices=["train", "test", "tune", "tune-test"])
    p.add_argument("--checkpoint", default=None, help="Lightning .ckpt / state-dict file")
    p.add_argument("--fast-dev-run", nargs="?", const=1, default=0, type=int)
    p.add_argument("--demo", action="store_true", help="use random:// volumes when no data is configured")
    p.add_argument("overrides", nargs="*", help="key.sub=value config overrides")
    return p.parse_args(argv)


def read_volume(spec: str, *, default_shape=(64, 128, 128), seed: int = 0) -> np.ndarray:
    if spec.startswith("random://"):
        shape = default_shape
        if "?shape=" in spec:
            shape = tuple(int(v) for v in spec.split("?shape=", 1)[1].split(","))
        return np.random.default_rng(seed).random(shape, dtype=np.float32)
    path = Path(spec)
    if path.suffix == ".npy":
        return np.load(path, mmap_mode="r")
    if path.suffix == ".npz":
        z = np.load(path)
        return z[z.files[0]]
    if path.suffix in (".h5", ".hdf5"):
        from .utils.h5lite import get_h5_backend
        be = get_h5_backend()
        if be is None:
            raise RuntimeError(f"{spec}: reading HDF5 needs h5py or the in-repo libpytc_h5.so (built from csrc/host/h5io.c "
                               "against libhdf5); neither is available -- convert the volume to .npy")
        with be.File(path, "r") as fh:
            return np.asarray(fh["main" if "main" in fh else list(fh.keys())[0]][...])
    if path.suffix.lower() in (".tif", ".tiff"):
        from .utils.tiffstack import read_tiff_volume
        return read_tiff_volume(str(path))
    if ".zarr" in spec:
        # zarr v2 / v3 directory store (optionally <store>.zarr/<array key>), read whole (inference/volume_source.py)
        from .inference.volume_source import open_zarr
        arr = open_zarr(spec)
        return np.asarray(arr[(slice(None),) * len(arr.shape)])
    if path.suffix.lower() == ".png":
        # a glob pattern of PNG slices (reference io.py:161-177 read_images + :346-350): sorted, stacked along z; colour slices
        # (D, H, W, C) become (C, D, H, W)
        import glob
        from PIL import Image
        files = sorted(glob.glob(spec))
        if not files:
            raise ValueError(f"No files found matching: {spec}")
        data = np.stack([np.asarray(Image.open(f)) for f in files], axis=0)
        return data.transpose(3, 0, 1, 2) if data.ndim == 4 else data
    if path.name.lower().endswith((".nii", ".nii.gz")):
        from .utils.niftilite import read_nifti
        return read_nifti(str(path))
    raise ValueError(f"unsupported volume format: {spec}")


def _load_checkpoint_file(path: str):
    """torch.load with weights_only=True first (tensors, containers, numbers: what this engine and plain state dicts hold);
    only a Lightning checkpoint that pickles other objects (hyper-parameter namespaces) falls back to the full unpickler,
    with a warning -- such a file executes code on load and must come from a trusted source."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as exc:      # noqa: BLE001 - pickle.UnpicklingError and friends
        logger.warning("checkpoint %s needs the full (unsafe) unpickler (%s): load only trusted files", path, type(exc).__name__)
        return torch.load(path, map_location="cpu", weights_only=False)


def load_checkpoint(model: torch.nn.Module, path: str) -> None:
    """Lightning checkpoints store the LightningModule state (`model.<wrapper attr>...`, model.py:244-297);
    plain state dicts and `_orig_mod.` / `module.` prefixes are accepted too (model_weights.py:45-72)."""
    blob = _load_checkpoint_file(path)
    sd = blob.get("state_dict", blob) if isinstance(blob, dict) else blob
    own = set(model.state_dict().keys())
    out = {}
    for k, v in sd.items():
        for pre in ("_orig_mod.", "module."):
            if k.startswith(pre):
                k = k[len(pre):]
        if k not in own and k.startswith("model.") and k[len("model."):] in own:
            k = k[len("model."):]
        if k.startswith("loss_functions."):
            continue
        out[k] = v
    missing, unexpected = model.load_state_dict(out, strict=False)
    if missing:
        raise RuntimeError(f"checkpoint {path} is missing keys: {missing[:5]}{'...' if len(missing) > 5 else ''}")
    if unexpected:
        logger.warning("checkpoint has %d unexpected keys (ignored)", len(unexpected))


def binary_jaccard(pred: torch.Tensor, label: torch.Tensor, threshold: float = 0.5) -> float:
    """TP / (TP + FP + FN) after thresholding (evaluation/metric_execution.py:166-200)."""
    p = pred > threshold
    t = label > 0
    inter = (p & t).sum().item()
    union = (p | t).sum().item()
    return float(inter) / float(union) if union else 1.0


def run_test(cfg, args) -> dict:
    from .inference import InferenceManager
    from .inference.chunked import is_chunked_inference_enabled, run_chunked_prediction_inference
    from .models import build_model
    if not torch.cuda.is_available():
        raise RuntimeError("test mode needs an MI355X (ROCm) device: this engine has no CPU path")
    dev = torch.device("cuda", 0)
    torch.manual_seed(int(cfg.system.seed))
    model = build_model(cfg).to(dev).eval()
    if args.checkpoint:
        load_checkpoint(model, args.checkpoint)
    precision = str(cfg.optimization.precision)
    if "bf16" in precision or "16" in precision:
        for mod in (model, getattr(model, "model", model)):
            if hasattr(mod, "compute_dtype"):
                mod.compute_dtype = torch.bfloat16    # fp16-mixed configs run as bf16 storage on this engine
    image_spec = cfg.data.test.image or ("random://demo" if args.demo else None)
    if image_spec is None:
        raise ValueError("data.test.image is not set (use --demo for a random volume)")
    lazy_dl = bool(getattr(cfg.data.dataloader, "use_lazy_h5", False) or getattr(cfg.data.dataloader, "use_lazy_zarr", False))
    if is_chunked_inference_enabled(cfg) and lazy_dl and not str(image_spec).startswith("random://"):
        vol = None           # disk-backed: the chunked runner reads region by region through LazyVolumeAccessor
    else:
        vol = read_volume(str(image_spec))
    out_dir = Path(cfg.save_path) / "results"
    out_dir.mkdir(parents=True, exist_ok=True)
    name = Path(str(image_spec).split("?")[0]).stem or "volume"
    # data.test.mask: multiplied into the prediction after the ensemble (TTAPredictor._apply_mask_to_result; the lazy / chunked path
    # reads it region by region), with the reference's alignment switch (test_pipeline.py:282-288)
    from .utils.volume_normalize import mask_align_to_image, prepare_test_mask
    mask_spec = getattr(cfg.data.test, "mask", None)
    align_mask = mask_align_to_image(cfg)
    t0 = time.perf_counter()
    with torch.no_grad():
        if is_chunked_inference_enabled(cfg):
            pred = run_chunked_prediction_inference(cfg, model.forward, vol if vol is not None else str(image_spec),
                                                    output_path=out_dir / f"{name}_prediction.h5",
                                                    device=dev, image_path=str(image_spec), checkpoint_path=args.checkpoint,
                                                    mask_path=str(mask_spec) if mask_spec else None, mask_align_to_image=align_mask)
            pred_t = None if pred is None else torch.from_numpy(pred).unsqueeze(0)
        else:
            # the reference's test transforms, in their order (data/augmentation/build.py:416-655); the chunked branch above does the
            # same per region / window through the accessor, like the reference's lazy reader
            from .utils.volume_normalize import prepare_test_image
            vol = prepare_test_image(vol, cfg)              # val_transpose, context border (pad_size / pad_mode), normalisation
            host = np.ascontiguousarray(vol, dtype=np.float32)
            if not host.flags.writeable:          # a read-only memory map (.npy opened with mmap): torch wants a writable buffer
                host = host.copy()
            x = torch.from_numpy(host).to(dev)
            while x.dim() < 5:
                x = x.unsqueeze(0)
            mgr = InferenceManager(cfg=cfg, model=model, forward_fn=model.forward)
            from .inference.window import is_2d_inference_mode
            if is_2d_inference_mode(cfg) and x.shape[2] > 1:
                # data.*.do_2d ("extract 2D slices from 3D volumes", config/schema/data.py:221): every z-slice is one 2-D image
                # through the 2-D mode of the predictor (no sliding window, views over y / x); results restacked along z
                slice_mgr = mgr

                def _per_slice(images, mask=None, **k):
                    outs = []
                    for z in range(images.shape[2]):
                        mz = mask
                        if isinstance(mask, torch.Tensor) and mask.dim() >= 3 and mask.shape[-3] == images.shape[2]:
                            mz = mask[..., z:z + 1, :, :]
                        outs.append(slice_mgr.predict_with_tta(images[:, :, z:z + 1], mask=mz, **k))
                    return torch.stack(outs, dim=2)
                mgr = NS(cfg=cfg, predict_with_tta=_per_slice)
            # prediction-space crop: user crop_pad + DeepEM affinity border (test_pipeline.py:734, prediction_crops.py:240-256)
            from .inference.crop import crop_spatial_by_pad, resolve_global_prediction_crop
            from .inference.stage import run_prediction_inference
            crop_pad = resolve_global_prediction_crop(cfg)
            cropping = any(lo or hi for lo, hi in crop_pad)
            art = out_dir / f"{name}_prediction.h5"
            # predict; with no crop the stage writes the artifact itself (transform + storage dtype on the device, one D2H
            # copy of the stored representation); with a crop the stage is re-entered on the cropped prediction
            mask_t = None
            if mask_spec:
                mask_t = torch.from_numpy(np.ascontiguousarray(prepare_test_mask(read_volume(str(mask_spec)), cfg), dtype=np.float32)).to(dev)
                while mask_t.dim() < 5:
                    mask_t = mask_t.unsqueeze(0)
            pred_t = run_prediction_inference(mgr, x, mask=mask_t, mask_align_to_image=align_mask, output_path=None if cropping else art,
                                              image_path=str(image_spec), checkpoint_path=args.checkpoint, input_shape=vol.shape[-3:],
                                              crop_pad=crop_pad)
            if cropping:
                pred_t = crop_spatial_by_pad(pred_t, crop_pad, item_name="prediction").contiguous()
                held = NS(cfg=cfg, predict_with_tta=lambda *_a, **_k: pred_t)
                run_prediction_inference(held, x, output_path=art, image_path=str(image_spec), checkpoint_path=args.checkpoint,
                                         input_shape=vol.shape[-3:], crop_pad=crop_pad)
    dt = time.perf_counter() - t0
    out_vox = float(np.prod(vol.shape[-3:])) if vol is not None else (float(np.prod(pred_t.shape[-3:])) if pred_t is not None else 0.0)
    metrics = {"seconds": dt, "output_voxels_per_s": out_vox / dt}
    label_spec = cfg.data.test.label
    if label_spec and pred_t is not None:
        lab = torch.from_numpy(np.array(read_volume(str(label_spec)), copy=True))       # an own, writable array (memmaps are read-only)
        metrics["jaccard"] = binary_jaccard(pred_t[0, 0].float().cpu(), lab)
    (out_dir / f"{name}_metrics.json").write_text(json.dumps(metrics, indent=2))
    logger.info("test done: %s", metrics)
    return metrics


def run_train(cfg, args) -> dict:
    """DDP-ready training loop on synthetic (random://) patches or patches sampled from .npy volumes."""
    import os
    from .training import ConnectomicsModule, fit, synthetic_batches
    if not torch.cuda.is_available():
        raise RuntimeError("train mode needs an MI355X (ROCm) device: this engine has no CPU path")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    torch.manual_seed(int(cfg.system.seed))                   # identical initial weights on every rank
    module = ConnectomicsModule(cfg)
    if args.checkpoint:
        module.load_checkpoint_dict(_load_checkpoint_file(args.checkpoint))
    patch = tuple(cfg.data.dataloader.patch_size or cfg.model.input_size or (64, 64, 64))
    bs = int(cfg.data.dataloader.batch_size)
    from .training.module import resolve_training_steps
    steps, per_epoch = resolve_training_steps(cfg, fast_dev_run=int(args.fast_dev_run or 0))
    if args.fast_dev_run and args.checkpoint:
        steps += int(module.global_step)          # --fast-dev-run N on a resumed run = N MORE steps (Lightning runs N batches)
    img_spec = cfg.data.train.image
    if img_spec is None or str(img_spec).startswith("random://") or args.demo:
        batches = synthetic_batches(bs, patch, in_channels=cfg.model.in_channels, out_channels=cfg.model.out_channels,
                                    seed=int(cfg.system.seed) + rank, device=dev)
    else:
        from .utils.volume_normalize import normalize_image_for_config
        vol = torch.from_numpy(np.ascontiguousarray(normalize_image_for_config(read_volume(str(img_spec)), cfg), dtype=np.float32))
        lab = torch.from_numpy(np.ascontiguousarray(read_volume(str(cfg.data.train.label))).astype(np.float32))
        g = torch.Generator().manual_seed(int(cfg.system.seed) + rank)

        def sampler():
            while True:
                xs, ys = [], []
                for _ in range(bs):
                    # a 2-D patch_size on a 3-D volume = one z-slice per sample (data.*.do_2d semantics)
                    full = ((1,) * (vol.dim() - len(patch))) + tuple(patch)
                    o = [int(torch.randint(0, vol.shape[a] - full[a] + 1, (1,), generator=g)) for a in range(vol.dim())]
                    sl = tuple(slice(o[a], o[a] + full[a]) for a in range(vol.dim()))
                    xv, yv = vol[sl].reshape(patch), lab[sl].reshape(patch)
                    xs.append(xv[None]); ys.append((yv > 0).float()[None])
                yield {"image": torch.stack(xs), "label": torch.stack(ys)}
        batches = sampler()
    t0 = time.perf_counter()
    history, opt = fit(module, batches, max_steps=steps, device=dev, ddp=world > 1, log=logger.info if rank == 0 else None,
                       steps_per_epoch=per_epoch)
    if not history:
        raise RuntimeError(f"nothing to train: the checkpoint is already at step {module.global_step} of {steps}")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"steps": len(history), "global_step": int(module.global_step), "first_loss": history[0], "last_loss": history[-1],
           "voxels_per_s": world * len(history) * bs * float(np.prod(patch)) / dt}
    if rank == 0:
        ck_dir = Path(cfg.save_path) / "checkpoints"
        ck_dir.mkdir(parents=True, exist_ok=True)
        torch.save(module.checkpoint_dict(opt), ck_dir / "last.ckpt")
        logger.info("train done: %s", out)
    return out


def main(argv: Optional[Sequence[str]] = None):
    args = parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    cfg = load_config(args.config, mode=args.mode, overrides=args.overrides)
    if args.mode in ("test", "tune-test"):
        return run_test(cfg, args)
    if args.mode == "train":
        return run_train(cfg, args)
    raise NotImplementedError(f"--mode {args.mode}: hyper-parameter tuning is outside the hot path of this engine")


if __name__ == "__main__":
    main(sys.argv[1:])
