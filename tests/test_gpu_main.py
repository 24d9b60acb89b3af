"""GPU end-to-end: the CLI test mode (config -> build_model -> checkpoint -> TTA sliding-window -> artifact)."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

YAML = """
experiment_name: e2e
save_path: {save}
default:
  optimization: {{precision: "32"}}
  model:
    arch: {{type: mednext_custom}}
    in_channels: 1
    out_channels: 1
    mednext: {{base_channels: 8, exp_r: 2, kernel_size: 3, block_counts: [1,1,1,1,1,1,1,1,1]}}
  inference:
    window: {{window_size: [32, 32, 32], overlap: 0.5, blending: bump, sw_batch_size: 4}}
    model: {{channel_activations: [{{channels: ":", activation: sigmoid}}]}}
    test_time_augmentation: {{enabled: true, flip_axes: [[2]]}}
  data:
    image_transform: {{normalize: none}}
test:
  data:
    test: {{image: "{img}", label: "{lab}"}}
"""


def test_cli_test_mode_end_to_end(tmp_path):
    from pytorch_connectomics_amd.config import load_config
    from pytorch_connectomics_amd.main import main
    from pytorch_connectomics_amd.models import build_model
    rng = np.random.default_rng(0)
    img = rng.random((40, 44, 48), dtype=np.float32)
    np.save(tmp_path / "img.npy", img)
    np.save(tmp_path / "lab.npy", (rng.random((40, 44, 48)) > 0.5).astype(np.uint8))
    cfg_path = tmp_path / "cfg.yaml"
    cfg_path.write_text(YAML.format(save=tmp_path / "out", img=tmp_path / "img.npy", lab=tmp_path / "lab.npy"))
    # a Lightning-style checkpoint: LightningModule state dict = "model." + wrapper keys
    torch.manual_seed(3)
    ref_model = build_model(load_config(cfg_path, mode="test"))
    ckpt = {"state_dict": {"model." + k: v for k, v in ref_model.state_dict().items()}, "pytc_metadata": {}}
    torch.save(ckpt, tmp_path / "last.ckpt")
    metrics = main(["--config", str(cfg_path), "--mode", "test", "--checkpoint", str(tmp_path / "last.ckpt")])
    from pytorch_connectomics_amd.inference.artifact import read_prediction_artifact
    pred = read_prediction_artifact(tmp_path / "out" / "results" / "img_prediction.h5")
    assert pred.dtype == np.float32 and pred.shape == (1, 40, 44, 48) and 0.0 <= pred.min() and pred.max() <= 1.0
    assert 0.0 <= metrics["jaccard"] <= 1.0
    assert json.loads((tmp_path / "out" / "results" / "img_metrics.json").read_text())["jaccard"] == metrics["jaccard"]
    # the checkpoint really was loaded: same weights run directly give the same answer
    from pytorch_connectomics_amd.inference import InferenceManager
    cfg = load_config(cfg_path, mode="test")
    m = ref_model.cuda().eval()
    direct = InferenceManager(cfg=cfg, model=m, forward_fn=m.forward).predict_with_tta(torch.from_numpy(img).cuda())
    np.testing.assert_allclose(direct[0].cpu().numpy(), pred, atol=1e-6)


def test_cli_writes_uint8_artifact_with_metadata(tmp_path):
    """prediction_transform (x255 -> uint8 on the device) + the CZYX artifact with its attribute vocabulary."""
    from pytorch_connectomics_amd.inference.artifact import read_prediction_artifact
    from pytorch_connectomics_amd.main import main
    rng = np.random.default_rng(1)
    np.save(tmp_path / "img.npy", rng.random((34, 36, 40), dtype=np.float32))
    cfg_path = tmp_path / "cfg.yaml"
    cfg_path.write_text(YAML.format(save=tmp_path / "out", img=tmp_path / "img.npy", lab="").replace(
        "    test_time_augmentation:", "    prediction_transform: {enabled: true, intensity_scale: 255.0, intensity_dtype: uint8}\n"
        "    test_time_augmentation:").replace(', label: ""', ""))
    main(["--config", str(cfg_path), "--mode", "test"])
    arr, attrs = read_prediction_artifact(tmp_path / "out" / "results" / "img_prediction.h5", return_metadata=True)
    assert arr.dtype == np.uint8 and arr.shape == (1, 34, 36, 40) and arr.max() > 0
    assert attrs["layout"] == "CZYX" and attrs["intensity_dtype"] == "uint8" and attrs["intensity_scale"] == 255.0
    assert json.loads(attrs["final_shape"]) == [34, 36, 40] and attrs["model_architecture"] == "mednext_custom"
    assert json.loads(attrs["input_shape"]) == [34, 36, 40] and attrs["compression"] == "gzip"


def test_cli_chunked_hdf5_in_and_out(tmp_path):
    """An HDF5 test volume (dataset `main`) through inference.chunking -> chunk_*.h5 + index.json + the stitched CZYX HDF5
    artifact; equals the whole-volume prediction of the same model (sigmoid before blending in both, lazy global grid)."""
    from pytorch_connectomics_amd.inference.artifact import read_prediction_artifact
    from pytorch_connectomics_amd.main import main
    from pytorch_connectomics_amd.utils import h5lite
    be = h5lite.get_h5_backend()
    if be is None:
        pytest.skip("no HDF5 backend on this box")
    rng = np.random.default_rng(5)
    img = rng.random((40, 44, 48), dtype=np.float32)
    with be.File(tmp_path / "img.h5", "w") as fh:
        fh.create_dataset("main", data=img, compression="gzip")
    cfg_path = tmp_path / "cfg.yaml"
    cfg_path.write_text(YAML.format(save=tmp_path / "out", img=tmp_path / "img.h5", lab="").replace(', label: ""', "").replace(
        "    test_time_augmentation: {enabled: true, flip_axes: [[2]]}",
        "    test_time_augmentation: {enabled: false}\n    chunking: {enabled: true, chunk_size: [24, 44, 24], halo: [16, 0, 16]}"))
    main(["--config", str(cfg_path), "--mode", "test"])
    res = tmp_path / "out" / "results"
    arr, attrs = read_prediction_artifact(res / "img_prediction.h5", return_metadata=True)
    assert arr.shape == (1, 40, 44, 48) and 0.0 <= arr.min() and arr.max() <= 1.0
    assert json.loads(attrs["chunk_shape"]) == [24, 44, 24] and json.loads(attrs["halo"]) == [16, 0, 16]
    assert len(list((res / "img_prediction.h5.chunks").glob("chunk_*.h5"))) == 4
    idx = json.loads((res / "img_prediction.h5.index.json").read_text())
    assert [c["key"] for c in idx["chunks"]] == ["z0_y0_x0", "z0_y0_x1", "z1_y0_x0", "z1_y0_x1"]
    # whole-volume lazy prediction with the same seed-built model
    from pytorch_connectomics_amd.config import load_config
    from pytorch_connectomics_amd.inference.lazy import lazy_predict_volume
    from pytorch_connectomics_amd.models import build_model
    cfg = load_config(cfg_path, mode="test")
    torch.manual_seed(int(cfg.system.seed))
    m = build_model(cfg).cuda().eval()
    full = lazy_predict_volume(cfg, m.forward, img, device="cuda")
    np.testing.assert_array_equal(full[0].cpu().numpy(), arr)


def test_cli_minimal_monai_unet_train_then_test(tmp_path):
    """BASELINE configs[0] (the reference's tutorials/minimal.yaml: MONAI U-Net filters [16,32,64], one residual unit,
    32x64x64 patches of random:// data, Dice on channel 0, fp32, 1 epoch x 1 step) through --mode train then --mode test
    with the written checkpoint: the strided / transposed-conv HIP kernels, PReLU + instance norm, the generic loss path."""
    from pytorch_connectomics_amd.inference.artifact import read_prediction_artifact
    from pytorch_connectomics_amd.main import main
    cfg = tmp_path / "minimal.yaml"
    cfg.write_text(f"""
experiment_name: minimal_demo
save_path: {tmp_path / 'out'}
default:
  optimization: {{precision: "32"}}
  model:
    arch: {{type: monai_unet}}
    in_channels: 1
    out_channels: 1
    input_size: [32, 64, 64]
    output_size: [32, 64, 64]
    monai: {{filters: [16, 32, 64], num_res_units: 1, kernel_size: 3, dropout: 0.0}}
    loss:
      losses:
        - {{function: DiceLoss, weight: 1.0, pred_slice: "0:1", target_slice: "0:1"}}
  data:
    train: {{image: "random://minimal/train_image", label: "random://minimal/train_label"}}
    dataloader: {{batch_size: 1, patch_size: [32, 64, 64]}}
    image_transform: {{normalize: none}}
  inference:
    window: {{window_size: [32, 64, 64], overlap: 0.5, sw_batch_size: 2}}
    model: {{channel_activations: [{{channels: ":", activation: sigmoid}}]}}
train:
  optimization:
    max_epochs: 1
    n_steps_per_epoch: 1
    precision: "32"
    optimizer: {{name: AdamW, lr: 1.0e-4}}
  system: {{seed: 42}}
test:
  data:
    test: {{image: "random://minimal/test_image?shape=40,96,80"}}
""")
    out = main(["--config", str(cfg), "--mode", "train"])
    # the reference's minimal.yaml gives DiceLoss no kwargs: MONAI's default sigmoid=False, i.e. Dice on the raw logits
    assert out["steps"] == 1 and np.isfinite(out["first_loss"])
    ck = tmp_path / "out" / "checkpoints" / "last.ckpt"
    blob = torch.load(ck, weights_only=True)
    assert blob["global_step"] == 1 and any(k.startswith("model.model.") for k in blob["state_dict"])
    m = main(["--config", str(cfg), "--mode", "test", "--checkpoint", str(ck)])
    assert m["output_voxels_per_s"] > 0
    pred = read_prediction_artifact(next((tmp_path / "out" / "results").glob("*_prediction.h5")))
    assert pred.shape == (1, 40, 96, 80) and 0.0 <= pred.min() and pred.max() <= 1.0 and pred.std() > 0
    # the trained weights were really used: the checkpoint's model through the manager gives the artifact
    from pytorch_connectomics_amd.config import load_config
    from pytorch_connectomics_amd.inference import InferenceManager
    from pytorch_connectomics_amd.main import load_checkpoint, read_volume
    from pytorch_connectomics_amd.models import build_model
    c = load_config(cfg, mode="test")
    net = build_model(c).cuda().eval()
    load_checkpoint(net, str(ck))
    x = torch.from_numpy(np.ascontiguousarray(read_volume("random://minimal/test_image?shape=40,96,80"), dtype=np.float32)).cuda()
    direct = InferenceManager(cfg=c, model=net, forward_fn=net.forward).predict_with_tta(x)
    np.testing.assert_allclose(direct[0].cpu().numpy(), pred, atol=1e-6)


def test_cli_mednext_2d_train_then_slicewise_test(tmp_path):
    """A `mednext.dim: 2d` configuration shaped like the reference's tutorials/mito_mitolab.yaml (deep supervision, Dice + two BCE
    terms + tanh-MSE on channel slices, 2-D patches): --mode train on random 2-D patches, then --mode test with data.*.do_2d,
    where every z-slice of the test volume is one 2-D image (no sliding window) and the results are restacked."""
    from pytorch_connectomics_amd.inference.artifact import read_prediction_artifact
    from pytorch_connectomics_amd.main import main
    cfg = tmp_path / "m2d.yaml"
    cfg.write_text(f"""
experiment_name: mednext2d
save_path: {tmp_path / 'out'}
default:
  optimization: {{precision: "32"}}
  model:
    arch: {{type: mednext_custom}}
    in_channels: 1
    out_channels: 3
    mednext: {{base_channels: 8, exp_r: 2, kernel_size: 3, block_counts: [1,1,1,1,1,1,1,1,1], dim: 2d}}
    loss:
      deep_supervision: true
      losses:
        - {{function: DiceLoss, weight: 1.0, kwargs: {{include_background: false, sigmoid: true, smooth_nr: 1.0e-5, smooth_dr: 1.0e-5}}, pred_slice: "0:1", target_slice: "0:1"}}
        - {{function: BCEWithLogitsLoss, weight: 0.5, pred_slice: "0:1", target_slice: "0:1"}}
        - {{function: BCEWithLogitsLoss, weight: 0.5, pred_slice: "1:2", target_slice: "1:2"}}
        - {{function: WeightedMSELoss, weight: 1.0, kwargs: {{tanh: true}}, pred_slice: "2:3", target_slice: "2:3"}}
  data:
    train: {{image: "random://m2d/train_image", label: "random://m2d/train_label", do_2d: true}}
    dataloader: {{batch_size: 2, patch_size: [64, 48]}}
    image_transform: {{normalize: none}}
  inference:
    model:
      channel_activations:
        - {{channels: "0:2", activation: sigmoid}}
        - {{channels: "2:3", activation: tanh}}
    test_time_augmentation: {{enabled: true, flip_axes: [[0], [1]]}}
train:
  optimization:
    max_epochs: 1
    n_steps_per_epoch: 2
    optimizer: {{name: AdamW, lr: 1.0e-3}}
  system: {{seed: 7}}
test:
  data:
    test: {{image: "random://m2d/test_image?shape=3,64,48"}}
""")
    out = main(["--config", str(cfg), "--mode", "train"])
    assert out["steps"] == 2 and np.isfinite(out["first_loss"]) and np.isfinite(out["last_loss"])
    ck = tmp_path / "out" / "checkpoints" / "last.ckpt"
    blob = torch.load(ck, weights_only=True)
    assert blob["state_dict"]["model.model.enc_block_0.0.conv1.weight"].dim() == 4              # Conv2d parameters
    m = main(["--config", str(cfg), "--mode", "test", "--checkpoint", str(ck)])
    assert m["output_voxels_per_s"] > 0
    pred = read_prediction_artifact(next((tmp_path / "out" / "results").glob("*_prediction.h5")))
    assert pred.shape == (3, 3, 64, 48)
    assert 0.0 <= pred[:2].min() and pred[:2].max() <= 1.0 and -1.0 <= pred[2].min() and pred[2].max() <= 1.0
    # slice z of the artifact = the 3-view ensemble of the 2-D model on that slice
    from pytorch_connectomics_amd.config import load_config
    from pytorch_connectomics_amd.main import load_checkpoint, read_volume
    from pytorch_connectomics_amd.models import build_model
    c = load_config(cfg, mode="test")
    net = build_model(c).cuda().eval()
    load_checkpoint(net, str(ck))
    vol = torch.from_numpy(np.ascontiguousarray(read_volume("random://m2d/test_image?shape=3,64,48"), dtype=np.float32)).cuda()

    def act(o):
        return torch.cat([torch.sigmoid(o[:, :2]), torch.tanh(o[:, 2:3])], 1)
    with torch.no_grad():
        for z in range(3):
            x = vol[z][None, None]
            out0 = net(x)
            out0 = out0["output"] if isinstance(out0, dict) else (out0[0] if isinstance(out0, list) else out0)
            views = [act(out0)]
            for ax in (2, 3):
                o = net(torch.flip(x, [ax]).contiguous())
                o = o["output"] if isinstance(o, dict) else (o[0] if isinstance(o, list) else o)
                views.append(torch.flip(act(o), [ax]))
            np.testing.assert_allclose(torch.stack(views).mean(0)[0].cpu().numpy(), pred[:, z], atol=2e-5)


def test_cli_context_border_crop_and_mask(tmp_path):
    """The SNEMI-style test flow through the CLI: `data_transform.pad_size` adds a reflected context border to the loaded image,
    `inference.model.crop_pad` removes it from the prediction again (back to the label's field of view), `data.test.mask` (binarised,
    zero border) is multiplied into the ensemble -- equal to the same steps done by hand around the InferenceManager."""
    from pytorch_connectomics_amd.config import load_config
    from pytorch_connectomics_amd.inference import InferenceManager
    from pytorch_connectomics_amd.inference.artifact import read_prediction_artifact
    from pytorch_connectomics_amd.main import main
    from pytorch_connectomics_amd.models import build_model
    rng = np.random.default_rng(11)
    img = (rng.random((20, 24, 28)) * 255).astype(np.uint8)
    mask = (rng.random((20, 24, 28)) > 0.3).astype(np.uint8) * 200
    np.save(tmp_path / "img.npy", img)
    np.save(tmp_path / "mask.npy", mask)
    cfg_path = tmp_path / "cfg.yaml"
    cfg_path.write_text(f"""
experiment_name: border
save_path: {tmp_path / 'out'}
default:
  optimization: {{precision: "32"}}
  model:
    arch: {{type: mednext_custom}}
    in_channels: 1
    out_channels: 1
    mednext: {{base_channels: 8, exp_r: 2, kernel_size: 3, block_counts: [1,1,1,1,1,1,1,1,1]}}
  data:
    data_transform: {{pad_size: [4, 8, 8], pad_mode: reflect}}
    image_transform: {{normalize: divide-255}}
    mask_transform: {{binarize: true, threshold: 0.0}}
  inference:
    window: {{window_size: [16, 16, 16], overlap: 0.5, blending: bump, sw_batch_size: 4}}
    model: {{channel_activations: [{{channels: ":", activation: sigmoid}}], crop_pad: [4, 4, 8, 8, 8, 8]}}
    test_time_augmentation: {{enabled: true, flip_axes: [[1]]}}
test:
  data:
    test: {{image: "{tmp_path / 'img.npy'}", mask: "{tmp_path / 'mask.npy'}"}}
""")
    main(["--config", str(cfg_path), "--mode", "test"])
    pred = read_prediction_artifact(tmp_path / "out" / "results" / "img_prediction.h5")
    assert pred.shape == (1, 20, 24, 28)
    assert np.all(pred[0][mask == 0] == 0.0) and pred[0][mask > 0].min() > 0.0
    cfg = load_config(cfg_path, mode="test")
    torch.manual_seed(int(cfg.system.seed))
    net = build_model(cfg).cuda().eval()
    padded = np.pad(img, [(4, 4), (8, 8), (8, 8)], mode="reflect") / 255.0
    padded_mask = np.pad((mask > 0).astype(np.float32), [(4, 4), (8, 8), (8, 8)], mode="constant")
    direct = InferenceManager(cfg=cfg, model=net, forward_fn=net.forward).predict_with_tta(
        torch.from_numpy(padded.astype(np.float32)).cuda()[None, None], mask=torch.from_numpy(padded_mask).cuda()[None, None])
    np.testing.assert_allclose(direct[0, :, 4:-4, 8:-8, 8:-8].cpu().numpy(), pred, atol=1e-6)
