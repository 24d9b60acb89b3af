"""Real multi-GPU checks over RCCL (`-m gpu`; SKIPPED unless at least two devices are visible): the single-volume inference
protocols and DDP training on two ranks against the single-rank result of the same code.

The gloo tests of tests/test_host_slab.py / test_host_distributed_inference.py cover the protocols with torch stand-ins for the
kernels; here the HIP kernels, device tensors in batch_isend_irecv / reduce / all_gather, the custom autograd Functions under DDP
with gradient_as_bucket_view, and HSA_ENABLE_IPC_MODE_LEGACY=0 are what is being exercised."""
import os
import socket
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2,
                                 reason="needs at least two GPUs on the node (RCCL)")]

ROI = (32, 32, 32)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(rank)
    torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))


def _model(out_channels=2):
    from pytorch_connectomics_amd.models.architectures.mednext import MedNeXt
    torch.manual_seed(0)
    m = MedNeXt(1, 32, out_channels, exp_r=2, kernel_size=3, do_res=True, do_res_up_down=True, block_counts=[1] * 9)
    return m


def _engine(swb=2):
    from pytorch_connectomics_amd.inference.window import EagerSlidingWindowEngine
    return EagerSlidingWindowEngine(roi_size=ROI, sw_batch_size=swb, overlap=0.5, mode="bump", padding_mode="constant", cval=0.0)


def _lazy_cfg(sharded):
    return NS(model=NS(primary_head=None, heads=None, out_channels=2, output_size=None),
              data=NS(dataloader=NS(batch_size=1, use_lazy_h5=True, use_lazy_zarr=False)),
              inference=NS(sliding_window=NS(window_size=list(ROI), sw_batch_size=2, overlap=0.5, blending="bump",
                                             padding_mode="constant", cval=0.0, snap_to_edge=False, target_context=None,
                                             border_mask=None, distributed_sharding=sharded, distributed_reduce_chunk_mb=4),
                           model=NS(head=None, select_channel=None, output_dtype=None, channel_activations=None, crop_pad=None),
                           test_time_augmentation=NS(enabled=False, distributed_sharding=False)))


def _tta_cfg(sharded):
    return NS(model=NS(primary_head=None, heads=None, out_channels=2),
              data=NS(train=NS(do_2d=False), val=NS(do_2d=False), dataloader=NS(batch_size=1), label_transform=None),
              inference=NS(sliding_window=NS(window_size=list(ROI), sw_batch_size=2, overlap=0.5, blending="bump",
                                             padding_mode="constant", cval=0.0, keep_input_on_cpu=False, sw_device=None,
                                             output_device=None, border_mask=None, distributed_sharding=False),
                           model=NS(head=None, select_channel=None, output_dtype=None,
                                    channel_activations=[{"channels": ":", "activation": "sigmoid"}], crop_pad=None),
                           test_time_augmentation=NS(enabled=True, flip_axes="all", rotation90_axes=None, rotate90_k=None,
                                                     ensemble_mode="mean", patch_first_local=True, distributed_sharding=sharded,
                                                     distributed_reduce_chunk_mb=4, apply_mask=True)))


def _inference_worker(rank, world, port, tmp):
    _init(rank, world, port)
    dev = torch.device("cuda", rank)
    from pytorch_connectomics_amd.inference.lazy import lazy_predict_volume
    from pytorch_connectomics_amd.inference.slab import slab_extent, slab_predict_volume
    from pytorch_connectomics_amd.inference.tta import TTAPredictor
    from pytorch_connectomics_amd.inference.window import build_sliding_inferer
    model = _model().to(dev).eval()
    model.compute_dtype = torch.float32
    shape = (48, 80, 64)
    vol = torch.rand((1,) + shape, generator=torch.Generator().manual_seed(5)).to(dev)
    eng = _engine()
    with torch.no_grad():
        solo = eng(vol.unsqueeze(0), model)[0]                                   # single-rank engine on this GPU
        # 1. slab ownership + p2p halo bands: gathered slabs == single-rank volume (sums arrive in another order: fp32 rounding)
        full = slab_predict_volume(vol, eng, model, gather=True)
        assert tuple(full.shape) == tuple(solo.shape)
        torch.testing.assert_close(full, solo, rtol=2e-5, atol=2e-5)
        # ... and with every rank holding only its own planes
        ax, (lo, hi) = slab_extent(shape, eng, world, rank)
        part = slab_predict_volume(vol.narrow(ax + 1, lo, hi - lo).contiguous(), eng, model, gather=True, full_size=shape)
        assert torch.equal(part, full)
        # 2. window sharding [rank::world] + in-place reduce of the HBM accumulators onto rank 0
        host = vol.cpu().numpy()
        sharded = lazy_predict_volume(_lazy_cfg(True), model.forward, host, device=dev)
        single = lazy_predict_volume(_lazy_cfg(False), model.forward, host, device=dev)
        if rank == 0:
            torch.testing.assert_close(sharded, single, rtol=2e-5, atol=2e-5)
        else:
            assert sharded.numel() == 0
        # 3. TTA views [rank::world], one reduce of the ensemble
        out = {}
        for sharded_views in (True, False):
            cfg = _tta_cfg(sharded_views)
            pred = TTAPredictor(cfg=cfg, sliding_inferer=build_sliding_inferer(cfg), forward_fn=model.forward, model=model)
            out[sharded_views] = pred.predict(vol.unsqueeze(0).clone())
        if rank == 0:
            torch.testing.assert_close(out[True], out[False], rtol=2e-5, atol=2e-5)
        else:
            assert out[True].numel() == 0
    torch.distributed.barrier()
    if rank == 0:
        open(os.path.join(tmp, "inference_ok"), "w").write("ok")
    torch.distributed.destroy_process_group()


def test_two_rank_inference_protocols_over_rccl(tmp_path):
    mp.spawn(_inference_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "inference_ok").exists()


def _ddp_worker(rank, world, port, tmp):
    _init(rank, world, port)
    dev = torch.device("cuda", rank)
    from torch.nn.parallel import DistributedDataParallel as DDP
    from pytorch_connectomics_amd.training.fused import FusedAdamW, bce_dice_loss
    g = torch.Generator().manual_seed(9)
    xs = torch.rand(3, 2, 1, 32, 32, 32, generator=g)                             # 3 steps x global batch 2
    ys = (torch.rand(3, 2, 2, 32, 32, 32, generator=g) > 0.8).float()

    def run(ddp):
        m = _model().to(dev).train()
        m.compute_dtype = torch.bfloat16
        net = DDP(m, device_ids=[rank], find_unused_parameters=True, gradient_as_bucket_view=True) if ddp else m
        opt = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0)
        losses = []
        for i in range(3):
            x, y = (xs[i, rank:rank + 1], ys[i, rank:rank + 1]) if ddp else (xs[i], ys[i])
            opt.zero_grad(set_to_none=True)
            loss, _ = bce_dice_loss(net(x.to(dev)), y.to(dev))
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        return m, losses

    m_ddp, _ = run(True)
    # every rank holds the same weights after 3 all-reduced steps
    flat = torch.cat([p.detach().flatten() for p in m_ddp.parameters()])
    other = [torch.empty_like(flat) for _ in range(world)]
    torch.distributed.all_gather(other, flat)
    assert torch.equal(other[0], other[1])
    if rank == 0:
        # ... and they follow the single-process run on the whole batch (DDP averages the per-rank mean losses; the fused loss
        # is a mean over the batch, so the gradients agree up to bf16 accumulation order)
        m_one, _ = run(False)
        ref = torch.cat([p.detach().flatten() for p in m_one.parameters()])
        rel = float((flat - ref).norm() / ref.norm())
        assert rel < 5e-3, rel
        open(os.path.join(tmp, "ddp_ok"), "w").write(f"{rel:.3e}")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_ddp_training_over_rccl(tmp_path):
    mp.spawn(_ddp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ddp_ok").exists()
