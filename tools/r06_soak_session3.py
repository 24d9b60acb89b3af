"""Round-6 soak of what the third session made concurrent or stateful: (a) the lazy loop on 2 / 3 / 4 / 6 side streams against one stream, 20 passes
of a 5 x 5 x 5 window grid (MedNeXt-S bf16, sw 2, ragged last batch): every pass bit-equal; (b) the training step (skip mailbox, asynchronous
optimizer tables, row-major deep GEMMs): two runs of 6 steps from the same seed give the same losses and weights bit for bit."""
import os
import sys
from pathlib import Path
from types import SimpleNamespace as NS

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402


def lazy_soak(dev):
    from pytorch_connectomics_amd.inference import lazy as lz
    model = bench.build_model(dev)
    roi = (64, 64, 64)
    cfg = NS(model=NS(primary_head=None, heads=None, out_channels=1, output_size=list(roi)), system=NS(num_workers=0),
             data=NS(train=NS(do_2d=False), val=NS(do_2d=False), dataloader=NS(batch_size=1)),
             inference=NS(sliding_window=NS(window_size=list(roi), sw_batch_size=2, overlap=0.5, blending="bump", padding_mode="reflect",
                                            cval=0.0, border_mask=[], distributed_sharding=False, snap_to_edge=False, target_context=[]),
                          model=NS(head=None, select_channel=None, output_dtype=None, channel_activations=[{"channels": ":", "activation": "sigmoid"}]),
                          chunking=None, test_time_augmentation=NS(enabled=False)))
    vol = torch.rand(1, 192, 192, 192, generator=torch.Generator().manual_seed(3)).numpy()
    os.environ["PYTC_LAZY_SW_STREAMS"] = "1"
    ref = lz.lazy_predict_volume(cfg, model.forward, vol, device="cuda")
    bad = 0
    for it in range(20):
        n = ("2", "3", "4", "6")[it % 4]
        os.environ["PYTC_LAZY_SW_STREAMS"] = n
        y = lz.lazy_predict_volume(cfg, model.forward, vol, device="cuda")
        if not torch.equal(y, ref):
            bad += 1
            print(f"pass {it} ({n} streams): MISMATCH max |d| {float((y - ref).abs().max()):.3g}", flush=True)
    print(f"lazy soak: 20 passes, {bad} mismatches", flush=True)
    os.environ.pop("PYTC_LAZY_SW_STREAMS", None)
    return bad


def train_soak(dev):
    from pytorch_connectomics_amd.config import ConfigNode, schema_defaults
    from pytorch_connectomics_amd.models import build_model as bm
    from pytorch_connectomics_amd.training.fused import bce_dice_loss
    from pytorch_connectomics_amd.training.module import build_optimizer, synthetic_batches

    def run():
        cfg = ConfigNode(schema_defaults())
        cfg.model.arch.type, cfg.model.in_channels, cfg.model.out_channels = "mednext", 1, 1
        cfg.model.mednext.size, cfg.model.mednext.kernel_size = "S", 3
        cfg.optimization.optimizer.name, cfg.optimization.optimizer.lr = "AdamW", 1e-3
        cfg.optimization.gradient_clip_val = 1.0
        torch.manual_seed(0)
        model = bm(cfg).to(dev).train()
        model.model.compute_dtype = torch.bfloat16
        opt = build_optimizer(cfg, model)
        it = synthetic_batches(3, (64, 64, 64), seed=5, device=dev)
        pool = [next(it) for _ in range(2)]
        losses = []
        for i in range(6):
            b = pool[i % 2]
            opt.zero_grad(set_to_none=True)
            loss, _ = bce_dice_loss(model(b["image"]), b["label"])
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        return [float(v) for v in losses], [p.detach().clone() for p in model.parameters()]

    l0, w0 = run()
    l1, w1 = run()
    same = l0 == l1 and all(torch.equal(a, b) for a, b in zip(w0, w1))
    print(f"train soak: losses {l0[:3]} ... equal across two runs: {same}", flush=True)
    return 0 if same else 1


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    with torch.no_grad():
        bad = lazy_soak(dev)
    bad += train_soak(dev)
    sys.exit(1 if bad else 0)
