"""Two ranks sharing ONE GPU over gloo: the DDP training step of bench.py's training leg with progress marks per rank (debug aid for
the N > 1 control flow on a single-GPU box).
    PYTHONFAULTHANDLER=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/ddp_shared_probe.py"""
import faulthandler
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch.distributed as dist

faulthandler.enable(all_threads=True)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])


def mark(msg):
    torch.cuda.synchronize()
    print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)


torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from pytorch_connectomics_amd.config import ConfigNode, schema_defaults
from pytorch_connectomics_amd.models import build_model
from pytorch_connectomics_amd.training.fused import bce_dice_loss
from pytorch_connectomics_amd.training.module import build_optimizer, synthetic_batches
from torch.nn.parallel import DistributedDataParallel as DDP

cfg = ConfigNode(schema_defaults())
cfg.model.arch.type, cfg.model.in_channels, cfg.model.out_channels = "mednext", 1, 1
cfg.model.mednext.size, cfg.model.mednext.kernel_size = "S", 3
cfg.optimization.optimizer.name, cfg.optimization.optimizer.lr = "AdamW", 1e-3
torch.manual_seed(0)
model = build_model(cfg).to(dev).train()
model.model.compute_dtype = torch.bfloat16
bucket_view = os.environ.get("BUCKET_VIEW", "1") == "1"
net = DDP(model, device_ids=[0], find_unused_parameters=True, gradient_as_bucket_view=bucket_view)
opt = build_optimizer(cfg, model)
it = synthetic_batches(2, (64, 64, 64), seed=11 + rank, device=dev)
mark("built")
for i in range(3):
    b = next(it)
    opt.zero_grad(set_to_none=True)
    out = net(b["image"])
    mark(f"step {i} forward")
    loss, _ = bce_dice_loss(out, b["label"])
    mark(f"step {i} loss")
    loss.backward()
    mark(f"step {i} backward")
    opt.step()
    mark(f"step {i} optimizer, loss {float(loss):.4f}")
dist.barrier()
mark("done")
dist.destroy_process_group()
