"""bench.py -- headline benchmark of the MI355X engine for the PyTorch Connectomics hot path.

Metric (BASELINE.json): voxels/s, MedNeXt-S 112^3 bf16.  A "step" is one sliding-window batch of the
Lucchi++ inference workload (configs[1]): gather `sw_batch_size`=8 windows of 112^3 from the HBM-resident
165x1024x768 volume -> MedNeXt-S forward (bf16 storage, fp32 accumulation) -> bump-weighted overlap-add
into the HBM-resident accumulators.  value = window-voxels/s of the whole job (all ranks).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: the path shards by independent volumes (the reference's volume-per-rank sharding,
training/lightning/data.py:234-266): every rank owns one volume, no data-path collective -> weak scaling.
"""
from __future__ import annotations

import argparse
import json
import os

# RCCL / cross-process device memory on this driver stack needs dmabuf IPC (already exported on the target image)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ROI = (112, 112, 112)
VOLUME = (165, 1024, 768)          # Lucchi++ test volume (tutorials/mito_lucchi++/README.md:114)
SW_BATCH = 8
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s spec


def build_model(device):
    from types import SimpleNamespace as NS
    from pytorch_connectomics_amd.models import build_model as bm
    cfg = NS(model=NS(arch=NS(type="mednext"), in_channels=1, out_channels=1,
                      mednext=NS(size="S", kernel_size=3), loss=NS(deep_supervision=False), heads=None))
    torch.manual_seed(0)
    model = bm(cfg).to(device).eval()
    model.model.compute_dtype = torch.bfloat16
    return model


def cpu_baseline(model, seconds_cap: float = 25.0):
    """Oracle (CPU restatement, kind='port') timed on the host cores on ONE window of the same workload
    (112^3 when it fits the time cap, 64^3 otherwise); same weights, fp32.  The thread count is the best of a
    short sweep on a 32^3 window (PyTorch's CPU depthwise conv does not scale to hundreds of threads)."""
    from oracle import mednext_oracle as MO
    st = {k: v.detach().float().cpu() for k, v in model.model.state_dict().items()}
    cores = os.cpu_count() or 1
    kw = dict(n_channels=32, exp_r=2, kernel_size=3, block_counts=[2] * 9)

    def run(side, reps=1):
        x = torch.rand(1, 1, side, side, side)
        best = float("inf")
        for _ in range(reps):
            t0 = time.perf_counter()
            MO.forward(st, x, **kw)
            best = min(best, time.perf_counter() - t0)
        return best

    with torch.no_grad():
        cands = sorted({t for t in (8, 16, 32, 64, cores) if t <= cores})
        timing = {}
        for t in cands:
            torch.set_num_threads(t)
            run(32)
            timing[t] = run(32, reps=2)
        threads = min(timing, key=timing.get)
        torch.set_num_threads(threads)
        t64 = run(64)
        side, dt = 64, t64
        if t64 * (112 / 64) ** 3 < seconds_cap:
            side, dt = 112, run(112)
    return {"value": side ** 3 / dt, "unit": "voxels/s", "cores": threads, "kind": "port",
            "sample": f"oracle MedNeXt-S fp32 forward of one {side}^3 window ({dt:.2f} s), torch CPU, "
                      f"{threads} threads (best of {cands}) on a {cores}-core host"}


def train_leg(dev, rank, world, args, barrier):
    """The training half of the metric: DDP (RCCL all-reduce) MedNeXt-S steps on synthetic 112^3 patches, bf16
    storage / fp32 master weights: HIP forward + HIP backward + BCE/Dice loss + grad clip + AdamW, nothing skipped.
    Reported next to `value` (which stays the sliding-window inference rate the target is quoted on)."""
    from types import SimpleNamespace as NS
    from pytorch_connectomics_amd.models import build_model as bm
    from pytorch_connectomics_amd.training.module import build_optimizer, synthetic_batches
    from pytorch_connectomics_amd.training.fused import bce_dice_loss
    from pytorch_connectomics_amd.utils.hostgc import quiesce_gc
    from pytorch_connectomics_amd.config import ConfigNode, schema_defaults
    cfg = ConfigNode(schema_defaults())
    cfg.model.arch.type, cfg.model.in_channels, cfg.model.out_channels = "mednext", 1, 1
    cfg.model.mednext.size, cfg.model.mednext.kernel_size = "S", 3
    cfg.optimization.optimizer.name, cfg.optimization.optimizer.lr = "AdamW", 1e-3
    cfg.optimization.gradient_clip_val = 1.0          # applied inside the fused AdamW kernel (no host sync)
    torch.manual_seed(0)
    model = bm(cfg).to(dev).train()
    model.model.compute_dtype = torch.bfloat16
    net = model
    if world > 1:
        from torch.nn.parallel import DistributedDataParallel as DDP
        net = DDP(model, device_ids=[dev.index], find_unused_parameters=True, gradient_as_bucket_view=True)
    opt = build_optimizer(cfg, model)
    it = synthetic_batches(args.train_batch, ROI, seed=11 + rank, device=dev)
    pool = [next(it) for _ in range(2)]                                # patches resident in HBM before timing
    steps = max(1, min(args.steps, 10))

    def tstep(i):
        b = pool[i % len(pool)]
        opt.zero_grad(set_to_none=True)
        out = net(b["image"])
        loss, _ = bce_dice_loss(out, b["label"])
        loss.backward()
        opt.step()
        return loss

    for i in range(max(1, min(args.warmup, 3))):
        tstep(i)
    quiesce_gc()          # what training/module.py:fit does after its first steps (utils/hostgc.py)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = tstep(i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    vox = world * args.train_batch * ROI[0] * ROI[1] * ROI[2] * steps
    del opt, net, model
    torch.cuda.empty_cache()
    return {"value": vox / dt, "unit": "voxels/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "batch_per_gpu": args.train_batch, "patch": list(ROI), "dtype": "bf16 activations, fp32 master weights",
            "parallelism": f"ddp{world}" if world > 1 else "single", "scaling": "weak",
            "includes": "forward + backward + fused BCE/Dice loss + grad-norm clip + AdamW step, all HIP kernels",
            "final_loss": float(loss.detach())}


def rsunet_leg(dev, args):
    """The path's second architecture, reported next to the headline (single GPU): RSUNet [16, 32, 64, 128], BatchNorm,
    anisotropic 2 x 18 x 160 x 160 patches, bf16 storage: training step (HIP forward + backward, fused loss, fused AdamW)
    and inference forward."""
    from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet
    from pytorch_connectomics_amd.training.fused import FusedAdamW, bce_dice_loss
    from pytorch_connectomics_amd.utils.hostgc import quiesce_gc
    torch.manual_seed(0)
    patch, batch = (18, 160, 160), 2
    m = RSUNet(1, 3, width=[16, 32, 64, 128], norm="batch", activation="relu").to(dev).train()
    m.compute_dtype = torch.bfloat16
    opt = FusedAdamW(m.parameters(), lr=1e-4, weight_decay=1e-2, max_grad_norm=1.0)
    x = torch.rand(batch, 1, *patch, device=dev)
    y = (torch.rand(batch, 3, *patch, device=dev) > 0.85).float()

    def tstep():
        opt.zero_grad(set_to_none=True)
        loss, _ = bce_dice_loss(m(x), y)
        loss.backward()
        opt.step()
        return loss

    steps = max(1, min(args.steps, 10))
    for _ in range(4):
        tstep()
    quiesce_gc()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tstep()
    torch.cuda.synchronize()
    dt_train = (time.perf_counter() - t0) / steps
    m.eval()
    with torch.no_grad():
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m(x)
        torch.cuda.synchronize()
        dt_inf = (time.perf_counter() - t0) / steps
    vox = batch * patch[0] * patch[1] * patch[2]
    del opt, m
    torch.cuda.empty_cache()
    return {"model": "RSUNet width [16,32,64,128], BatchNorm, relu", "batch": batch, "patch": list(patch), "dtype": "bf16 activations, fp32 master weights",
            "train_ms_per_step": dt_train * 1e3, "train_voxels_per_s": vox / dt_train, "infer_ms_per_forward": dt_inf * 1e3,
            "infer_voxels_per_s": vox / dt_inf, "final_loss": float(loss.detach())}


def pmc_traffic_bytes(label):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/r01_bench_hbm_counters.csv: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs by
    tools/profile_bench.sh; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None if the file or
    the kernel is missing -- counters cannot be collected from inside the timed process."""
    import csv
    import re
    f = ROOT / "profiles" / "r01_bench_hbm_counters.csv"
    if not f.exists():
        return None
    m = re.match(r"pw_mlp_fwd\[(\d+)->(\d+)->(\d+)\]", label)
    if m:
        key = f"pw_mlp_kernel<{int(m.group(1)) // 32}, {int(m.group(3)) // 16},"
    else:
        return None          # other kernels run at several shapes under one name: no per-shape counter average
    tot = n = 0.0
    for row in csv.DictReader(open(f)):          # the plain and the head-epilogue variant of the shape, launch-weighted
        if key in row["kernel"]:
            k = float(row.get("launches") or 1)
            tot += k * (2 * float(row["FETCH_SIZE_KB_mean"]) + float(row["WRITE_SIZE_KB_mean"])) * 1024
            n += k
    return int(tot / n) if n else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--train-batch", type=int, default=4)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path)"
    # PYTC_BENCH_SHARE_GPU=1 (test hook): ranks share the visible GPUs round-robin and talk over gloo, so the N > 1
    # control flow (barriers, MAX over ranks, DDP) can be exercised on a single-GPU box; never set by the driver
    share = os.environ.get("PYTC_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from pytorch_connectomics_amd import hip_ops as ops
    from pytorch_connectomics_amd.inference.window import EagerSlidingWindowEngine

    model = build_model(dev)
    eng = EagerSlidingWindowEngine(roi_size=ROI, sw_batch_size=SW_BATCH, overlap=0.5, mode="bump",
                                   padding_mode="constant", cval=0.0)
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    vol = torch.rand((1,) + VOLUME, device=dev, generator=g)            # resident in HBM before timing
    image_size, starts = eng.plan(VOLUME)
    (wz, wy, wx), combine = eng._axis_vectors(dev)
    value = torch.zeros((1,) + image_size, device=dev)
    weight = torch.zeros(image_size, device=dev)
    batches = [starts[i:i + SW_BATCH] for i in range(0, len(starts) - SW_BATCH + 1, SW_BATCH)]

    def step(i):
        b = batches[i % len(batches)]
        x = ops.gather_windows(vol, b, ROI, pad_mode="constant", cval=0.0)
        y = model.forward_cl(x)
        ops.blend_accumulate(y, b, value, weight, wz, wy, wx, combine=combine, floor_w=1e-5)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for i in range(args.warmup):
            step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    vox_per_step = SW_BATCH * ROI[0] * ROI[1] * ROI[2]
    value_vps = world * vox_per_step * args.steps / dt

    roofline = None
    if rank == 0 and not args.no_roofline:
        with torch.no_grad(), ops.profiled() as prof:
            for i in range(min(args.steps, 5)):
                step(i)
        summ = prof.summary()
        name, rec = max(summ.items(), key=lambda kv: kv[1]["ms"])
        per_launch_bytes = rec["bytes"] / rec["launches"]
        per_launch_s = rec["ms"] / rec["launches"] / 1e3
        achieved = per_launch_bytes / per_launch_s / 1e9
        roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic_bytes(name),
                    "launch_us": round(per_launch_s * 1e6, 1), "algorithmic_bytes": int(per_launch_bytes),
                    "share_of_step": round(rec["ms"] / sum(r["ms"] for r in summ.values()), 3),
                    "kernels_ms_per_step": {k: round(v["ms"] / min(args.steps, 5), 3) for k, v in
                                            sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:8]},
                    "kernel_ms_total_per_step": round(sum(r["ms"] for r in summ.values()) / min(args.steps, 5), 3)}
        if os.environ.get("PYTC_BENCH_VERBOSE"):
            for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
                n = min(args.steps, 5)
                print(f"  {k:34s} launches/step={v['launches'] / n:5.1f} ms/step={v['ms'] / n:7.3f} "
                      f"GB/s={v['bytes'] / max(v['ms'], 1e-9) / 1e6:8.1f}", file=sys.stderr)

    train = None
    if not args.no_train:
        train = train_leg(dev, rank, world, args, barrier)

    rsu = None
    if rank == 0 and world == 1 and not args.no_train:
        try:                       # a secondary figure must never cost the headline line
            rsu = rsunet_leg(dev, args)
        except Exception as e:     # noqa: BLE001 - reported in the JSON
            rsu = {"error": f"{type(e).__name__}: {e}"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(model)

    if rank == 0:
        out = {
            "metric": "voxels/s (train + sliding-window infer), MedNeXt-S 112^3 bf16",
            "value": value_vps, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "Lucchi++ sliding-window inference (configs[1]): MedNeXt-S k3, "
                                   "165x1024x768 volume, roi 112^3, overlap 0.5, bump blending, "
                                   "sw_batch_size 8, random-init weights; value = window-voxels/s",
                       "volume": list(VOLUME), "roi": list(ROI), "sw_batch_size": SW_BATCH,
                       "sharding": "one independent volume per rank, no collective"},
            "roofline": roofline, "cpu_baseline": cpu, "train": train, "rsunet": rsu,
        }
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
