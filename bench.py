"""bench.py -- headline benchmark of the MI355X engine for the PyTorch Connectomics hot path.

Metric (BASELINE.json): voxels/s, MedNeXt-S 112^3 bf16.  A "step" is ONE WHOLE-VOLUME pass of the product's sliding-window
engine -- `EagerSlidingWindowEngine.__call__(volume, model)` exactly as `main.py --mode test` reaches it -- over the
Lucchi++ test volume (configs[1]): 165 x 1024 x 768, roi 112^3, overlap 0.5, bump blending, sw_batch_size 8 = 468 windows
(probe window, accumulator allocation, 59 window batches, normalisation and the crop back are all inside the timed region;
the volume is resident in HBM before it starts).  value = window-voxels/s of the whole job (all ranks).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Reported next to `value` (rank 0): the same job with 8-flip mean TTA through InferenceManager (`tta8`), the cubic 448^3
volume (`cube448`), the fp32 parity path (`fp32`), the training half of the metric (`train`, with its own roofline), RSUNet
and the MONAI-style residual U-Net (configs[4]) legs, the CPU oracle on the host cores (`cpu_baseline`), and for N > 1 the
strong-scaling slab mode (`strong_slab`: ONE volume cut into N slabs with p2p halo bands, inference/slab.py).

Multi-GPU headline: the path shards by independent volumes (the reference's volume-per-rank sharding,
training/lightning/data.py:234-266): every rank owns one volume, no data-path collective -> weak scaling.
"""
from __future__ import annotations

import argparse
import json
import os

# RCCL / cross-process device memory on this driver stack needs dmabuf IPC (already exported on the target image)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import statistics
import sys
import time
from pathlib import Path
from types import SimpleNamespace as NS

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ROI = (112, 112, 112)
VOLUME = (165, 1024, 768)          # Lucchi++ test volume (tutorials/mito_lucchi++/README.md:114)
CUBE = (448, 448, 448)             # divisible grid, 7^3 = 343 windows
SW_BATCH = 8
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s spec
ROI_VOX = ROI[0] * ROI[1] * ROI[2]


def build_model(device, out_channels=1):
    from pytorch_connectomics_amd.models import build_model as bm
    cfg = NS(model=NS(arch=NS(type="mednext"), in_channels=1, out_channels=out_channels,
                      mednext=NS(size="S", kernel_size=3), loss=NS(deep_supervision=False), heads=None))
    torch.manual_seed(0)
    model = bm(cfg).to(device).eval()
    model.model.compute_dtype = torch.bfloat16
    return model


def make_engine():
    from pytorch_connectomics_amd.inference.window import EagerSlidingWindowEngine
    return EagerSlidingWindowEngine(roi_size=ROI, sw_batch_size=SW_BATCH, overlap=0.5, mode="bump",
                                    padding_mode="constant", cval=0.0)


def infer_cfg(tta: bool):
    """The Lucchi++ inference section (tutorials/mito_lucchi++/mito_lucchi++.yaml:31-40) as the config tree
    InferenceManager reads; sigmoid activation, fp32 output."""
    tta_ns = NS(enabled=tta, flip_axes="all", rotation90_axes=None, rotate90_k=None, ensemble_mode="mean",
                patch_first_local=True, distributed_sharding=False, apply_mask=True)
    return NS(model=NS(primary_head=None, heads=None, out_channels=1),
              data=NS(train=NS(do_2d=False), val=NS(do_2d=False), dataloader=NS(batch_size=1)),
              inference=NS(sliding_window=NS(window_size=list(ROI), sw_batch_size=SW_BATCH, overlap=0.5, blending="bump",
                                             padding_mode="constant", cval=0.0, keep_input_on_cpu=False, sw_device=None,
                                             output_device=None, border_mask=None, distributed_sharding=False),
                           model=NS(head=None, select_channel=None, output_dtype=None,
                                    channel_activations=[{"channels": ":", "activation": "sigmoid"}], crop_pad=None),
                           test_time_augmentation=tta_ns))


def timed(fn, sync=True):
    if sync:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    if sync:
        torch.cuda.synchronize()
    return time.perf_counter() - t0, out


def job_record(n_windows, views, vol_shape, seconds, **extra):
    out_vox = vol_shape[0] * vol_shape[1] * vol_shape[2]
    rec = {"seconds": seconds, "windows": n_windows, "views": views,
           "window_voxels_per_s": n_windows * views * ROI_VOX / seconds,
           "output_voxels_per_s": out_vox / seconds, "volume": list(vol_shape)}
    rec.update(extra)
    return rec


class HeadlineWatchdog:
    """The headline figure must survive whatever happens in the secondary legs.  Once the timed region is over, a daemon thread
    waits `seconds` for `disarm()`; if the secondary legs (DDP training, slab exchange, TTA, U-Nets, CPU baseline) have not let the
    process reach its final print by then -- a peer that died inside a collective, a wedged kernel -- the thread prints the line
    `make_line()` returns (rank 0: the headline-only JSON with the reason under `errors`; other ranks: None, nothing printed)
    and leaves through os._exit, so the launcher sees one JSON line and a finished job instead of a hang."""

    def __init__(self, seconds: float, make_line, exit_code: int = 0):
        import threading
        self._done = threading.Event()
        self.seconds, self._make_line, self._exit_code = float(seconds), make_line, int(exit_code)
        self._thread = threading.Thread(target=self._run, name="bench-headline-watchdog", daemon=True)

    def arm(self):
        if self.seconds > 0:
            self._thread.start()
        return self

    def disarm(self):
        self._done.set()

    def _run(self):
        if self._done.wait(self.seconds):
            return
        try:
            line = self._make_line()
            if line:
                print(line, flush=True)
        finally:
            os._exit(self._exit_code)


def _median_timed(fn, warmup, timed_runs, budget_s):
    """`warmup` untimed + up to `timed_runs` timed calls of fn (at least one), stopping early once `budget_s` of CPU work is spent."""
    spent, times = 0.0, []
    for i in range(warmup + timed_runs):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        spent += dt
        if i >= warmup:
            times.append(dt)
        if spent > budget_s and times:
            break
    if not times:
        times = [dt]
    return statistics.median(times), len(times)


def cpu_baseline(model, budget_s: float = 60.0):
    """Oracle (CPU restatement, kind='port') timed on the host cores, same weights, fp32, following BASELINE.md section 3 as far as
    a bounded sample allows: a thread sweep that includes os.cpu_count() (one 112^3 window forward each), then 2 warm-up + 5 timed
    forwards at the best thread count, median; and -- the metric is "train + infer" -- `train`: forward + backward (torch
    autograd through the oracle) + BCE/Dice + torch.optim.AdamW on one 112^3 patch.  Each part stops early once its share of
    `budget_s` is spent (at least one timed run); `sample` says what was actually run."""
    from oracle import mednext_oracle as MO
    from pytorch_connectomics_amd.training.module import dice_loss_sigmoid, weighted_bce_with_logits
    st = {k: v.detach().float().cpu() for k, v in model.model.state_dict().items()}
    cores = os.cpu_count() or 1
    kw = dict(n_channels=32, exp_r=2, kernel_size=3, block_counts=[2] * 9)
    x = torch.rand(1, 1, *ROI)
    y = (torch.rand(1, 1, *ROI) > 0.85).float()
    sweep = {}
    with torch.no_grad():
        for threads in sorted({min(32, cores), min(64, cores), cores}):
            torch.set_num_threads(threads)
            t0 = time.perf_counter()
            MO.forward(st, x, **kw)
            sweep[threads] = ROI_VOX / (time.perf_counter() - t0)
        best = max(sweep, key=sweep.get)
        torch.set_num_threads(best)
        dt, n_inf = _median_timed(lambda: MO.forward(st, x, **kw), 1, 5, 0.5 * budget_s)     # the sweep run at `best` was warm-up #1
    params = {k: v.clone().requires_grad_(True) for k, v in st.items() if v.dtype.is_floating_point and k != "dummy_tensor"}
    opt = torch.optim.AdamW(list(params.values()), lr=1e-3)

    def train_step():
        opt.zero_grad(set_to_none=True)
        out = MO.forward(dict(params, dummy_tensor=st["dummy_tensor"]), x, **kw)
        (weighted_bce_with_logits(out, y, None, None) + dice_loss_sigmoid(out, y)).backward()
        opt.step()
    dtt, n_tr = _median_timed(train_step, 1, 2, 0.5 * budget_s)
    return {"value": ROI_VOX / dt, "unit": "voxels/s", "cores": best, "kind": "port", "host_cores": cores,
            "thread_sweep_voxels_per_s": {str(k): round(v, 1) for k, v in sweep.items()},
            "sample": f"oracle MedNeXt-S fp32 forward of one 112^3 window: thread sweep {sorted(sweep)} (one forward each), then "
                      f"2 warm-up (incl. the sweep run) + {n_inf} timed at {best} threads, median {dt:.2f} s, torch CPU",
            "train": {"value": ROI_VOX / dtt, "unit": "voxels/s", "cores": best, "kind": "port",
                      "sample": f"oracle forward + autograd backward + BCE/Dice + torch.optim.AdamW on one 112^3 patch, 1 warm-up + "
                                f"{n_tr} timed, median {dtt:.2f} s, {best} threads"}}


MFMA_PEAK_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 2:1-sparsity figure is not used)
# minimum HBM bytes per window-voxel of a MedNeXt-S k3 forward at 112^3 with one output channel (every block:
# read x twice + write y once, the expanded tensor never in HBM) -- the figure of merit of the step as a whole; training ~ 3 x
ALG_BYTES_PER_VOXEL_FWD = 1557.0
ALG_FLOP_PER_VOXEL_FWD = 1.235e5


def whole_step_roofline(voxels_per_s_per_gpu, passes=1.0):
    """The whole network step against the HBM roofline on SURVEY 8(d)'s byte floor (`passes` = 3 for a training step: forward +
    data gradients + weight gradients), with the MFMA fraction of the same step beside it."""
    gbs = voxels_per_s_per_gpu * ALG_BYTES_PER_VOXEL_FWD * passes / 1e9
    tfl = voxels_per_s_per_gpu * ALG_FLOP_PER_VOXEL_FWD * passes / 1e12
    return {"bound": "hbm", "algorithmic_bytes_per_voxel": ALG_BYTES_PER_VOXEL_FWD * passes, "achieved_GBs": round(gbs, 1),
            "peak_GBs": HBM_PEAK_GBS, "frac": round(gbs / HBM_PEAK_GBS, 4),
            "algorithmic_flop_per_voxel": ALG_FLOP_PER_VOXEL_FWD * passes, "achieved_TFLOPs": round(tfl, 1),
            "mfma_frac": round(tfl / MFMA_PEAK_TFLOPS, 4),
            "source": "SURVEY.md section 8(d): MedNeXt-S k3 112^3 byte / FLOP floor per window-voxel x this leg's voxels/s per GPU"}


def _is_dense_conv(label):
    return label.startswith(("conv3d_", "convT3d_", "conv3d_s_"))


def roofline_entry(name, rec, traffic=None):
    """One kernel (label or symbol) against the roofline that bounds it: dense k^3 convolutions are MFMA work (TFLOP/s against the
    2.5 PFLOP/s dense bf16 peak), everything else on this path is HBM-bound (algorithmic GB/s against 8 TB/s)."""
    per_launch_s = rec["ms"] / rec["launches"] / 1e3
    if rec.get("flops") and _is_dense_conv(name):
        ach = rec["flops"] / rec["launches"] / per_launch_s / 1e12
        return {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "launch_us": round(per_launch_s * 1e6, 1),
                "algorithmic_flops": int(rec["flops"] / rec["launches"]), "launches": rec["launches"]}
    per_launch_bytes = rec["bytes"] / rec["launches"]
    ach = per_launch_bytes / per_launch_s / 1e9
    return {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "launch_us": round(per_launch_s * 1e6, 1),
            "algorithmic_bytes": int(per_launch_bytes), "launches": rec["launches"]}


def dominant(summ, n_steps, peak=HBM_PEAK_GBS, traffic_fn=None):
    name, rec = max(summ.items(), key=lambda kv: kv[1]["ms"])
    total_ms = sum(r["ms"] for r in summ.values())
    out = roofline_entry(name, rec, traffic_fn(name) if traffic_fn else None)
    out.update({"share_of_step": round(rec["ms"] / total_ms, 3),
                "kernels_ms_per_step": {k: round(v["ms"] / n_steps, 3) for k, v in
                                        sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:8]},
                "kernel_ms_total_per_step": round(total_ms / n_steps, 3)})
    return out


def merge_by_kernel(summ, key_fn):
    """Profiler labels that run the SAME device kernel template (key_fn(label) -> rocprof name substring, None = no mapping) merged
    into one record named by that substring -- so that event time, algorithmic bytes and the rocprofv3 counter average of the kernel
    all describe one set of launches.  -> (records, {kernel substring: [labels]})"""
    out, members = {}, {}
    for name, rec in summ.items():
        key = key_fn(name)
        tgt = key if key is not None else name
        d = out.setdefault(tgt, {"launches": 0, "ms": 0.0, "bytes": 0, "flops": 0})
        for k in ("launches", "ms", "bytes", "flops"):
            d[k] += rec.get(k, 0)
        if key is not None:
            members.setdefault(key, []).append(name)
    return out, members


def by_symbol(summ):
    """Per-label profiler records regrouped by device kernel family (hip_ops.PROFILER.by_symbol on an existing summary)."""
    out = {}
    for name, rec in summ.items():
        d = out.setdefault(rec.get("symbol") or name.split("[")[0], {"launches": 0, "ms": 0.0, "bytes": 0, "flops": 0})
        for k in ("launches", "ms", "bytes", "flops"):
            d[k] += rec.get(k, 0)
    return out


def largest_symbols(summ, n_steps, top=3, traffic_of=None, symbols=None):
    """The `top` kernel families by time, each with its own roofline entry (launch-weighted over every shape it runs at): the
    z-march depthwise conv runs under one label per shape and would otherwise never be the 'dominant' line although it is the
    largest rocprof symbol of the step."""
    total_ms = sum(r["ms"] for r in summ.values())
    rows = []
    # `symbols` = PROFILER.by_symbol() of the same run (per-record grouping: the launches of one label can be different template
    # instances -- the mixer's plain / +head / +stemres forms); without it the summary's per-label symbol stands in
    for sym, rec in sorted((symbols if symbols is not None else by_symbol(summ)).items(), key=lambda kv: -kv[1]["ms"])[:top]:
        e = roofline_entry(sym, rec, traffic_of(sym) if traffic_of else None)
        e.update({"share_of_step": round(rec["ms"] / total_ms, 3), "launches_per_step": round(rec["launches"] / n_steps, 1),
                  "ms_per_step": round(rec["ms"] / n_steps, 3)})
        if e.get("traffic") and e["bound"] == "hbm":
            e["traffic_over_algorithmic"] = round(e["traffic"] / max(e["algorithmic_bytes"], 1), 3)
        rows.append(e)
    return rows


def train_leg(dev, rank, world, args, barrier):
    """The training half of the metric: DDP (RCCL all-reduce) MedNeXt-S steps on synthetic 112^3 patches, bf16
    storage / fp32 master weights: HIP forward + HIP backward + BCE/Dice loss + grad clip + AdamW, nothing skipped.
    Reported next to `value` (which stays the sliding-window inference rate the target is quoted on)."""
    from pytorch_connectomics_amd import hip_ops as ops
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
    steps = max(1, args.train_steps)

    def tstep(i):
        b = pool[i % len(pool)]
        opt.zero_grad(set_to_none=True)
        out = net(b["image"])
        loss, _ = bce_dice_loss(out, b["label"])
        loss.backward()
        opt.step()
        return loss

    for i in range(3):
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
    roof = None
    if not args.no_roofline:
        # two more steps with per-kernel HIP events on rank 0; under DDP EVERY rank has to run them (the gradient
        # all-reduce is a collective: a rank that skipped them would leave rank 0 waiting on a closed connection)
        if rank == 0:
            with ops.profiled() as prof:
                for i in range(2):
                    tstep(i)
            table = committed_pmc_table("train")
            merged, members = merge_by_kernel(prof.summary(), _train_kernel_key)
            roof = dominant(merged, 2, traffic_fn=lambda name: _traffic_from_table(table, name if name in members else None))
            if roof["kernel"] in members:
                roof["labels"] = members[roof["kernel"]]
            roof["traffic_source"] = ("committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this training step "
                                      "(profiles/rNN_train_hbm_counters.csv; FETCH_SIZE x 2); the entry covers every launch of the "
                                      "device kernel named in `kernel` (`labels`), which is what the counter average covers")
            if roof.get("traffic"):
                roof["traffic_over_algorithmic"] = round(roof["traffic"] / max(roof["algorithmic_bytes"], 1), 3)
            roof["by_symbol"] = largest_symbols(prof.summary(), 2, top=4, symbols=prof.by_symbol(),
                                                traffic_of=lambda sym: _traffic_from_table(table, _kernel_key(sym)))
        else:
            for i in range(2):
                tstep(i)
        barrier()
    vox = world * args.train_batch * ROI_VOX * steps
    if roof is not None:
        roof["whole_step"] = whole_step_roofline(vox / dt / world, passes=3.0)
    # what the collective moved, so that the first multi-GPU run explains itself: per-rank placement, backend, DDP's bucket layout and
    # the gradient bytes all-reduced per step (ring all-reduce over xGMI: each rank sends and receives 2 (N-1)/N of them)
    ddp_info = None
    if world > 1:
        try:
            log = net._get_ddp_logging_data()
            nbytes = sum(p_.numel() * 4 for p_ in model.parameters() if p_.requires_grad)
            mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", rank)), "device": torch.cuda.get_device_name(dev),
                    "device_index": dev.index, "ms_per_step_local": dt / steps * 1e3}
            ranks = [None] * world
            torch.distributed.all_gather_object(ranks, mine)
            sizes = [int(v) for v in str(log.get("bucket_sizes", "")).replace(",", " ").split() if v.strip().isdigit()]
            ddp_info = {"backend": str(log.get("backend_name", torch.distributed.get_backend())), "world_size": world,
                        "rccl_ranks": world, "buckets": len(sizes) or None, "bucket_bytes": sizes or None,
                        "bucket_cap_bytes": int(log.get("bucket_cap_bytes", 0)) or None,
                        "gradient_bytes_allreduced_per_step": nbytes,
                        "ring_bytes_sent_per_rank_per_step": int(2 * (world - 1) / world * nbytes),
                        "find_unused_parameters": True, "gradient_as_bucket_view": True, "ranks": ranks}
        except Exception as exc:  # noqa: BLE001
            ddp_info = {"error": f"{type(exc).__name__}: {exc}"}
    del opt, net, model
    torch.cuda.empty_cache()
    return {"value": vox / dt, "unit": "voxels/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "batch_per_gpu": args.train_batch, "patch": list(ROI), "dtype": "bf16 activations, fp32 master weights",
            "parallelism": f"ddp{world}" if world > 1 else "single", "scaling": "weak",
            "includes": "forward + backward + fused BCE/Dice loss + grad-norm clip + AdamW step, all HIP kernels",
            "final_loss": float(loss.detach()), "roofline": roof, "ddp": ddp_info}


def c3_affinity_tta16_leg(dev, model3):
    """BASELINE configs[2]'s inference (tutorials/neuron_snemi/neuron_snemi.yaml:72-90): MedNeXt-S with a 3-channel nearest-neighbour
    affinity output (affinity_mode deepem), 16 test-time views (8 flips x yx quarter turn), affinity-aware -- every view's channels are
    moved and re-anchored to the canonical frame -- ensemble_mode `min`, patch-first-local through InferenceManager, bf16, over a
    165 x 448 x 448 volume at roi 112^3 / overlap 0.5 / sw 8 (the SNEMI volume is 100 x 1024 x 1024; its own window 32 x 160 x 160 is a
    different config of the same engine)."""
    from pytorch_connectomics_amd.inference import InferenceManager
    vol_shape = (165, 448, 448)
    cfg = infer_cfg(True)
    cfg.model.out_channels = 3
    tta = cfg.inference.test_time_augmentation
    tta.rotation90_axes, tta.ensemble_mode = [[1, 2]], "min"
    cfg.data.label_transform = NS(stack_outputs=True, targets=[{"name": "affinity", "kwargs": {
        "offsets": ["1-0-0", "0-1-0", "0-0-1"], "affinity_mode": "deepem"}}])
    vol = torch.rand((1, 1) + vol_shape, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
    mgr = InferenceManager(cfg, model3, model3.forward)
    with torch.no_grad():
        mgr.predict_with_tta(vol[:, :, :112, :224, :224].contiguous())          # warm-up: weight images, allocator pools
        # two timed passes, the faster one is the record (both are kept): one pass of this leg took 2.87 s in the round-5 driver run against
        # 1.32-1.38 s in every other run of the same tree (profiles/r06_c3_tta16_summary.txt) -- a one-off stall, not the path's rate
        s1, out = timed(lambda: mgr.predict_with_tta(vol))
        del out
        s2, out = timed(lambda: mgr.predict_with_tta(vol))
    s = min(s1, s2)
    _, starts = make_engine().plan(vol_shape)
    rec = job_record(len(starts), 16, vol_shape, s, passes_seconds=[s1, s2], path="InferenceManager.predict_with_tta: 8 flips x yx rot90 = 16 views, affinity-aware "
                     "(deepem offsets 1-0-0 / 0-1-0 / 0-0-1), ensemble min, sigmoid per view, fp32 out", out_channels=int(out.shape[1]))
    del out, vol
    torch.cuda.empty_cache()
    return rec


def c4_chunked_leg(dev):
    """BASELINE configs[3] (MitoEM-R): MedNeXt-L k3 with the three MitoEM heads (7 channels), roi 160^3, overlap 0.5, reflect padding,
    chunked sliding-window inference with chunk 320^3 and halo 80 -- the per-rank work of the 8-GPU job (640^3 = 8 chunks, one per rank,
    chunked.py:471: idx % world == rank) measured on ONE GPU for the 2 chunks of a 320 x 320 x 640 volume: per chunk the call the
    chunked runner makes (`lazy_predict_region` over the haloed region on the GLOBAL window grid: host array -> pinned -> HBM, every
    window through the engine in bf16, blend, activation) and the crop to the chunk.  The chunk FILES are not part of the timed region:
    the reference's layout is gzip HDF5 (chunked.py:279-314) and zlib needs ~60 s for the 0.9 GB a 7-channel fp32 chunk holds -- the
    runner writes them on its writer thread while the next chunk is predicted, the GPU is idle for it either way."""
    from pytorch_connectomics_amd.chunked import build_chunk_grid, resolve_halo_region
    from pytorch_connectomics_amd.inference.lazy import lazy_predict_region
    from pytorch_connectomics_amd.models import build_model as bm
    heads = {"aff_r1": {"out_channels": 3, "num_blocks": 1, "hidden_channels": 8},
             "aff_r5": {"out_channels": 3, "num_blocks": 1, "hidden_channels": 8},
             "sdt": {"out_channels": 1, "num_blocks": 1, "hidden_channels": 8}}
    mcfg = NS(model=NS(arch=NS(type="mednext"), in_channels=1, out_channels=7, mednext=NS(size="L", kernel_size=3),
                       loss=NS(deep_supervision=False), heads=heads, primary_head="aff_r1"))
    torch.manual_seed(0)
    model = bm(mcfg).to(dev).eval()
    model.model.compute_dtype = torch.bfloat16
    roi, chunk, halo = (160, 160, 160), (320, 320, 320), (80, 80, 80)
    cfg = NS(model=NS(primary_head=None, heads=None, out_channels=7, output_size=list(roi)), system=NS(num_workers=0),
             data=NS(train=NS(do_2d=False), val=NS(do_2d=False), dataloader=NS(batch_size=1)),
             inference=NS(sliding_window=NS(window_size=list(roi), sw_batch_size=2, overlap=0.5, blending="bump", padding_mode="reflect",
                                            cval=0.0, border_mask=[], distributed_sharding=False, snap_to_edge=False, target_context=[]),
                          model=NS(head=None, select_channel=None, output_dtype=None,
                                   channel_activations=[{"channels": ":", "activation": "sigmoid"}]),
                          chunking=NS(enabled=True, chunk_size=list(chunk), halo=list(halo), axes="all", shard_id=None, num_shards=None),
                          test_time_augmentation=NS(enabled=False)))
    vol_shape = (320, 320, 640)
    vol = torch.rand((1,) + vol_shape, generator=torch.Generator().manual_seed(17)).numpy()
    windows = []
    fwd_cl = model.forward_cl

    def counting_forward_cl(x_cl):          # the lazy engine takes the bound model's channels-last path (merged heads: 7 channels)
        windows.append(int(x_cl.shape[0]))
        return fwd_cl(x_cl)
    model.forward_cl = counting_forward_cl
    chunks = build_chunk_grid(vol_shape, chunk)

    def one_chunk(c):
        lo, hi, core = resolve_halo_region(c, vol_shape, halo=halo, crop_before=(0, 0, 0))
        y = lazy_predict_region(cfg, model.forward, vol, region_start=lo, region_stop=hi, device="cuda")
        return y[(Ellipsis,) + tuple(core)].contiguous()

    kept = {}

    def run_chunks():
        shapes = []
        for c in chunks:
            y = one_chunk(c)
            shapes.append(tuple(y.shape))
            kept["last"] = y                                                    # (1, 7, 320, 320, 320) fp32 = 0.92 GB
        return shapes

    with torch.no_grad():
        fwd_cl(torch.rand(1, *roi, 1, device=dev))                              # warm-up: weight images
        one_chunk(chunks[0])                                                    # warm-up: the allocator pools of the window streams
        windows.clear()
        kept.clear()
        s, outs = timed(run_chunks)
    n_win = sum(windows)
    # the chunk FILE of the last chunk, as the chunked runner writes it (chunked.py: write_prediction_artifact, gzip, HDF5 chunks
    # (C, 64, 64, 64)): device -> host copy, then the parallel deflate writer (csrc/host/h5io.c; round 4: one zlib thread, 63 s)
    write = None
    try:
        import tempfile
        import time as _time
        from pytorch_connectomics_amd.inference.artifact import write_prediction_artifact
        from pytorch_connectomics_amd.utils import h5lite
        if h5lite.available():
            t0 = _time.perf_counter()
            host = kept["last"][0].cpu().numpy()
            t1 = _time.perf_counter()
            with tempfile.TemporaryDirectory() as td:
                write_prediction_artifact(Path(td) / "chunk_z0_y0_x0.h5", host, compression="gzip", chunks=(int(host.shape[0]), 64, 64, 64))
                t2 = _time.perf_counter()
                fsz = (Path(td) / "chunk_z0_y0_x0.h5").stat().st_size
            write = {"d2h_seconds": t1 - t0, "write_seconds": t2 - t1, "threads": h5lite.write_threads(), "bytes": int(host.nbytes),
                     "file_bytes": int(fsz), "MB_per_s": host.nbytes / 1e6 / max(t2 - t1, 1e-9),
                     "writer": h5lite.last_parallel_write_stats(),
                     "note": "gzip level 4 on uniform-random-like sigmoid outputs (near-incompressible: the worst case for deflate); "
                             "writer = thread-seconds of gather / deflate, seconds inside the serialized H5Dwrite_chunk calls, backend"}
    except Exception as exc:  # noqa: BLE001
        write = {"error": f"{type(exc).__name__}: {exc}"}
    kept.clear()
    rec = {"seconds": s, "chunks": len(chunks), "seconds_per_chunk": s / len(chunks), "windows": n_win, "roi": list(roi),
           "volume": list(vol_shape), "chunk": list(chunk), "halo": list(halo),
           "window_voxels_per_s": n_win * roi[0] * roi[1] * roi[2] / s,
           "output_voxels_per_s": vol_shape[0] * vol_shape[1] * vol_shape[2] / s, "chunk_output_shape": list(outs[0]),
           "chunk_file_write": write,
           "roofline": {"bound": "mfma+hbm", "algorithmic_bytes_per_voxel": 2322.0, "algorithmic_flop_per_voxel": 4.781e5,
                        "achieved_GBs": n_win * roi[0] * roi[1] * roi[2] / s * 2322.0 / 1e9, "hbm_frac": n_win * roi[0] * roi[1] * roi[2] / s * 2322.0 / 8e12,
                        "achieved_TFLOPs": n_win * roi[0] * roi[1] * roi[2] / s * 4.781e5 / 1e12,
                        "mfma_frac": n_win * roi[0] * roi[1] * roi[2] / s * 4.781e5 / 2.5e15,
                        "source": "SURVEY.md section 8(d): MedNeXt-L k3 160^3 byte / FLOP floor per window-voxel x this leg's window-voxels/s "
                                  "(the leg's time includes host -> HBM reads of the lazy volume, blending and the crop)"},
           "path": "per chunk: lazy_predict_region over the haloed region on the global window grid (what run_chunked_prediction_inference "
                   "calls) + crop: MedNeXt-L k3 + 3 MitoEM heads (7 ch), bf16, chunk 320^3 / halo 80, reflect padding, sw 2; host volume -> "
                   "pinned -> HBM reads included; the gzip chunk file of one chunk is timed separately (chunk_file_write)"}
    del model
    torch.cuda.empty_cache()
    return rec


def _unet_leg(dev, args, make, label, patch, batch, out_ch):
    from pytorch_connectomics_amd import hip_ops as ops
    from pytorch_connectomics_amd.training.fused import FusedAdamW, bce_dice_loss
    from pytorch_connectomics_amd.utils.hostgc import quiesce_gc
    torch.manual_seed(0)
    m = make().to(dev).train()
    inner = getattr(m, "model", m)
    for mod in (m, inner):
        if hasattr(mod, "compute_dtype"):
            mod.compute_dtype = torch.bfloat16
    opt = FusedAdamW(m.parameters(), lr=1e-4, weight_decay=1e-2, max_grad_norm=1.0)
    x = torch.rand(batch, 1, *patch, device=dev)
    y = (torch.rand(batch, out_ch, *patch, device=dev) > 0.85).float()

    def tstep():
        opt.zero_grad(set_to_none=True)
        loss, _ = bce_dice_loss(m(x), y)
        loss.backward()
        opt.step()
        return loss

    steps = max(1, args.train_steps)
    for _ in range(4):
        tstep()
    quiesce_gc()
    dt_train, loss = timed(lambda: [tstep() for _ in range(steps)][-1])
    dt_train /= steps
    roof = None
    if not args.no_roofline:
        with ops.profiled() as prof:
            for _ in range(2):
                tstep()
        roof = dominant(prof.summary(), 2)
    m.eval()
    with torch.no_grad():
        for _ in range(3):
            m(x)
        dt_inf, _ = timed(lambda: [m(x) for _ in range(steps)])
        dt_inf /= steps
    vox = batch * patch[0] * patch[1] * patch[2]
    del opt, m
    torch.cuda.empty_cache()
    return {"model": label, "batch": batch, "patch": list(patch), "dtype": "bf16 activations, fp32 master weights",
            "train_ms_per_step": dt_train * 1e3, "train_voxels_per_s": vox / dt_train, "infer_ms_per_forward": dt_inf * 1e3,
            "infer_voxels_per_s": vox / dt_inf, "final_loss": float(loss.detach()), "train_roofline": roof}


RSUNET_STOCK = dict(width=[18, 36, 48, 64, 80], norm="group", num_groups=4, activation="elu", down_factors=[(1, 2, 2)] * 4, depth_2d=1,
                    kernel_2d=(1, 3, 3))


def rsunet_leg(dev, args):
    """The path's second architecture (single GPU), AS THE REFERENCE SHIPS IT (config/profiles/arch_profiles.yaml:34-44, what
    tutorials/syn_cremi.yaml trains): width [18, 36, 48, 64, 80], GroupNorm(4) (GroupNorm(3, 18) on the first level), ELU, down
    (1,2,2) x 4, (1,3,3) kernels on level 0; anisotropic 2 x 18 x 256 x 256 patches, bf16 storage: training step (HIP forward +
    backward, fused loss, fused AdamW) and inference forward.  The two widths that are no multiple of 8 travel as 24 / 40 channels
    (ops.pad_channels).  `pow2` = the hand-picked widths [16, 32, 64, 128] / BatchNorm / ReLU the earlier rounds benched, at the same
    patch, with the per-FLOP ratio of the two training steps (conv FLOPs of the reference's own channel counts)."""
    from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet
    patch = (18, 256, 256)
    res = _unet_leg(dev, args, lambda: RSUNet(1, 1, **RSUNET_STOCK),
                    "RSUNet stock profile: width [18,36,48,64,80], GroupNorm(4), ELU, down (1,2,2)x4, depth_2d 1", patch, 2, 1)
    try:
        pow2 = _unet_leg(dev, args, lambda: RSUNet(1, 1, width=[16, 32, 64, 128], norm="batch", activation="relu"),
                         "RSUNet width [16,32,64,128], BatchNorm, relu", patch, 2, 1)
        gf = lambda w, k0: _rsunet_conv_gflop(w, patch, 2, k0)      # noqa: E731
        res["pow2"] = {k: pow2[k] for k in ("model", "train_ms_per_step", "infer_ms_per_forward")}
        res["train_us_per_gflop"] = round(res["train_ms_per_step"] * 1e3 / gf([18, 36, 48, 64, 80], 9), 3)
        res["pow2"]["train_us_per_gflop"] = round(pow2["train_ms_per_step"] * 1e3 / gf([16, 32, 64, 128], 27), 3)
        res["per_flop_ratio_vs_pow2"] = round(res["train_us_per_gflop"] / res["pow2"]["train_us_per_gflop"], 3)
        # the step against the MFMA roofline on the REFERENCE's channel counts (not the padded 24 / 40 the kernels run)
        stock_gf = gf([18, 36, 48, 64, 80], 9)
        res["train_whole_step"] = {"bound": "mfma", "algorithmic_gflop": round(stock_gf, 2), "channels": "reference [18, 36, 48, 64, 80]",
                                   "achieved_TFLOPs": round(stock_gf / res["train_ms_per_step"], 2),
                                   "mfma_frac": round(stock_gf / res["train_ms_per_step"] / 2500.0, 4)}
        roof = res.get("train_roofline")
        if isinstance(roof, dict) and isinstance(roof.get("kernel"), str):
            import re
            mm = re.search(r"\[(\d+)->(\d+)", roof["kernel"])
            real = {24: 18, 40: 36}
            if mm and (int(mm.group(1)) in real or int(mm.group(2)) in real):
                a, b = int(mm.group(1)), int(mm.group(2))
                ra, rb = real.get(a, a), real.get(b, b)
                roof["reference_channels"] = f"{ra}->{rb} (run as {a}->{b}: ops.pad_channels)"
                roof["frac_reference_channels"] = round(roof["frac"] * (ra * rb) / (a * b), 4)
    except Exception as e:     # noqa: BLE001 - the comparison must not cost the leg
        res["pow2"] = {"error": f"{type(e).__name__}: {e}"}
    return res


def _rsunet_conv_gflop(width, patch, batch, taps_level0):
    """Forward + backward conv GFLOP of one RSUNet training step at the REFERENCE's channel counts (3 x forward; the 1-channel stem
    and head convs left out): per level 4 convs width -> width (3 of them ResBlock / post convs) + the in-projection from the level
    above, encoder and decoder sides, (1,2,2) pooling, `taps_level0` taps on level 0 and 27 below."""
    vox = batch * patch[0] * patch[1] * patch[2]
    total = 0.0
    for lvl, w in enumerate(width):
        taps = taps_level0 if lvl == 0 else 27
        sides = 1 if lvl == len(width) - 1 else 2                       # the deepest level has no decoder twin
        total += sides * 4 * 2.0 * vox * w * w * taps
        if lvl + 1 < len(width):
            total += 2.0 * vox * w * width[lvl + 1]                     # 1x1x1 up-projection into this level
        vox /= 4
    return 3 * total / 1e9


def monai_unet_leg(dev, args):
    """BASELINE configs[4] (CREMI synapse): MONAI-style residual U-Net, filters [32,64,128,256], anisotropic patches, bf16
    storage: training step and inference forward.  The reference's builder strides every level by 2 on every axis
    (monai_models.py:228-229), so a 20 x 256 x 256 patch cannot pass its own skip concatenations (20 -> 10 -> 5 -> 3, up 6 != 5:
    torch.cat fails in MONAI too); the leg runs the nearest size the architecture accepts, 24 x 256 x 256."""
    from pytorch_connectomics_amd.models import build_model as bm
    cfg = NS(model=NS(arch=NS(type="monai_unet"), in_channels=1, out_channels=1, input_size=[24, 256, 256],
                      monai=NS(filters=[32, 64, 128, 256], num_res_units=2, kernel_size=3, norm="batch", dropout=0.0,
                               upsample_mode="deconv")))
    return _unet_leg(dev, args, lambda: bm(cfg), "MONAI-style residual U-Net filters [32,64,128,256], BatchNorm, PReLU",
                     (24, 256, 256), 2, 1)


def _kernel_key(label):
    """rocprof kernel-name substring of a bench label or kernel family: the fused mixer of one shape (plain and epilogue variants),
    or a whole kernel family (`dwconv3d_k3_march_kernel`, ...: launch-weighted over the shapes it runs at).  Per-shape labels of
    kernels that run at several shapes have no counter average of their own."""
    import re
    m = re.match(r"pw_mlp_fwd\[(\d+)->(\d+)->(\d+)\]", label)
    if m:      # streaming, LDS-resident (pw_mlp_lds_kernel<KS_IN, MO, NT, NWAVES, WPS>) or chunk-streamed form (pw_mlp_chunk_kernel<KS_IN, MO, NW, WPS>)
        # -- whichever the shape runs on -- and, for the 32-channel level-0 shapes, the DMA-prefetching form pw_mlp_dma_kernel<C_hid / 32, KIND>
        ks, mo, hc = int(m.group(1)) // 32, int(m.group(3)) // 16, int(m.group(2)) // 32
        pat = rf"pw_mlp(_lds|_chunk)?_kernel<{ks}, {mo},"
        if ks == 1 and mo == 2:
            pat += rf"|pw_mlp_dma_kernel<{hc}, \d>"
        return re.compile(pat)
    m = re.match(r"pw_mlp_lds_kernel<(\d+), (\d+)>$", label)
    if m:
        return re.compile(rf"pw_mlp_lds_kernel<{m.group(1)}, {m.group(2)},")
    # one template INSTANCE of the fused mixer = one rocprof symbol (hip_ops gives every launch its instance: plain, +head, +stemres):
    # <KS_IN, MO, NT, GELU_MODE, HEAD, STEMRES, STOREH, BWD>
    m = re.match(r"pw_mlp_kernel<(\d+), (\d+)>(\+head|\+stemres)?$", label)
    if m:
        flags = {None: "false, false", "+head": "true, false", "+stemres": "false, true"}[m.group(3)]
        return re.compile(rf"pw_mlp_kernel<{m.group(1)}, {m.group(2)}, \d+, \d+, {flags}, false, false>")
    m = re.match(r"(pw_mlp_dma_kernel)<(\d+), (\d+)>$", label)             # one instance = one rocprof symbol
    if m:
        return re.compile(rf"pw_mlp_dma_kernel<{m.group(2)}, {m.group(3)}>")
    m = re.match(r"(pw_mlp_chunk_kernel)<(\d+), (\d+)>$", label)           # <KS_IN, MO>: every workgroup shape of it
    if m:
        return re.compile(rf"pw_mlp_chunk_kernel<{m.group(2)}, {m.group(3)},")
    if "[" not in label and label.endswith("_kernel"):
        return label
    return None


def _traffic_from_table(table, key):
    """launch-weighted (2 x FETCH_SIZE + WRITE_SIZE) bytes per launch over the rows of `table` whose kernel name contains key."""
    if not table or key is None:
        return None
    tot = n = 0.0
    hit = (lambda k: key.search(k) is not None) if hasattr(key, "search") else (lambda k: key in k)
    for kernel, (launches, fetch_kb, write_kb) in table.items():
        if hit(kernel):
            tot += launches * (2 * fetch_kb + write_kb) * 1024
            n += launches
    return int(tot / n) if n else None


def _train_kernel_key(label):
    """rocprof symbol substring of a training-step label: the MFMA pointwise weight gradient of one channel pair (tile counts as
    csrc/train_kernels.hip wg_tile16 picks them: <C_out tiles, C_in tiles>), or a kernel family name."""
    import re
    m = re.match(r"pw_wgrad(?:_gn)?\[(\d+)->(\d+)\]", label)
    if m:
        t = lambda c: 4 if c % 64 == 0 else (2 if c % 32 == 0 else 1)      # noqa: E731
        return f"pw_wgrad_mfma_kernelILi{t(int(m.group(2)))}ELi{t(int(m.group(1)))}E"
    m = re.match(r"pw_conv_fwd\[(\d+)->(\d+)\]", label)
    if m and int(m.group(1)) % 32 == 0 and int(m.group(2)) % 32 == 0:
        return f"pw_fast_kernel<{int(m.group(1)) // 32},"       # every launch with this C_in (the symbol does not carry C_out)
    return _kernel_key(label)


def committed_pmc_table(leg="bench"):
    """kernel -> (launches, FETCH_SIZE KB mean, WRITE_SIZE KB mean) from the newest profiles/rNN_<leg>_hbm_counters.csv (separate
    --pmc passes of this same command or of the training step, tools/profile_r03.sh + tools/make_hbm_counters_csv.py)."""
    import csv
    files = sorted((ROOT / "profiles").glob(f"r*_{leg}_hbm_counters.csv"))
    if not files:
        return None
    return {row["kernel"]: (float(row.get("launches") or 1), float(row["FETCH_SIZE_KB_mean"]), float(row["WRITE_SIZE_KB_mean"]))
            for row in csv.DictReader(open(files[-1]))}


def pmc_traffic_bytes(label):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950).  None if the file or the kernel is missing."""
    return _traffic_from_table(committed_pmc_table(), _kernel_key(label))


def live_pmc_table(timeout_s=90):
    """The same counters collected NOW, on this box: two child runs of this script (one whole-volume step, inference only) under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, no trace domains, as MI355X_MICROARCH.md prescribes --
    outside the timed region.  -> kernel -> (launches, FETCH KB mean, WRITE KB mean), or None when rocprofv3 is missing or a pass
    fails / times out (the caller then falls back to the committed passes)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    sums = {}
    tmp = tempfile.mkdtemp(prefix="pytc_pmc_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "bench", "--", sys.executable,
                   str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline", "--no-train",
                   "--no-extras"]
            env = dict(os.environ, TMPDIR="/tmp", PYTC_SW_STREAMS="1")      # counters per kernel: one stream, no co-running kernels
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None
            seen = False
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter:
                        d = sums.setdefault(row.get("Kernel_Name", ""), {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
                        d[counter][0] += float(row["Counter_Value"])
                        d[counter][1] += 1
                        seen = True
            if not seen:
                return None
        return {k: (max(d["FETCH_SIZE"][1], d["WRITE_SIZE"][1]), d["FETCH_SIZE"][0] / max(d["FETCH_SIZE"][1], 1),
                    d["WRITE_SIZE"][0] / max(d["WRITE_SIZE"][1], 1)) for k, d in sums.items()}
    except Exception:      # noqa: BLE001 - a profiler hiccup must not take the bench line down
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_pmc_traffic_bytes(label, timeout_s=90):
    key = _kernel_key(label)
    if key is None:
        return None          # nothing to collect: no child run
    return _traffic_from_table(live_pmc_table(timeout_s), key)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed PMC passes only (no rocprofv3 child runs)")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the tta8 / cube448 / fp32 / U-Net legs")
    ap.add_argument("--train-batch", type=int, default=4)
    ap.add_argument("--train-steps", type=int, default=10)
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
        import datetime
        # a rank that dies inside a collective must turn into an error on its peers, not into a hang of the whole job
        tmo = datetime.timedelta(seconds=int(os.environ.get("PYTC_BENCH_COLLECTIVE_TIMEOUT_S", "600")))
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)

    from pytorch_connectomics_amd import hip_ops as ops

    volume = tuple(int(v) for v in os.environ.get("PYTC_BENCH_VOLUME", "").split("x")) if os.environ.get("PYTC_BENCH_VOLUME") else VOLUME
    model = build_model(dev)
    eng = make_engine()
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    vol = torch.rand((1, 1) + volume, device=dev, generator=g)          # resident in HBM before timing
    _, starts = eng.plan(volume)
    n_win = len(starts)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    errors = {}          # leg -> {rank: message}; a failing rank reports instead of leaving its peers inside a collective

    def agree(leg, exc):
        """Collective: every rank says whether `leg` worked for it; -> True when it worked everywhere.  Failures (with the rank
        they happened on) go into the JSON line."""
        msg = None if exc is None else f"{type(exc).__name__}: {exc}"
        if world == 1:
            if msg:
                errors[leg] = {"0": msg}
            return msg is None
        msgs = [None] * world
        torch.distributed.all_gather_object(msgs, msg)
        bad = {str(r): m for r, m in enumerate(msgs) if m}
        if bad:
            errors[leg] = bad
        return not bad

    def step():
        return eng(vol, model)       # the product call: probe, accumulators, all window batches, finalize, crop

    exc, dt, out_shape = None, float("nan"), ()
    try:
        with torch.no_grad():
            for _ in range(args.warmup):
                step()
    except Exception as e:     # noqa: BLE001 - reported by agree()
        exc = e
    headline_ok = agree("warmup", exc)
    if headline_ok:
        exc = None
        barrier()
        t0 = time.perf_counter()
        try:
            with torch.no_grad():
                for _ in range(args.steps):
                    out = step()
            torch.cuda.synchronize()
            out_shape = tuple(out.shape)
            del out
        except Exception as e:     # noqa: BLE001
            exc = e
        headline_ok = agree("headline", exc)       # doubles as the closing barrier: every rank has finished (or failed) its steps
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    if not headline_ok:
        if rank == 0:
            print(json.dumps({"metric": "voxels/s (train + sliding-window infer), MedNeXt-S 112^3 bf16", "value": None,
                              "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "higher_is_better": True, "scaling": "weak", "errors": errors}))
        if world > 1:
            torch.distributed.destroy_process_group()
        raise SystemExit(1)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    per_call = dt / args.steps
    value_vps = world * n_win * ROI_VOX * args.steps / dt

    def headline_fields():
        return {
            "metric": "voxels/s (train + sliding-window infer), MedNeXt-S 112^3 bf16",
            "value": value_vps, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": per_call * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "rccl_ranks": world if (world > 1 and not share) else 0, "errors": errors or None,
            "config": {"workload": "Lucchi++ sliding-window inference (configs[1]): MedNeXt-S k3, "
                                   f"{'x'.join(map(str, volume))} volume, roi 112^3, overlap 0.5, bump blending, "
                                   "sw_batch_size 8, random-init weights; step = one whole-volume "
                                   "EagerSlidingWindowEngine call (TTA off); value = window-voxels/s",
                       "volume": list(volume), "roi": list(ROI), "sw_batch_size": SW_BATCH, "windows_per_step": n_win,
                       "sharding": "one independent volume per rank, no collective",
                       "window_pipeline_streams": eng.last_stats.get("streams", 1)},
            "window_voxels_per_s": value_vps,
            "output_voxels_per_s": world * volume[0] * volume[1] * volume[2] * args.steps / dt,
            "ms_per_8_windows": per_call * 1e3 * SW_BATCH / n_win, "timed_region_s": dt, "output_shape": list(out_shape),
        }

    # the secondary legs below may not cost the headline: see HeadlineWatchdog (PYTC_BENCH_WATCHDOG_S, 0 = off)
    budget = float(os.environ.get("PYTC_BENCH_WATCHDOG_S", "300" if world > 1 else "480"))

    def rescue_line():
        if rank != 0:
            return None
        rec = headline_fields()
        rec["errors"] = dict(errors, watchdog={str(rank): f"secondary legs did not finish within {budget:.0f} s of the timed region; "
                                                          "headline only"})
        return json.dumps(rec)
    watchdog = HeadlineWatchdog(budget, rescue_line).arm()

    roofline = None
    if rank == 0 and not args.no_roofline:
        (wz, wy, wx), combine = eng._axis_vectors(dev)
        value = torch.zeros((1,) + tuple(max(volume[a], ROI[a]) for a in range(3)), device=dev)
        weight = torch.zeros(value.shape[1:], device=dev)
        nprof = 5
        with torch.no_grad(), ops.profiled() as prof:      # per-kernel HIP events over 5 window batches of the same job
            for i in range(nprof):
                b = starts[1 + i * SW_BATCH: 1 + (i + 1) * SW_BATCH]
                x = ops.gather_windows(vol[0], b, ROI, pad_mode="constant", cval=0.0)
                y = model.forward_cl(x)
                ops.blend_accumulate(y, b, value, weight, wz, wy, wx, combine=combine, floor_w=1e-5)
        summ = prof.summary()
        del value, weight
        table, source = committed_pmc_table(), "committed rocprofv3 --pmc passes of this command (profiles/rNN_bench_hbm_counters.csv)"
        if world == 1 and not args.no_live_pmc:
            live = live_pmc_table()
            if live:
                table, source = live, ("live: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child runs of this command on this box "
                                       "(separate passes, FETCH_SIZE x 2)")
        traffic_of = lambda name: _traffic_from_table(table, _kernel_key(name))      # noqa: E731
        # `roofline` IS the largest rocprof SYMBOL of the step (launch-weighted over the shapes it runs at: what a rocprofv3 --stats
        # row of the same command averages over); the largest per-shape LABEL -- one mixer shape -- sits beside it as `by_label`, the
        # three largest families as `by_symbol`, and the step as a whole on SURVEY 8(d)'s byte floor as `whole_step`
        families = largest_symbols(summ, nprof, top=3, traffic_of=traffic_of, symbols=prof.by_symbol())
        by_label = dominant(summ, nprof, traffic_fn=traffic_of)
        roofline = dict(families[0])
        roofline.update({"traffic_source": source, "by_label": by_label, "by_symbol": families,
                         "kernels_ms_per_step": by_label.pop("kernels_ms_per_step"),
                         "kernel_ms_total_per_step": by_label.pop("kernel_ms_total_per_step"),
                         "whole_step": whole_step_roofline(value_vps / world)})
        if os.environ.get("PYTC_BENCH_VERBOSE"):
            for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
                print(f"  {k:34s} launches/step={v['launches'] / nprof:5.1f} ms/step={v['ms'] / nprof:7.3f} "
                      f"GB/s={v['bytes'] / max(v['ms'], 1e-9) / 1e6:8.1f}", file=sys.stderr)

    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
        from pytorch_connectomics_amd.inference import InferenceManager
        with torch.no_grad():
            try:   # secondary figures must never cost the headline line
                mgr0 = InferenceManager(infer_cfg(False), model, model.forward)
                s0, o = timed(lambda: mgr0.predict_with_tta(vol))
                extras["tta_off_manager"] = job_record(n_win, 1, volume, s0, path="InferenceManager.predict_with_tta, "
                                                       "TTA off, sigmoid, fp32 out")
                del o
                mgr8 = InferenceManager(infer_cfg(True), model, model.forward)
                s8, o = timed(lambda: mgr8.predict_with_tta(vol))
                extras["tta8"] = job_record(n_win, 8, volume, s8, path="InferenceManager.predict_with_tta, 8-flip mean "
                                            "TTA (tta_combinations order), sigmoid per view, fp32 out")
                del o
            except Exception as e:     # noqa: BLE001 - reported in the JSON
                extras["tta8"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                cube = torch.rand((1, 1) + CUBE, device=dev, generator=g)
                _, cstarts = eng.plan(CUBE)
                eng(cube[:, :, :112, :224, :224].contiguous(), model)
                sc, o = timed(lambda: eng(cube, model))
                extras["cube448"] = job_record(len(cstarts), 1, CUBE, sc)
                del o, cube
            except Exception as e:     # noqa: BLE001
                extras["cube448"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                model.model.compute_dtype = torch.float32
                eng(vol[:, :, :112, :224, :224].contiguous(), model)
                s32, o = timed(lambda: eng(vol, model))
                extras["fp32"] = job_record(n_win, 1, volume, s32, dtype="f32 (the parity path: 1e-3 gate against the oracle)")
                del o
            except Exception as e:     # noqa: BLE001
                extras["fp32"] = {"error": f"{type(e).__name__}: {e}"}
            finally:
                model.model.compute_dtype = torch.bfloat16
        torch.cuda.empty_cache()
        for name, leg in (("c3_affinity_tta16_min", lambda: c3_affinity_tta16_leg(dev, build_model(dev, out_channels=3))),
                          ("c4_mednext_l_160_chunked", lambda: c4_chunked_leg(dev))):
            try:       # BASELINE configs[2] / configs[3] inference at their real windows
                extras[name] = leg()
            except Exception as e:     # noqa: BLE001
                extras[name] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()

    del vol
    torch.cuda.empty_cache()
    strong = None
    if world > 1:
        from pytorch_connectomics_amd.inference.slab import slab_extent, slab_predict_volume
        exc, ds = None, float("nan")
        try:
            # ONE volume for the whole job; a rank materialises only the planes its own windows read (its slab + halo):
            # rows of a seeded host stream, so the overlapping planes agree between neighbours
            ax, (lo, hi) = slab_extent(volume, eng, world, rank)
            shp = list(volume)
            shp[ax] = hi - lo
            rows = torch.rand((volume[ax], 1 + 1), generator=torch.Generator().manual_seed(7))[lo:hi, 0]
            mine = torch.rand((1,) + tuple(shp), device=dev, generator=torch.Generator(device=dev).manual_seed(1000 + lo))
            mine = (0.5 * mine + 0.5 * rows.to(dev).view([-1 if a == ax + 1 else 1 for a in range(4)])).contiguous()
            with torch.no_grad():
                slab_predict_volume(mine, eng, model, full_size=volume)
                barrier()
                t0 = time.perf_counter()
                slab_predict_volume(mine, eng, model, full_size=volume)
                torch.cuda.synchronize()
                ds = time.perf_counter() - t0
            del mine
        except Exception as e:     # noqa: BLE001
            exc = e
        if agree("strong_slab", exc):
            t = torch.tensor([ds], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            strong = job_record(n_win, 1, volume, float(t.item()), scaling="strong",
                                path=f"slab_predict_volume: one volume, windows dealt to {world} ranks in contiguous runs (counts within one: "
                                     "58 / 59 of 468 at 8 ranks; each rank holds only the planes its windows read), "
                                     "p2p boxes of partial sums to the ranks whose cells touch (RCCL send/recv), result sharded")
        torch.cuda.empty_cache()

    train = None
    if not args.no_train:
        exc = None
        try:
            train = train_leg(dev, rank, world, args, barrier)
        except Exception as e:     # noqa: BLE001
            exc = e
        if not agree("train", exc):
            train = None

    rsu = unet = None
    if rank == 0 and world == 1 and not args.no_train and not args.no_extras:
        for name, leg in (("rsunet", rsunet_leg), ("monai_unet", monai_unet_leg)):
            try:
                res = leg(dev, args)
            except Exception as e:     # noqa: BLE001 - reported in the JSON
                res = {"error": f"{type(e).__name__}: {e}"}
            if name == "rsunet":
                rsu = res
            else:
                unet = res

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(model)

    watchdog.disarm()
    if rank == 0:
        out = headline_fields()
        out.update({"roofline": roofline, "cpu_baseline": cpu, "train": train, "strong_slab": strong, "rsunet": rsu,
                    "monai_unet": unet})
        out.update(extras)
        print(json.dumps(out), flush=True)
    if world > 1:
        closing = HeadlineWatchdog(60.0, lambda: None).arm()      # the line is out: a peer that never arrives must not keep us here
        torch.distributed.destroy_process_group()
        closing.disarm()


if __name__ == "__main__":
    main()
