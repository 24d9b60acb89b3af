"""Three MedNeXt-S bf16 training steps (4 x 112^3) for profiling; same step as bench.py's training leg."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench

class A: steps = 3; warmup = 1; train_batch = 4; train_steps = 6; no_roofline = True
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
print(bench.train_leg(dev, 0, 1, A, torch.cuda.synchronize))
if "--ops" in sys.argv:
    from pytorch_connectomics_amd import hip_ops as ops
    with ops.profiled() as prof:
        A.train_steps = 2
        bench.train_leg(dev, 0, 1, A, torch.cuda.synchronize)
    summ = prof.summary()
    tot = sum(v["ms"] for v in summ.values())
    print(f"total kernel ms per step (3 warm-up + 2 timed steps profiled): {tot / 5:.2f}")
    for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get("PROBE_TOP", "45"))]:
        print(f"  {k:34s} launches={v['launches'] // 5:4d} ms={v['ms'] / 5:8.3f}/step avg_us={v['ms'] / v['launches'] * 1e3:8.1f} "
              f"GB/s={v['bytes'] / max(v['ms'], 1e-9) / 1e6:8.1f}")
