"""Deep-level weight-gradient launches of the MedNeXt-S training step (4 x 112^3: 1372 / 10976 / 87808 rows at levels 4 / 3 / 2) in
isolation; run under rocprofv3 (tools/r04_prof_generic.sh) for true kernel durations: the event loop is host-bound below ~15 us."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from pytorch_connectomics_amd import _native as nat  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
for k in sys.argv[1:]:
    if "=" in k:
        ops.set_tuning(*k.split("=")[:1], int(k.split("=")[1]))
N = 4
for rows, cin, cout, act in ((343, 512, 1024, 0), (343, 1024, 512, 1), (2744, 256, 512, 0), (2744, 512, 256, 1), (2744, 512, 1024, 0),
                            (21952, 128, 256, 0), (21952, 256, 128, 1)):
    x = torch.randn(N, rows, cin, device=dev).to(bf)
    dy = torch.randn(N, rows, cout, device=dev).to(bf)
    for rep in range(6):
        ops.pw_wgrad(x, dy, N=N, rows_per_sample=rows, c_in=cin, c_out=cout, x_act=nat.ACT_GELU if act else nat.ACT_NONE)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for rep in range(10):
        ops.pw_wgrad(x, dy, N=N, rows_per_sample=rows, c_in=cin, c_out=cout, x_act=nat.ACT_GELU if act else nat.ACT_NONE)
    e.record()
    torch.cuda.synchronize()
    print(f"pw_wgrad rows {N * rows:6d} {cin:4d}->{cout:4d} act={act}: {s.elapsed_time(e) * 100:7.1f} us per call (events)", flush=True)
