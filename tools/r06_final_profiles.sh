#!/bin/bash
# round 6: the driver's bench command on the current tree, then the profile passes of tools/profile_r06.sh (bench with one / three window
# streams, training step: kernel trace + stats, FETCH / WRITE counters)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_cmd.log 2>&1
grep "^{" gpurun_out/r06_bench_driver_cmd.log | tail -1 > gpurun_out/r06_bench_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_line.json"))
print("ms/8win", d.get("ms_per_8_windows"), "value", d.get("value"), "roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "traffic")})
print("whole_step", d["roofline"].get("whole_step"))
print("train", {k: d["train"].get(k) for k in ("value", "ms_per_step")} if d.get("train") else None)
for k in ("tta8", "c3_affinity_tta16_min", "c4_mednext_l_160_chunked", "cube448", "fp32"):
    v = d.get(k)
    if v: print(k, {kk: v[kk] for kk in v if kk in ("seconds", "passes_seconds", "window_voxels_per_s", "error")})
print("monai", (d.get("monai_unet") or {}).get("train_ms_per_step"), "rsunet", (d.get("rsunet") or {}).get("train_ms_per_step"))
print("cpu_baseline", d.get("cpu_baseline"))
PY
bash tools/profile_r06.sh > gpurun_out/prof_r06_run.log 2>&1
tail -45 gpurun_out/prof_r06_run.log
