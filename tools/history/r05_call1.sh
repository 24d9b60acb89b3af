#!/bin/bash
# Round 5, first GPU call: (1) the additivity diagnosis (telemetry, pinned clocks, per-section cycles of the depthwise kernel),
# (2) SQ / GRBM counters of the level-0 depthwise conv and mixer, (3) MedNeXt-L forward: label table, rocprofv3 kernel table, HBM counters.
set -u
OUT=$PWD/gpurun_out/r05_call1
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python - > $OUT/stats_only_check.txt 2>&1 <<'PY'
import torch
from pytorch_connectomics_amd import hip_ops as ops
for (N, D, H, W, C) in ((2, 24, 40, 33, 32), (1, 16, 16, 16, 64), (2, 9, 14, 14, 128), (3, 30, 17, 25, 32)):
    x = torch.randn(N, D, H, W, C, device="cuda").bfloat16()
    taps, b = torch.randn(27, C, device="cuda") * 0.2, torch.randn(C, device="cuda")
    for v in (0, 1):
        ops.set_tuning("dwconv_mfma_variant", v)
        y, st = ops.dwconv3d(x, taps, b, K=3)
        y2, st2 = ops.dwconv3d(x, taps, b, K=3, store=False)
        print((N, D, H, W, C), "variant", v, "stats-only pass bit-identical:", y2 is None and torch.equal(st, st2), tuple(st.shape))
ops.set_tuning("dwconv_mfma_variant", 0)
PY
cat $OUT/stats_only_check.txt
timeout 600 python tools/r05_additivity.py phases telemetry clocks > $OUT/additivity.txt 2>&1
tail -5 $OUT/additivity.txt
timeout 500 bash tools/r04_pmc.sh r05_l0 dw0 mix0 copy0 > $OUT/pmc_l0.txt 2>&1
cp gpurun_out/pmc_r05_l0/summary.txt $OUT/pmc_l0_summary.txt 2>/dev/null
rm -rf gpurun_out/pmc_r05_l0/pmc1 gpurun_out/pmc_r05_l0/pmc2 gpurun_out/pmc_r05_l0/pmc3
timeout 300 python tools/r05_l_forward.py 3 2 > $OUT/l_forward_labels.txt 2>&1
head -30 $OUT/l_forward_labels.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/l_trace -o l -- python tools/r05_l_forward.py 3 2 --no-table > $OUT/l_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/l_fetch -o l -- python tools/r05_l_forward.py 1 2 --no-table > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/l_write -o l -- python tools/r05_l_forward.py 1 2 --no-table > /dev/null 2>&1
S=$(find $OUT/l_trace -name "*kernel_stats.csv" | head -1); cp $S $OUT/l_kernel_stats.csv
python - $OUT <<'PY' > $OUT/l_hbm_counters.csv
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
for sub, ctr in (("l_fetch", "FETCH_SIZE"), ("l_write", "WRITE_SIZE")):
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == ctr:
                agg[row["Kernel_Name"]][ctr].append(float(row["Counter_Value"]))
print("kernel,launches,fetch_MB_x2_per_launch,write_MB_per_launch,total_GB_all_launches")
rows = []
for k, d in agg.items():
    n = max(len(d["FETCH_SIZE"]), len(d["WRITE_SIZE"]), 1)
    f = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1) * 2 / 1024      # KB -> MB, doubled (gfx950 guide)
    w = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1) / 1024
    rows.append((k, n, f, w, (f + w) * n / 1e3))
for k, n, f, w, t in sorted(rows, key=lambda r: -r[4]):
    print(f'"{k[:110]}",{n},{f:.1f},{w:.1f},{t:.3f}')
PY
rm -rf $OUT/l_trace $OUT/l_fetch $OUT/l_write
head -12 $OUT/l_kernel_stats.csv
head -8 $OUT/l_hbm_counters.csv
ls -la $OUT
