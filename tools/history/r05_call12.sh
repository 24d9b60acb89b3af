#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r05_call12
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mednext.py -x -q -k "same_bits or merged" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
timeout 300 python tools/r05_l_forward.py 3 2 > $OUT/l_forward_labels.txt 2>&1; head -40 $OUT/l_forward_labels.txt | cut -c1-220
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
    f = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1) * 2 / 1024
    w = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1) / 1024
    rows.append((k, n, f, w, (f + w) * n / 1e3))
for k, n, f, w, t in sorted(rows, key=lambda r: -r[4]):
    print(f'"{k[:110]}",{n},{f:.1f},{w:.1f},{t:.3f}')
PY
rm -rf $OUT/l_trace $OUT/l_fetch $OUT/l_write
bash tools/r05_ab_r04.sh 2>&1 | tail -8
