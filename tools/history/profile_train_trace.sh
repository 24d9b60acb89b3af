#!/bin/bash
# rocprofv3 kernel trace + stats of the MedNeXt-S training probe (tools/train_probe.py) -> gpurun_out/prof_train/kernel_stats.csv
set -u
OUT=$PWD/gpurun_out/prof_train
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o train -- python tools/train_probe.py > $OUT/train.log 2>&1
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cp $S $OUT/kernel_stats.csv
rm -rf $OUT/trace
grep -a ms_per_step $OUT/train.log | head -2
python - <<'PY'
import csv, re
rows = list(csv.DictReader(open("gpurun_out/prof_train/kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel ms per step (9 steps): {tot / 9e6:.2f}")
for r in rows[:int(__import__('os').environ.get('TOP', '40'))]:
    name = re.sub(r"[(].*", "", r["Name"])[-80:]
    print(f"{float(r['TotalDurationNs']) / 9e6:7.3f} ms/step  calls/step {int(r['Calls']) / 9:6.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {name}")
PY
