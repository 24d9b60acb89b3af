#!/bin/bash
# rocprofv3 kernel trace of the RSUNet training probe (run on the GPU box). Writes gpurun_out/prof_rsunet/.
set -u
OUT=$PWD/gpurun_out/prof_rsunet
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$PWD
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o rsunet -- python tools/history/rsunet_train_probe.py --steps 5 --gc freeze > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
python - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("$OUT/kernel_stats.csv", "w") as o:
    o.write("name,calls,avg_ns,total_ns,pct\n")
    for r in rows:
        o.write('"%s",%s,%s,%s,%s\n' % (r["Name"][:160], r["Calls"], r["AverageNs"], r["TotalDurationNs"], r["Percentage"]))
for r in rows[:14]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
rm -rf $OUT/trace
