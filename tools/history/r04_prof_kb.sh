#!/bin/bash
# rocprofv3 kernel statistics of tools/r04_kb.py targets (true kernel durations: the event loop of r04_kb.py is host-bound below ~15 us)
#   tools/history/r04_prof_kb.sh <name> <targets...>   -> gpurun_out/prof_<name>/stats.txt
NAME=$1; shift
OUT=$PWD/gpurun_out/prof_$NAME
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python tools/r04_kb.py "$@" > $OUT/run.log 2>&1
python - "$OUT" <<'PY' | tee $OUT/stats.txt
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[(row["Kernel_Name"][:110], row.get("Grid_Size", ""))].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if "pytc" in k or "pw_" in k:
        v2 = sorted(v)[len(v) // 4: len(v) - len(v) // 4] or v
        print(f"{k:112s} grid={g:>9s} n={len(v):4d} median_us={sorted(v)[len(v)//2]:9.1f} iqr_mean_us={sum(v2)/len(v2):9.1f}")
PY
