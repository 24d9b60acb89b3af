#!/bin/bash
# rocprofv3 kernel trace of any python command, per-(kernel, grid) medians:  tools/r04_prof_generic.sh <name> <python args...>
NAME=$1; shift
OUT=$PWD/gpurun_out/prof_$NAME
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python "$@" > $OUT/run.log 2>&1
python - "$OUT" <<'PY' | tee $OUT/stats.txt
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "pytc" in k or "pw_" in k:
            agg[(k.replace("void pytc::", "")[:64], row["Grid_Size_X"], row["Grid_Size_Y"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for (k, gx, gy), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"{k:66s} grid=({gx:>7s},{gy:>4s}) n={len(v):4d} median_us={v[len(v)//2]:8.1f} min={v[0]:8.1f}")
PY
