cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/prof_c4
rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c4 -- python tools/r06_lazy_streams.py 4 > $OUT/run.log 2>&1
F=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY' > gpurun_out/r06_c4_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms (3 chunks: 1 warm-up + 2 timed; + model warm-up)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print(f"{float(r['TotalDurationNs']) / 1e6:9.2f} ms {int(r['Calls']):6d} x {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:110]}")
PY
head -45 gpurun_out/r06_c4_kernel_stats.txt
rm -rf $OUT/trace
