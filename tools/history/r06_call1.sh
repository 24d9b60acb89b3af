#!/bin/bash
# round 6, GPU call 1: baseline of the tree on today's box (headline + extras), and the profile of the C3 leg
set -u
OUT=$PWD/gpurun_out/r06_call1
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-train --no-live-pmc > $OUT/bench.log 2>&1
grep "^{" $OUT/bench.log | tail -1 > $OUT/bench_line.json
python - $OUT/bench_line.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("ms/8win", d.get("ms_per_8_windows"), "value", d.get("value"))
for k in ("tta8_mean", "c3_affinity_tta16_min", "c4_mednext_l_160_chunked", "cube448", "fp32"):
    v = d.get(k) or (d.get("extras") or {}).get(k)
    if v: print(k, {kk: v[kk] for kk in v if kk in ("value", "seconds", "error", "windows", "views")})
PY
timeout 300 python tools/r06_c3_leg.py --host > $OUT/c3_host.txt 2>&1; head -70 $OUT/c3_host.txt | cut -c1-180
PYTC_SW_STREAMS=1 timeout 300 python tools/r06_c3_leg.py --twice > $OUT/c3_1stream.txt 2>&1; tail -1 $OUT/c3_1stream.txt | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_trace -o c3 -- python tools/r06_c3_leg.py > $OUT/c3_trace.log 2>&1
S=$(find $OUT/c3_trace -name "*kernel_stats.csv" | head -1); cp $S $OUT/c3_tta16_kernel_stats.csv
F=$(find $OUT/c3_trace -name "*kernel_trace.csv" | head -1)
python tools/trace_overlap.py $F 40 > $OUT/c3_overlap.txt 2>&1
rm -rf $OUT/c3_trace
head -25 $OUT/c3_tta16_kernel_stats.csv | cut -c1-200
head -14 $OUT/c3_overlap.txt
