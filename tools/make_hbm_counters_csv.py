"""rocprofv3 PMC passes of tools/profile_bench.sh -> the committed profiles/rNN_bench_hbm_counters.csv that bench.py reads for
`roofline.traffic`: per kernel the number of profiled launches and the mean FETCH_SIZE / WRITE_SIZE (KB, as reported; bench.py
doubles FETCH_SIZE as MI355X_MICROARCH.md prescribes for gfx950).

    python tools/make_hbm_counters_csv.py gpurun_out/prof_bench profiles/r02_bench_hbm_counters.csv"""
import csv
import glob
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]


def collect(sub, counter):
    agg = defaultdict(list)
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return agg


fetch, write = collect("pmc_fetch", "FETCH_SIZE"), collect("pmc_write", "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write), key=lambda k: -(sum(fetch.get(k, [0])) + sum(write.get(k, [0])))):
    f, w = fetch.get(k, []), write.get(k, [])
    rows.append((k, max(len(f), len(w)), sum(f) / len(f) if f else float("nan"), sum(w) / len(w) if w else float("nan")))
with open(out, "w", newline="") as fh:
    wr = csv.writer(fh)
    wr.writerow(["kernel", "launches", "FETCH_SIZE_KB_mean", "WRITE_SIZE_KB_mean"])
    for k, n, f, w in rows:
        wr.writerow([k, n, f"{f:.1f}", f"{w:.1f}"])
print(f"wrote {out}: {len(rows)} kernels")
