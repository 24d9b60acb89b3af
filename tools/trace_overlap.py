"""Concurrency summary of a rocprofv3 kernel trace (…_kernel_trace.csv): wall time covered by >= 1 and by >= 2 kernels,
per-queue busy time, and a short timeline excerpt -- the evidence that the window batches of the sliding-window engine
overlap on their HIP streams (profiles/r03_overlap.txt)."""
import csv
import sys
from collections import defaultdict


def main(path, excerpt=60):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"),
                         r.get("Stream_Id", "?")))
    rows.sort()
    if not rows:
        print("no kernels"); return
    ev = []
    for s, e, *_ in rows:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    depth = 0; last = ev[0][0]; cover = defaultdict(int)
    for t, d in ev:
        cover[min(depth, 3)] += t - last
        last = t; depth += d
    span = rows[-1][1] - rows[0][0]
    busy = defaultdict(int)
    for s, e, _, q, st in rows:
        busy[(q, st)] += e - s
    print(f"kernels {len(rows)}  span {span / 1e6:.3f} ms  sum of kernel durations {sum(e - s for s, e, *_ in rows) / 1e6:.3f} ms")
    for k in sorted(cover):
        lab = {0: "idle", 1: "exactly 1 kernel", 2: "exactly 2 kernels", 3: ">= 3 kernels"}[k]
        print(f"  {lab:18s} {cover[k] / 1e6:9.3f} ms  {100.0 * cover[k] / max(span, 1):5.1f} %")
    for k, v in sorted(busy.items()):
        print(f"  queue {k[0]} stream {k[1]}: busy {v / 1e6:.3f} ms")
    mid = len(rows) // 2
    t0 = rows[mid][0]
    print(f"timeline excerpt ({excerpt} kernels from the middle; us relative to the first one shown):")
    for s, e, name, q, st in rows[mid:mid + excerpt]:
        short = name.split("(")[0][-60:]
        print(f"  q{q:>3s} s{st:>3s} {(s - t0) / 1e3:10.1f} -> {(e - t0) / 1e3:10.1f}  {short}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
