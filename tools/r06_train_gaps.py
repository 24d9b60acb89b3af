"""Gaps between consecutive kernels of a one-stream rocprofv3 kernel trace (training step): where the GPU idles and after which kernel."""
import csv
import sys
from collections import defaultdict


def short(n):
    return n.split("(")[0].replace("void ", "").replace("pytc::", "")[-48:]


def main(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # steps: split at the optimizer kernel
    ends = [i for i, r in enumerate(rows) if "adamw_multi" in r[2]]
    if len(ends) < 3:
        print("fewer than 3 steps in the trace"); return
    a, b = ends[-3] + 1, ends[-1] + 1          # the last two whole steps
    seg = rows[a:b]
    span = seg[-1][1] - seg[0][0]
    ksum = sum(e - s for s, e, _ in seg)
    gaps = [(seg[i + 1][0] - seg[i][1], seg[i][2], seg[i + 1][2]) for i in range(len(seg) - 1)]
    gsum = sum(max(g, 0) for g, *_ in gaps)
    print(f"last 2 steps: {len(seg)} kernels, span {span / 2e6:.3f} ms per step, kernels {ksum / 2e6:.3f}, gaps {gsum / 2e6:.3f} ms per step")
    hist = defaultdict(lambda: [0, 0])
    for g, *_ in gaps:
        k = 0 if g < 1000 else 1 if g < 2000 else 2 if g < 4000 else 3 if g < 8000 else 4 if g < 20000 else 5
        hist[k][0] += 1; hist[k][1] += max(g, 0)
    for k, lab in enumerate(["< 1 us", "1-2 us", "2-4 us", "4-8 us", "8-20 us", ">= 20 us"]):
        print(f"  gaps {lab:9s}: {hist[k][0] / 2:7.1f} per step, {hist[k][1] / 2e6:7.3f} ms per step")
    by = defaultdict(lambda: [0, 0])
    for g, p, n in gaps:
        by[short(p) + "  ->  " + short(n)][0] += 1
        by[short(p) + "  ->  " + short(n)][1] += max(g, 0)
    print("largest gap totals by (previous -> next) kernel pair:")
    for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {t / 2e3:8.1f} us per step  {c / 2:6.1f} x  avg {t / c / 1e3:6.1f} us   {k}")


if __name__ == "__main__":
    main(sys.argv[1])


def sequence(path, out):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "adamw_multi" in r[2]]
    a, b = ends[-2] + 1, ends[-1] + 1
    with open(out, "w") as f:
        for s, e, n in rows[a:b]:
            f.write(f"{(s - rows[a][0]) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  {n[:150]}\n")


if __name__ == "__main__" and len(sys.argv) > 2:
    sequence(sys.argv[1], sys.argv[2])
