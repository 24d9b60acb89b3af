"""Summarise rocprofv3 (rocpd sqlite) outputs written by tools/history/prof_kernels.sh: per-kernel average
duration from the kernel trace and per-kernel mean counter values from the PMC passes."""
import glob
import re
import sqlite3
import sys
from collections import defaultdict

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "pytc"


def short(name):
    m = re.search(r"pytc\d*(\w+?)(I[\w]+E)?Ev", name)
    if name.startswith("_ZN4pytc"):
        body = name[len("_ZN4pytc"):]
        m = re.match(r"(\d+)", body)
        n = int(m.group(1))
        fn = body[len(m.group(1)):len(m.group(1)) + n]
        rest = body[len(m.group(1)) + n:]
        targs = rest[:rest.find("Ev")] if "Ev" in rest else ""
        targs = targs.replace("DF16b", "bf16,").replace("Li", "").replace("E", ",").strip("I,")
        return f"{fn}<{targs}>"
    return name.split("(")[0][-60:]


for f in sorted(glob.glob(f"{root}/trace/*.db")):
    c = sqlite3.connect(f)
    print("== kernel trace:", f)
    rows = c.execute("select name,total_calls,average,percentage from top_kernels").fetchall()
    for name, calls, avg, pct in rows[:14]:
        if pat in name:
            print(f"  {short(name):60s} calls={calls:5d} avg_us={avg:10.1f} pct={pct:5.1f}")
for f in sorted(glob.glob(f"{root}/pmc*/*.db")):
    c = sqlite3.connect(f)
    agg = defaultdict(lambda: defaultdict(list))
    q = "select kernel_name, grid_size, counter_name, value, duration from counters_collection"
    for name, grid, cname, val, dur in c.execute(q):
        if pat in name:
            agg[(short(name), grid)][cname].append(val)
            agg[(short(name), grid)]["_dur_us"].append(dur / 1e3)
    print("== counters:", f)
    for (k, grid), cs in sorted(agg.items(), key=lambda kv: -kv[0][1])[:12]:
        vals = {cn: (sum(v) / len(v)) for cn, v in cs.items()}
        txt = " ".join(f"{cn}={v:.4g}" for cn, v in sorted(vals.items()))
        print(f"  {k} grid={grid}: {txt}")
