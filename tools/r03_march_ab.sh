set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
python tools/exp_r03_march_tile.py 2>&1 | grep -v amdgpu.ids
for c in FETCH_SIZE WRITE_SIZE; do
  OUT=/tmp/pmc_$c; rm -rf $OUT
  rocprofv3 --pmc $c --output-format csv -d $OUT -o t -- python tools/exp_r03_march_tile.py 2 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "$c" and "march" in r["Kernel_Name"]:
            key = ("8x16" if "Li16ELi512" in r["Kernel_Name"] or "16, 512" in r["Kernel_Name"] else "8x8", r["Grid_Size"])
            acc[key].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("$c", k, "launches", len(v), "mean KB", round(sum(v) / len(v), 1))
PY
done
for k in 0 1; do
  PYTC_TUNING=dwconv_march_tx16=$k python - <<PY
import subprocess, json, os, sys
sys.path.insert(0, ".")
from pytorch_connectomics_amd import _native as nat
nat.check(nat.lib().pytc_set_tuning(b"dwconv_march_tx16", $k), "set")
import bench, torch, time
dev = torch.device("cuda", 0)
model = bench.build_model(dev); eng = bench.make_engine()
vol = torch.rand((1, 1) + bench.VOLUME, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
with torch.no_grad():
    eng(vol, model); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4): eng(vol, model)
    torch.cuda.synchronize()
print("tx16=$k  ms per 8 windows", round((time.perf_counter() - t0) / 4 / (467 / 8 + 0.3) * 1e3, 3))
PY
done
