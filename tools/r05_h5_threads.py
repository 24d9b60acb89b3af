"""Thread sweep of the parallel deflate chunk writer (csrc/host/h5io.c) on the host of the GPU box: one 7 x 320^3 fp32 chunk of sigmoid-like
values (0.92 GB, near-incompressible), gzip level 4, HDF5 chunks (7, 64, 64, 64), as bench.py's C4 leg writes it.  No GPU work."""
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, os.getcwd())
from pytorch_connectomics_amd.inference.artifact import write_prediction_artifact  # noqa: E402
from pytorch_connectomics_amd.utils import h5lite  # noqa: E402

rng = np.random.default_rng(0)
host = (1.0 / (1.0 + np.exp(-rng.standard_normal((7, 320, 320, 320), dtype=np.float32) * 3))).astype(np.float32)
print(f"host cores {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))}, {host.nbytes / 1e9:.2f} GB", flush=True)
# the floor any writer meets: the compressed bytes (0.83 GB) through plain write() calls of chunk size into the same directory
blob = os.urandom(6_650_000)
for rnd in range(2):
    with tempfile.TemporaryDirectory() as td:
        t0 = time.perf_counter()
        fd = os.open(str(Path(td) / "raw.bin"), os.O_WRONLY | os.O_CREAT)
        for _ in range(125):
            os.write(fd, blob)
        os.close(fd)
        dt = time.perf_counter() - t0
    print(f"plain write() of 125 x 6.65 MB into {tempfile.gettempdir()}: {dt:6.2f} s  {125 * 6.65 / dt:7.1f} MB/s", flush=True)
for rnd in range(2):
    for th in [int(v) for v in (sys.argv[1:] or [16, 32, 48, 64, 96, 128, 192, 256])]:
        os.environ["PYTC_H5_THREADS"] = str(th)
        with tempfile.TemporaryDirectory() as td:
            t0 = time.perf_counter()
            write_prediction_artifact(Path(td) / "c.h5", host, compression="gzip", chunks=(7, 64, 64, 64))
            dt = time.perf_counter() - t0
            fsz = (Path(td) / "c.h5").stat().st_size
        st = h5lite.last_parallel_write_stats()
        print(f"round {rnd} threads {th:4d} ({h5lite.write_threads()}): {dt:6.2f} s  {host.nbytes / 1e6 / dt:7.1f} MB/s  file {fsz / 1e6:.0f} MB | writer wall "
              f"{st['wall_s']:.2f} s: deflate {st['deflate_thread_s']:.1f} thread-s ({st['deflate_thread_s'] / 125 * 1e3:.0f} ms per chunk), gather "
              f"{st['gather_thread_s']:.2f} thread-s, serialized H5Dwrite_chunk {st['h5_write_serial_s']:.2f} s", flush=True)
