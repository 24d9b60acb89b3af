"""Build libpytc_hip.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

    python -m pytorch_connectomics_amd.csrc.build [--force]

hipcc cross-compiles without a GPU.  The .so lands in pytorch_connectomics_amd/lib/ (git-ignored,
but it travels to the GPU box with the gpurun snapshot).
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent
PKG = CSRC.parent
LIB_DIR = PKG / "lib"
LIB = LIB_DIR / "libpytc_hip.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result"]


NO_SPILL_KERNELS = {"dwconv_kernels.hip": ("dwconv3d_k3_march_kernel", "dw_wgrad_march_kernel")}


def _check_no_spills(fname: str, remarks: str, patterns) -> None:
    """Parse hipcc's kernel-resource-usage remarks: every kernel whose mangled name contains one of `patterns` must report
    `VGPRs Spill: 0` and `ScratchSize [bytes/lane]: 0`."""
    import re
    name = None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            continue
        m = re.search(r"(VGPRs Spill|ScratchSize \[bytes/lane\]): (\d+)", line)
        if m and name and any(p in name for p in patterns) and int(m.group(2)) != 0:
            raise RuntimeError(f"{fname}: kernel {name} reports {m.group(1)} = {m.group(2)}; the asm-load / counted-wait kernels "
                               "must be spill-free (lower the occupancy hint of that instantiation)")


def _sources():
    return sorted(CSRC.glob("*.hip"))


def _digest() -> str:
    h = hashlib.sha256()
    for p in _sources() + sorted(CSRC.glob("*.h")) + [PKG.parent / "include" / "pytc_hip.h"]:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> Path:
    LIB_DIR.mkdir(exist_ok=True)
    obj_dir = LIB_DIR / "obj"
    obj_dir.mkdir(exist_ok=True)
    stamp = LIB_DIR / "build.sha256"
    dig = _digest()
    if not force and LIB.exists() and stamp.exists() and stamp.read_text().strip() == dig:
        build_h5(verbose=verbose)
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def compile_one(src: Path) -> Path:
        obj = obj_dir / (src.stem + ".o")
        cmd = [hipcc, *FLAGS, "-c", str(src), "-o", str(obj)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        if src.name in NO_SPILL_KERNELS:
            # kernels that issue their global loads from inline asm and wait on counted s_waitcnt: a register the compiler
            # spills while such a load is still writing it is corrupted (seen: garbage output of a 31-spill variant), so a
            # spill in one of them is a BUILD error, not a performance note
            out = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], check=True, capture_output=True, text=True).stderr
            _check_no_spills(src.name, out, NO_SPILL_KERNELS[src.name])
        else:
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, _sources()))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(LIB)]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    stamp.write_text(dig)
    build_h5(verbose=verbose)
    return LIB


H5_LIB = LIB_DIR / "libpytc_h5.so"
H5_SRC = CSRC / "host" / "h5io.c"


def build_h5(verbose: bool = True):
    """libpytc_h5.so: the C shim over the image's HDF5 C library (csrc/host/h5io.c).  Optional: when no hdf5.h / libhdf5 is
    found the engine keeps its .npy artifact layout (utils/h5lite.available() is False)."""
    for root in (os.environ.get("HDF5_ROOT"), "/opt/conda", "/usr"):
        if root and (Path(root) / "include" / "hdf5.h").exists() and list((Path(root) / "lib").glob("libhdf5.so*")):
            break
    else:
        if verbose:
            print("[build] HDF5 headers not found: libpytc_h5.so skipped")
        return None
    if H5_LIB.exists() and H5_LIB.stat().st_mtime >= H5_SRC.stat().st_mtime:
        return H5_LIB
    LIB_DIR.mkdir(exist_ok=True)
    cmd = ["gcc", "-O2", "-fPIC", "-shared", f"-I{root}/include", str(H5_SRC), f"-L{root}/lib", "-lhdf5",
           f"-Wl,-rpath,{root}/lib", "-o", str(H5_LIB)]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return H5_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
