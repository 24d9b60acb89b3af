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


# per-file extra flags (packed-fp32 VALU results were observed corrupted when kernels of two HIP
# streams share a SIMD, so the compiler's SLP vectoriser -- the only source of v_pk_*_f32 in compiler-generated code -- is off)
_NO_SLP = os.environ.get("PYTC_NO_SLP_FILES", "dwconv_kernels.hip,dwconv_mfma_kernels.hip,train_kernels.hip,rsunet_train_kernels.hip,"
                         "conv3d_strided_kernels.hip,loss_optim_kernels.hip,volume_kernels.hip")
EXTRA_FLAGS = {p.name: ["-fno-slp-vectorize"] for p in Path(__file__).resolve().parent.glob("*.hip")
               if _NO_SLP == "all" or p.name in _NO_SLP.split(",")}

# (also the kernels of csrc/asm_check.py's build-time walk: asm-issued loads, counted waits)
NO_SPILL_KERNELS = {"dwconv_kernels.hip": ("dwconv3d_k3_march_kernel", "dw_wgrad_march_kernel"),
                    "dwconv_mfma_kernels.hip": ("dwconv3d_k3_mfma_kernel",)}


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


def asm_listing(src: Path, hipcc: str = "/opt/rocm/bin/hipcc") -> str:
    """gfx950 assembly listing of one source with the flags the object is built with (a second device-only compile, ~2 s)."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / (src.stem + ".s")
        subprocess.run([hipcc, *FLAGS, *EXTRA_FLAGS.get(src.name, []), "-S", "--cuda-device-only", str(src), "-o", str(out)],
                       check=True, capture_output=True)
        return out.read_text()


def _check_asm_load_discipline(src: Path, hipcc: str) -> None:
    """csrc/asm_check.py on the listing of an asm-load / counted-wait source: no compiler instruction may touch a staged register between
    an asm-issued load and the next hand-written s_waitcnt vmcnt, on any control-flow path.  A violation is a BUILD error."""
    from . import asm_check
    text = asm_listing(src, hipcc)
    pats = NO_SPILL_KERNELS[src.name]
    if asm_check.asm_loads_in(text, pats) == 0:
        raise RuntimeError(f"{src.name}: no asm-issued register loads found in kernels {pats}: the ISA check would be checking nothing")
    bad = asm_check.check_listing(text, pats)
    if bad:
        raise RuntimeError(f"{src.name}: {len(bad)} violations of the asm-load discipline (csrc/asm_check.py), e.g.\n  " + "\n  ".join(bad[:5]))


def _check_packed_fp32_selects(obj: Path) -> None:
    """Disassemble the gfx950 code object inside `obj` and fail on a packed-fp32 VALU instruction whose LOW lane selects a
    high source dword (`v_pk_{fma,mul,add}_f32 ... op_sel:[..1..]`).  hipcc's SLP vectoriser emits that form when it packs two
    outputs that share a scalar operand; the one kernel of this library that contained it (dwconv3d_xblock_kernel) returned
    wrong values in single quarter-waves whenever an MFMA kernel of another HIP stream shared the GPU, and is exact without
    it (tools/exp_r03_aggressor.py, profiles/r03_stream_pipeline.txt).  Files that trip this check go on the no-SLP list."""
    import re
    objdump = Path(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")).resolve().parent.parent / "lib" / "llvm" / "bin" / "llvm-objdump"
    if not objdump.exists():
        objdump = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")
    if not objdump.exists():
        return
    subprocess.run([str(objdump), "--offloading", obj.name], cwd=obj.parent, check=True, capture_output=True)
    pat = re.compile(r"v_pk_(fma|mul|add)_f32 .*op_sel:\[")
    try:
        for co in obj.parent.glob(obj.name + ".*gfx950*"):
            dis = subprocess.run([str(objdump), "-d", str(co)], check=True, capture_output=True, text=True).stdout
            hits = [ln.strip() for ln in dis.splitlines() if pat.search(ln)]
            if hits:
                raise RuntimeError(f"{obj.name}: {len(hits)} packed-fp32 instructions with a low-lane high-dword select, e.g. "
                                   f"`{hits[0]}`; add the source to PYTC_NO_SLP_FILES / EXTRA_FLAGS in csrc/build.py")
    finally:
        for extra in obj.parent.glob(obj.name + ".*"):
            extra.unlink()


def _sources():
    return sorted(CSRC.glob("*.hip"))


def _digest() -> str:
    h = hashlib.sha256()
    for p in _sources() + sorted(CSRC.glob("*.h")) + [PKG.parent / "include" / "pytc_hip.h"]:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
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

    headers = sorted(CSRC.glob("*.h")) + [PKG.parent / "include" / "pytc_hip.h", Path(__file__), CSRC / "asm_check.py"]

    def object_digest(src: Path) -> str:
        """One object's inputs: its source, every header of the directory, its flags and the build-time checks themselves."""
        h = hashlib.sha256()
        for q in [src] + headers:
            h.update(q.name.encode())
            h.update(q.read_bytes())
        h.update(" ".join(FLAGS + EXTRA_FLAGS.get(src.name, [])).encode())
        return h.hexdigest()

    def compile_one(src: Path) -> Path:
        obj = obj_dir / (src.stem + ".o")
        ostamp = obj_dir / (src.stem + ".sha256")
        odig = object_digest(src)
        if not force and obj.exists() and ostamp.exists() and ostamp.read_text().strip() == odig:
            return obj                  # unchanged since it last passed the checks below (the stamp is written after them)
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        if src.name in NO_SPILL_KERNELS:
            # kernels that issue their global loads from inline asm and wait on counted s_waitcnt: a register the compiler
            # spills while such a load is still writing it is corrupted (seen: garbage output of a 31-spill variant), so a
            # spill in one of them is a BUILD error, not a performance note
            out = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], check=True, capture_output=True, text=True).stderr
            _check_no_spills(src.name, out, NO_SPILL_KERNELS[src.name])
            _check_asm_load_discipline(src, hipcc)
        else:
            subprocess.run(cmd, check=True)
        _check_packed_fp32_selects(obj)
        ostamp.write_text(odig)
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
    cmd = ["gcc", "-O2", "-fPIC", "-shared", f"-I{root}/include", str(H5_SRC), f"-L{root}/lib", "-lhdf5", "-lz", "-lpthread", "-ldl",
           f"-Wl,-rpath,{root}/lib", "-o", str(H5_LIB)]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return H5_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
