"""Instruction mix of the loops of one kernel in a hipcc -S listing (gfx950): for every backward branch the instructions between
its target label and the branch are counted by class.  The per-step issue budget of the march kernels comes from here.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o k.s <file.hip> -I pytorch_connectomics_amd/csrc -I include
    python tools/isa_loop_mix.py k.s <substring of the mangled kernel name> [min loop length]
"""
import collections
import re
import sys


def classify(op: str) -> str:
    if op.startswith("v_mfma") or op.startswith("v_smfmac"): return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "lds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "lds_write"
    if op.startswith("ds_"): return "lds_other"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"): return "vmem_load"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store"): return "vmem_store"
    if op.startswith("global_atomic") or op.startswith("buffer_atomic"): return "vmem_atomic"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_cvt"): return "valu_cvt"
    if op.startswith("v_exp") or op.startswith("v_rcp") or op.startswith("v_log") or op.startswith("v_rsq") or op.startswith("v_sqrt"): return "valu_trans"
    if op.startswith("v_accvgpr"): return "valu_acc_mov"
    if op.startswith("v_"): return "valu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().splitlines()
    start = None
    for i, ln in enumerate(lines):
        if re.match(r"^_Z\S*:", ln) and pat in ln:
            start = i
            break
    if start is None:
        sys.exit(f"no kernel matching {pat}")
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    print(lines[start])
    body = lines[start:end + 1]
    labels = {m.group(1): i for i, ln in enumerate(body) if (m := re.match(r"^(\.LBB\S+):", ln))}
    ins = [(i, ln.split()[0], ln) for i, ln in enumerate(body) if ln.startswith("\t") and not ln.strip().startswith((".", ";"))]
    total = collections.Counter(classify(op) for _, op, _ in ins)
    print("whole kernel:", dict(total), "=", sum(total.values()))
    for i, op, ln in ins:
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = ln.split()[1]
            if tgt in labels and labels[tgt] < i:
                inner = [(j, o) for j, o, _ in ins if labels[tgt] < j <= i]
                if len(inner) < min_len:
                    continue
                c = collections.Counter(classify(o) for _, o in inner)
                print(f"loop {tgt} .. line {i}: {len(inner)} instructions:", dict(sorted(c.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
