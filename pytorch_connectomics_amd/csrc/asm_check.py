"""Build-time ISA check of the asm-load / counted-wait kernels.

The z-march kernels (csrc/dwconv_kernels.hip, csrc/dwconv_mfma_kernels.hip) issue their plane loads from inline asm
(`global_load_dwordx4 %0, %1, off` into "=&v" outputs) and wait for them later with a COUNTED `s_waitcnt vmcnt(N)` statement (N = loads
requested after the awaited one: loads return in order, so the awaited one has landed once at most N operations are outstanding).  hipcc
believes the destination registers are written when the asm statement ends: nothing stops it from reading them, copying them, spilling
them or handing them to another value while the data is still on its way -- the round-4 "3 planes, 4 + 4 waves" build did exactly that
and faulted (an asm-issued load landed in a register the allocator had reused for an address).

What is checked, on the assembly listing of a translation unit (`hipcc -S --cuda-device-only`), for every VMEM load with a register
destination between `;;#ASMSTART` / `;;#ASMEND`:

    on EVERY control-flow path from the load, no instruction names one of its destination registers -- as a source, as a destination, or
    as the destination of another load -- before an asm-issued `s_waitcnt vmcnt(..)` has executed; and no path reaches `s_endpgm` first.

i.e. the compiler's code never touches a staged register in the window where only the hand-written waits know whether it has landed.
That the wait's COUNT is the right one is the kernel author's arithmetic (a wave-uniform run-time selection among vmcnt immediates that a
static walk cannot follow); it is pinned dynamically: every output of these kernels is compared with an fp64 / fp32 convolution on ragged
shapes, chunk ends included, and by the soak test.  (`check_listing(..., strict=True)` reports the stronger property -- a wait whose
count provably covers the load on every static path -- which the run-time selected waits cannot satisfy; it is informational.)

`python -m pytorch_connectomics_amd.csrc.asm_check <file.s> [kernel-name substring ...]` prints the violations and exits non-zero.
"""
from __future__ import annotations

import re
import sys
from typing import Dict, Iterable, List, Optional, Sequence, Set, Tuple

_LOAD = re.compile(r"^(global_load|buffer_load|flat_load|scratch_load)")
_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
_VMCNT = re.compile(r"vmcnt\((\d+)\)")
_LABEL = re.compile(r"^(\.L[A-Za-z0-9_$.]+):")
_KERNEL = re.compile(r"^(_Z[A-Za-z0-9_$.]+):")


class Violation(Exception):
    pass


def _regs(text: str) -> Set[Tuple[str, int]]:
    out: Set[Tuple[str, int]] = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def _parse_kernel(lines: Sequence[str]):
    """-> (instructions [(mnemonic, operand text, in_asm, source line)], label -> instruction index)"""
    ins: List[Tuple[str, str, bool, str]] = []
    labels: Dict[str, int] = {}
    in_asm = False
    for ln in lines:
        s = ln.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = _LABEL.match(ln)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not s or s.startswith((";", ".", "//")) or not ln.startswith(("\t", " ")):
            continue
        code = s.split(";", 1)[0].strip()
        if not code:
            continue
        parts = code.split(None, 1)
        ins.append((parts[0], parts[1] if len(parts) > 1 else "", in_asm, s))
    return ins, labels


def _dest_regs(mn: str, ops: str) -> Set[Tuple[str, int]]:
    """destination registers of a VMEM load with a register destination (first operand); LDS-DMA forms have none"""
    if "lds" in mn or " lds" in ops:
        return set()
    first = ops.split(",", 1)[0]
    return _regs(first)


def check_kernel(name: str, lines: Sequence[str], strict: bool = False, max_younger: int = 24) -> List[str]:
    ins, labels = _parse_kernel(lines)
    n = len(ins)
    found: List[str] = []

    def successors(i: int) -> List[int]:
        mn, ops, _, _ = ins[i]
        if mn == "s_endpgm":
            return []
        if mn == "s_branch":
            t = ops.strip()
            return [labels[t]] if t in labels else []
        if mn.startswith("s_cbranch"):
            t = ops.strip().split(",")[-1].strip()
            nxt = [i + 1] if i + 1 < n else []
            return nxt + ([labels[t]] if t in labels else [])
        return [i + 1] if i + 1 < n else []

    # hipcc structures some regions with conditional branches on a mask it has just set to a constant (`s_mov_b64 s[a:b], -1` ...
    # `s_andn2_b64 vcc, exec, s[a:b]` ... `s_cbranch_vccz L`: always taken): following the dead edge would report paths that cannot
    # execute.  The walk therefore carries the scalar pairs it has seen set to 0 / -1 and the resulting state of vcc.
    def scalar_pair(tok: str) -> Optional[str]:
        tok = tok.strip()
        return tok if re.fullmatch(r"s\[\d+:\d+\]|vcc|exec", tok) else None

    def step_consts(mn: str, ops: str, consts: Tuple[Tuple[str, int], ...], vcc: Optional[int]):
        """-> (consts, vcc) after the instruction; vcc: 0 = known zero, 1 = known equal to exec (non-zero for a live wave), None = unknown"""
        parts = [t.strip() for t in ops.split(",")]
        d = dict(consts)
        dst = scalar_pair(parts[0]) if parts else None
        if mn == "s_mov_b64" and dst and len(parts) == 2 and parts[1] in ("0", "-1"):
            if dst == "vcc":
                return tuple(sorted(d.items())), (0 if parts[1] == "0" else 1)
            d[dst] = int(parts[1])
            return tuple(sorted(d.items())), vcc
        if mn == "s_andn2_b64" and dst == "vcc" and len(parts) == 3 and parts[1] == "exec" and parts[2] in d:
            return consts, (0 if d[parts[2]] == -1 else 1)
        if mn in ("s_and_b64",) and dst == "vcc" and len(parts) == 3 and parts[1] == "exec" and parts[2] in d:
            return consts, (1 if d[parts[2]] == -1 else 0)
        # anything else that writes a tracked pair or vcc forgets it
        if dst is not None and mn.startswith("s_") and not mn.startswith(("s_cmp", "s_cbranch", "s_branch", "s_waitcnt", "s_barrier", "s_nop")):
            if dst == "vcc":
                return consts, None
            if dst in d:
                del d[dst]
                return tuple(sorted(d.items())), vcc
        if mn.startswith("v_cmp") or (mn.startswith("v_") and "vcc" in parts[:1]):
            return consts, None
        return consts, vcc

    for i, (mn, ops, in_asm, src) in enumerate(ins):
        if not (in_asm and _LOAD.match(mn)):
            continue
        dest = _dest_regs(mn, ops)
        if not dest:
            continue
        seen: Set[Tuple] = set()
        stack = [(j, 0, (), None) for j in successors(i)]
        while stack:
            pc, younger, consts, vcc = stack.pop()
            key = (pc, younger, consts, vcc)
            if key in seen:
                continue
            seen.add(key)
            pmn, pops, pasm, psrc = ins[pc]
            if pmn == "s_waitcnt":
                m = _VMCNT.search(pops)
                if m and ((not strict and pasm) or (strict and int(m.group(1)) <= younger)):
                    continue                                   # a hand-written wait has executed on this path (strict: one that covers)
            if pmn == "s_endpgm":
                found.append(f"{name}: `{src}` (instruction {i}) can reach s_endpgm before an asm-issued s_waitcnt vmcnt")
                continue
            touched = _regs(pops) & dest
            if touched:
                regs = ", ".join(f"{k}{r}" for k, r in sorted(touched))
                found.append(f"{name}: `{psrc}` (instruction {pc}) touches {regs} before any asm-issued s_waitcnt vmcnt follows "
                             f"`{src}` (instruction {i})")
                continue                                       # one report per path is enough
            if strict and _LOAD.match(pmn):
                younger = min(younger + 1, max_younger)
            nxt = successors(pc)
            if pmn in ("s_cbranch_vccz", "s_cbranch_vccnz") and vcc is not None and len(nxt) == 2:
                taken = (vcc == 0) if pmn == "s_cbranch_vccz" else (vcc == 1)
                nxt = [nxt[1]] if taken else [nxt[0]]
            consts2, vcc2 = step_consts(pmn, pops, consts, vcc)
            for nx in nxt:
                stack.append((nx, younger, consts2, vcc2))
    return sorted(set(found))


def split_kernels(text: str) -> Dict[str, List[str]]:
    out: Dict[str, List[str]] = {}
    cur: Optional[str] = None
    for ln in text.splitlines():
        m = _KERNEL.match(ln)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is not None:
            out[cur].append(ln)
            if ln.strip().startswith("s_endpgm") and ".Lfunc_end" not in ln:
                pass
            if ln.strip().startswith(".Lfunc_end"):
                cur = None
    return out


def check_listing(text: str, patterns: Iterable[str] = (), strict: bool = False) -> List[str]:
    """All violations in the kernels of a listing whose mangled names contain one of `patterns` (all kernels when empty)."""
    pats = tuple(patterns)
    found: List[str] = []
    for name, lines in split_kernels(text).items():
        if pats and not any(p in name for p in pats):
            continue
        found.extend(check_kernel(name, lines, strict))
    return found


def asm_loads_in(text: str, patterns: Iterable[str] = ()) -> int:
    """number of tracked (asm-issued, register-destination) loads: a check that finds none has checked nothing"""
    pats = tuple(patterns)
    total = 0
    for name, lines in split_kernels(text).items():
        if pats and not any(p in name for p in pats):
            continue
        ins, _ = _parse_kernel(lines)
        total += sum(1 for mn, ops, in_asm, _ in ins if in_asm and _LOAD.match(mn) and _dest_regs(mn, ops))
    return total


if __name__ == "__main__":
    listing = open(sys.argv[1]).read()
    bad = check_listing(listing, sys.argv[2:])
    print(f"{asm_loads_in(listing, sys.argv[2:])} asm-issued register loads checked, {len(bad)} violations")
    for b in bad[:40]:
        print("  " + b)
    sys.exit(1 if bad else 0)
