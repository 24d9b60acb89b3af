"""Round 5: why do memory time and instruction time ADD in the level-0 kernels?  Three measurements on one box:

  1. telemetry -- every target is launched back to back for >= SECS seconds while a side PROCESS samples the SMU's gpu_metrics table
     (socket power, gfx / memory clocks per XCD, throttle status, activity) at ~50 Hz through amdsmi; rows: launch time, median power,
     median clocks.  Targets: idle, copy_ / read / fill of the level-0 bytes, the matrix-core depthwise conv (full, without matrix
     instructions, loads + commit only, statistics only), the 32->64->32 and 64->128->32 mixers (random and zero-filled operands),
     a bf16 GEMM (matrix pipes only), an L2-resident erf loop (VALU only).
  2. the same kernels at a pinned lower clock (perf determinism), when the box lets us set one: T(f) = T_mem + N_instr / f separates
     the two terms without editing a kernel.
  3. phases -- dwconv3d_k3_mfma_kernel<3, false, 4>: shader cycles per section of a plane step, per wave (knob dwconv_mfma_probe = 4).

    python tools/r05_additivity.py [telemetry] [clocks] [phases]          -> stdout (copied to profiles/r05_additivity.txt)
"""
import json
import os
import statistics
import subprocess
import sys
import time
from pathlib import Path

SECS = float(os.environ.get("PYTC_ADD_SECS", "2.5"))


# ---------------------------------------------------------------------------------------------------------------------------------
def _sampler(path: str, period: float):
    """Side process (`--sampler <path>`; ends when <path>.stop appears): gpu_metrics through amdsmi (falls back to hwmon sysfs
    files) -> JSON lines {t, power, gfxclk[], uclk, ...}."""
    src = None
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        h = amdsmi.amdsmi_get_processor_handles()[0]
        src = "amdsmi"
    except Exception as exc:  # noqa: BLE001
        src = f"sysfs ({type(exc).__name__}: {exc})"
        h = None
    hw = None
    if h is None:
        for d in sorted(Path("/sys/class/drm").glob("card*/device/hwmon/hwmon*")):
            hw = d
            break
    keys = ("average_socket_power", "current_socket_power", "current_gfxclk", "current_gfxclks", "current_uclk", "current_socclk",
            "average_gfx_activity", "average_umc_activity", "throttle_status", "indep_throttle_status", "temperature_hotspot",
            "temperature_mem", "energy_accumulator", "accumulation_counter", "prochot_residency_acc", "ppt_residency_acc",
            "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc", "gfxclk_lock_status", "average_gfxclk_frequency",
            "average_uclk_frequency")
    first = True
    with open(path, "w") as f:
        f.write(json.dumps({"source": src}) + "\n")
        while not os.path.exists(path + ".stop"):
            t = time.time()
            row = {"t": t}
            try:
                if h is not None:
                    import amdsmi
                    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                    if first:
                        f.write(json.dumps({"all_keys": sorted(m.keys())}) + "\n")
                        first = False
                    for k in keys:
                        if k in m:
                            v = m[k]
                            row[k] = v if not isinstance(v, (list, tuple)) else [x for x in v if isinstance(x, (int, float))][:8]
                elif hw is not None:
                    for fn in ("power1_average", "power1_input", "freq1_input", "freq2_input"):
                        fp = hw / fn
                        if fp.exists():
                            row[fn] = int(fp.read_text().strip())
            except Exception as exc:  # noqa: BLE001
                row["err"] = f"{type(exc).__name__}: {exc}"
            f.write(json.dumps(row, default=str) + "\n")
            f.flush()
            dt = period - (time.time() - t)
            if dt > 0:
                time.sleep(dt)


if len(sys.argv) > 2 and sys.argv[1] == "--sampler":        # before the heavy imports: the sampler starts within a few 100 ms
    _sampler(sys.argv[2], 0.02)
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pytorch_connectomics_amd import _native as nat  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16


def _num(v):
    if isinstance(v, (int, float)):
        return float(v)
    if isinstance(v, str):
        try:
            return float(v)
        except ValueError:
            return None
    return None


def _summ(samples, t0, t1):
    """median of every numeric column over the samples inside (t0 + 0.4 s, t1): the first 0.4 s are the ramp"""
    sel = [s for s in samples if "t" in s and t0 + 0.4 <= s["t"] <= t1]
    out = {"n": len(sel)}
    if not sel:
        return out
    for k in sel[0]:
        if k == "t":
            continue
        vals = []
        for s in sel:
            v = s.get(k)
            if isinstance(v, list):
                xs = [x for x in (_num(e) for e in v) if x is not None and x < 60000]
                if xs:
                    vals.append(sum(xs) / len(xs))
            else:
                x = _num(v)
                if x is not None:
                    vals.append(x)
        if vals:
            out[k] = statistics.median(vals)
            if k in ("energy_accumulator", "accumulation_counter", "ppt_residency_acc", "prochot_residency_acc", "socket_thm_residency_acc"):
                out[k + "_delta"] = vals[-1] - vals[0]
    return out


def knob(k, v):
    nat.check(nat.lib().pytc_set_tuning(k.encode(), int(v)), "set_tuning")


def run_for(fn, secs):
    """launch fn back to back for `secs` seconds -> (us per launch, wall start, wall end)"""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    w0 = time.time()
    s.record()
    while True:
        for _ in range(40):
            fn()
        n += 40
        torch.cuda.synchronize()
        if time.time() - w0 >= secs:
            break
    e.record()
    torch.cuda.synchronize()
    w1 = time.time()
    return s.elapsed_time(e) / n * 1e3, w0, w1


def make_targets(zero=False):
    N, D, C = 8, 112, 32
    x = torch.randn(N, D, D, D, C, device=dev).to(bf)
    if zero:
        x.zero_()
    y = torch.empty_like(x)
    taps = torch.randn(27, C, device=dev) * 0.2
    bias = torch.randn(C, device=dev)
    rows = D ** 3
    T = {}
    T["copy_ 0.72 GB -> 0.72 GB"] = (lambda: y.copy_(x), 2 * x.numel() * 2)
    T["read only (sum) 0.72 GB"] = (lambda: torch.sum(x, dtype=torch.float32), x.numel() * 2)
    T["fill (zero_) 0.72 GB"] = (lambda: y.zero_(), x.numel() * 2)

    def dwk(probe, store=True):
        def f():
            knob("dwconv_mfma_probe", probe)
            ops.dwconv3d(x, taps, bias, K=3, y=y if store else None, store=store)
        return f
    T["dwconv mfma full"] = (dwk(0), 2 * x.numel() * 2)
    T["dwconv mfma, no matrix instr (probe 1)"] = (dwk(1), 2 * x.numel() * 2)
    T["dwconv mfma, loads + commit only (probe 3)"] = (dwk(3), x.numel() * 2)
    T["dwconv mfma, statistics only (no output path)"] = (dwk(0, False), x.numel() * 2)

    def mixer(cin, chid, cout):
        t = torch.randn(N, rows, cin, device=dev).to(bf)
        res = torch.randn(N, rows, cout, device=dev).to(bf)
        if zero:
            t.zero_(); res.zero_()
        ab = torch.rand(N, 2, cin, device=dev)
        w2 = ops.pw_pack_weight_paired(torch.randn(chid, cin, device=dev) / cin ** 0.5)
        w3 = ops.pw_pack_weight_paired(torch.randn(cout, chid, device=dev) / chid ** 0.5, f16=True)
        b2, b3 = torch.randn(chid, device=dev), torch.randn(cout, device=dev)
        yy = torch.empty(N, rows, cout, device=dev, dtype=bf)
        kw = dict(N=N, rows_per_sample=rows, c_in=cin, c_hid=chid, c_out=cout, y=yy)
        return (lambda: ops.pw_mlp(t, ab, w2, b2, w3, b3, res=res, res_mode=nat.RES_ADD, **kw)), N * rows * 2 * (cin + 2 * cout)
    T["mixer 32->64->32 +res"] = mixer(32, 64, 32)
    T["mixer 64->128->32 +res"] = mixer(64, 128, 32)
    a = torch.randn(8192, 8192, device=dev).to(bf)
    b = torch.randn(8192, 8192, device=dev).to(bf)
    if zero:
        a.zero_(); b.zero_()
    T["bf16 GEMM 8192^3 (hipBLASLt)"] = (lambda: torch.mm(a, b), 0)
    v = torch.randn(4 * 1024 * 1024, device=dev)            # 16 MB fp32: L2 / MALL resident, VALU + transcendental bound
    T["erf_ in place, 16 MB (cache resident)"] = (lambda: torch.special.erf(v, out=v), 0)
    return T


def telemetry(tag=""):
    path = f"/tmp/r05_tele_{os.getpid()}_{tag or 'a'}.jsonl"
    proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--sampler", path])
    for _ in range(100):                      # wait for the first samples
        if os.path.exists(path) and os.path.getsize(path) > 200:
            break
        time.sleep(0.1)
    marks = []
    w0 = time.time(); time.sleep(2.0); marks.append(("idle", None, 0, w0, time.time()))
    for zero in (False, True):
        T = make_targets(zero)
        for name, (fn, nbytes) in T.items():
            if zero and not ("mixer 32" in name or "dwconv mfma full" in name or "copy_" in name or "GEMM" in name):
                continue
            us, a, b = run_for(fn, SECS)
            marks.append((name + (" [zero-filled]" if zero else ""), us, nbytes, a, b))
            time.sleep(0.3)
        del T
        torch.cuda.empty_cache()
    knob("dwconv_mfma_probe", 0)
    open(path + ".stop", "w").close()
    proc.wait(10)
    os.unlink(path + ".stop")
    lines = [json.loads(ln) for ln in open(path)]
    print(f"telemetry source: {lines[0].get('source')}   samples: {len(lines) - 1}   period 20 ms   {SECS:.1f} s per target {tag}")
    for ln in lines[:3]:
        if "all_keys" in ln:
            print("gpu_metrics keys:", ", ".join(ln["all_keys"]))
    samples = [ln for ln in lines if "t" in ln]
    hdr = f"{'target':58s} {'us/launch':>10s} {'GB/s':>7s} {'W(avg)':>7s} {'W(cur)':>7s} {'gfx MHz':>8s} {'uclk':>6s} {'act%':>5s} {'umc%':>5s} {'thr':>6s} {'hot C':>5s} n"
    print(hdr)
    for name, us, nbytes, a, b in marks:
        s = _summ(samples, a, b)
        g = lambda k, d=float('nan'): s.get(k, d)      # noqa: E731
        gfx = g("current_gfxclks", g("current_gfxclk", g("freq1_input", float('nan'))))
        pw = g("average_socket_power", g("power1_average", float('nan')))
        print(f"{name:58s} {(us if us is not None else float('nan')):10.1f} {(nbytes / us / 1e3 if us else 0):7.0f} {pw:7.0f} "
              f"{g('current_socket_power'):7.0f} {gfx:8.0f} {g('current_uclk', g('freq2_input')):6.0f} {g('average_gfx_activity'):5.0f} "
              f"{g('average_umc_activity'):5.0f} {g('throttle_status', g('indep_throttle_status')):6.0f} {g('temperature_hotspot'):5.0f} {s['n']}"
              + (f"  dE {s['energy_accumulator_delta']:.0f}" if 'energy_accumulator_delta' in s else "")
              + (f"  ppt_res {s['ppt_residency_acc_delta']:.0f}" if 'ppt_residency_acc_delta' in s else ""))
    os.unlink(path)


def sh(cmd):
    r = subprocess.run(cmd, shell=True, capture_output=True, text=True)
    return (r.stdout + r.stderr).strip()


def clocks():
    """the same launches at a pinned lower gfx clock (perf determinism): which term of each kernel scales with 1 / f?"""
    print("power cap / limits:", sh("rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -v '^=' | head -20"))
    T = make_targets(False)
    names = ["copy_ 0.72 GB -> 0.72 GB", "dwconv mfma full", "dwconv mfma, no matrix instr (probe 1)",
             "dwconv mfma, loads + commit only (probe 3)", "dwconv mfma, statistics only (no output path)", "mixer 32->64->32 +res",
             "mixer 64->128->32 +res", "erf_ in place, 16 MB (cache resident)", "bf16 GEMM 8192^3 (hipBLASLt)"]
    table = {}
    for mhz in (0, 1700, 1300, 0):
        if mhz:
            out = sh(f"rocm-smi --setperfdeterminism {mhz} 2>&1 | tail -3")
            print(f"--setperfdeterminism {mhz}: {out}")
        else:
            print("reset:", sh("rocm-smi --resetperfdeterminism 2>&1 | tail -2"))
        time.sleep(0.5)
        for nm in names:
            us, _, _ = run_for(T[nm][0], 1.0)
            table.setdefault(nm, []).append(us)
    knob("dwconv_mfma_probe", 0)
    print(f"{'target':58s} {'default':>9s} {'1700 MHz':>9s} {'1300 MHz':>9s} {'default':>9s}   (us per launch)")
    for nm in names:
        print(f"{nm:58s} " + " ".join(f"{u:9.1f}" for u in table[nm]))


def phases():
    """section cycle counts of the matrix-core depthwise conv's plane steps (probe 4) at the three levels of an 8-window batch"""
    for (N, D, C) in ((8, 112, 32), (8, 56, 64), (8, 28, 128)):
        x = torch.randn(N, D, D, D, C, device=dev).to(bf)
        taps = torch.randn(27, C, device=dev) * 0.2
        bias = torch.randn(C, device=dev)
        y = torch.empty_like(x)
        knob("dwconv_mfma_probe", 0)
        us0, _, _ = run_for(lambda: ops.dwconv3d(x, taps, bias, K=3, y=y), 0.5)
        knob("dwconv_mfma_probe", 4)
        us4, _, _ = run_for(lambda: ops.dwconv3d(x, taps, bias, K=3, y=y), 0.5)
        _, st = ops.dwconv3d(x, taps, bias, K=3, y=y)
        torch.cuda.synchronize()
        knob("dwconv_mfma_probe", 0)
        # stats (N, slots, 2, C): [:, :, 0, cg*32 + wave*8 + k]
        v = st[:, :, 0, :].reshape(N, -1, C // 32, 4, 8).double()
        steps = v[..., 6]
        per = v[..., :6] / steps.unsqueeze(-1)
        tot = v[..., 7]
        names = ["flush+issue", "matrix instr", "wait plane", "LDS commit", "round+tile", "barrier"]
        print(f"phases {N}x{D}^3x{C}: launch {us0:.1f} us (probe 4: {us4:.1f} us); workgroups {v.shape[0] * v.shape[1] * v.shape[2]}, "
              f"steps per workgroup {float(steps.mean()):.1f}; cycles per plane step, mean over waves (min .. max of workgroup means):")
        for i, nm in enumerate(names):
            m = per[..., i]
            print(f"   {nm:14s} {float(m.mean()):8.0f}   ({float(m.mean(-1).min()):6.0f} .. {float(m.mean(-1).max()):6.0f})   "
                  f"{100 * float((v[..., i]).sum() / tot.sum()):5.1f} % of wave life")
        print(f"   step total     {float(per.sum(-1).mean()):8.0f}   wave life {float(tot.mean()):.0f} cycles = "
              f"{float(tot.mean()) / 2.4e3:.1f} us at 2.4 GHz; by wave index: " +
              ", ".join(f"w{w} wait {float(per[..., w, 2].mean()):.0f} barrier {float(per[..., w, 5].mean()):.0f}" for w in range(4)))


def phases_mix():
    """section cycles of the fused block kernel (dwconv3d_k3_mfma_kernel<3, false, 4, true, 2>) at level 0"""
    N, D, c_hid = 8, 112, 64
    x = torch.randn(N, D, D, D, 32, device=dev).to(bf)
    taps, b1 = torch.randn(27, 32, device=dev) * 0.2, torch.randn(32, device=dev)
    gamma, beta = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
    w2, b2 = (torch.randn(c_hid, 32, device=dev) / 32 ** 0.5).contiguous(), torch.randn(c_hid, device=dev) * 0.1
    w3 = ops.pw_pack_weight_paired((torch.randn(32, c_hid, device=dev) / c_hid ** 0.5).contiguous(), f16=True)
    b3 = torch.randn(32, device=dev) * 0.1
    _, st = ops.dwconv3d(x, taps, b1, K=3, store=False)
    w2n, b2n = ops.groupnorm_fold_mlp(st, float(D ** 3), gamma, beta, 1e-5, w2, b2)
    y = torch.empty_like(x)
    slots = st.shape[1]
    for residual in (True, False):
        knob("dwconv_mfma_probe", 0)
        us0, _, _ = run_for(lambda: ops.dwmix(x, taps, b1, w2n, b2n, w3, b3, c_hid=c_hid, residual=residual, y=y), 0.5)
        prof = torch.zeros(N, slots, 4, 9, device=dev)
        ptr = prof.data_ptr()
        lo, hi = ptr & 0xffffffff, ptr >> 32
        knob("dwmix_prof_lo", lo - (1 << 32) if lo >= (1 << 31) else lo)
        knob("dwmix_prof_hi", hi)
        knob("dwconv_mfma_probe", 4)
        us4, _, _ = run_for(lambda: ops.dwmix(x, taps, b1, w2n, b2n, w3, b3, c_hid=c_hid, residual=residual, y=y), 0.5)
        torch.cuda.synchronize()
        knob("dwconv_mfma_probe", 0)
        v = prof.double()
        steps = v[..., 6]
        names = ["load issue", "matrix instr", "wait plane", "LDS commit", "round+tile", "barrier", None, None, "channel mixer"]
        print(f"fused block {N}x{D}^3x32 -> {c_hid} -> 32, residual {residual}: launch {us0:.1f} us (probe 4: {us4:.1f} us); cycles per plane step, mean over waves:")
        tot = v[..., 7]
        for i, nm in enumerate(names):
            if nm is None:
                continue
            m = v[..., i] / steps
            print(f"   {nm:14s} {float(m.mean()):8.0f}   {100 * float(v[..., i].sum() / tot.sum()):5.1f} % of wave life")
        print(f"   step total     {float(((v[..., :6].sum(-1) + v[..., 8]) / steps).mean()):8.0f}   wave life {float(tot.mean()):.0f} cycles")


if __name__ == "__main__":
    what = sys.argv[1:] or ["telemetry", "clocks", "phases"]
    print(torch.cuda.get_device_name(0))
    if "phases" in what:
        phases()
    if "phases_mix" in what:
        phases_mix()
    if "telemetry" in what:
        telemetry()
    if "clocks" in what:
        clocks()
