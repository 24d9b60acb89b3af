"""Round-3 probe: which concurrently running kernel corrupts the deep-level depthwise conv?  One quiet forward is recorded as
a list of (op, args); the victim op is replayed in a loop on one HIP stream while ONE other op of the forward is replayed in
a loop on a second stream, and the victim's outputs are compared with its quiet result."""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

CALLS = None
NAMES = ("dwconv3d", "groupnorm_finalize", "pw_mlp", "pw_mlp_head", "pw_mlp_stemres", "stem_dwconv3d", "pw_conv")


def wrap(name):
    orig = getattr(ops, name)

    def f(*a, **k):
        out = orig(*a, **k)
        if CALLS is not None:
            CALLS.append((name, orig, a, k, out))
        return out
    setattr(ops, name, f)


def first(out):
    return out[0] if isinstance(out, tuple) else out


def desc(c):
    shp = [tuple(t.shape) for t in c[2] if isinstance(t, torch.Tensor)][:1]
    extra = {k: v for k, v in c[3].items() if k in ("K", "stride", "transposed", "c_in", "c_hid", "c_out")}
    return f"{c[0]} {shp} {extra}"


def main():
    global CALLS
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    for n in NAMES:
        wrap(n)
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.rand((8, 112, 112, 112, 1), device=dev, generator=g)
    with torch.no_grad():
        model.model.forward_cl(x[:1]); torch.cuda.synchronize()
        CALLS = []
        model.model.forward_cl(x); torch.cuda.synchronize()
        calls, CALLS = CALLS, None
        victims = [int(v) for v in os.environ.get("VICTIMS", "30,40").split(",") if v != "all"]
        if os.environ.get("VICTIMS") == "all":
            victims, dd = [], set()
            for i, c in enumerate(calls):
                if desc(c) not in dd:
                    dd.add(desc(c)); victims.append(i)
        only_aggr = [int(v) for v in os.environ.get("AGGR", "").split(",") if v]
        for k, v in [kv.split("=") for kv in os.environ.get("KNOBS", "").split(",") if kv]:
            from pytorch_connectomics_amd import _native as nat
            nat.check(nat.lib().pytc_set_tuning(k.encode(), int(v)), "set_tuning")
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        reps_v = int(os.environ.get("REPS_V", "60"))
        seen = set()
        for vi in victims:
            vc = calls[vi]
            ref = first(vc[1](*vc[2], **vc[3])).clone()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); vc[1](*vc[2], **vc[3]); e1.record(); torch.cuda.synchronize()
            reps_v = max(10, min(600, int(float(os.environ.get("V_MS", "12")) / max(e0.elapsed_time(e1), 0.01))))
            print("victim", vi, desc(vc))
            for ai, ac in enumerate(calls):
                d = desc(ac)
                if only_aggr and ai not in only_aggr:
                    continue
                if d in seen and os.environ.get("ALL") != "1":
                    continue
                seen.add(d)
                # aggressor duration -> repetitions for ~30 ms of overlap
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ac[1](*ac[2], **ac[3]); e1.record(); torch.cuda.synchronize()
                ms = max(e0.elapsed_time(e1), 0.01)
                reps_a = max(3, min(400, int(25.0 / ms)))
                bad = torch.zeros((), dtype=torch.int64, device=dev)
                sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(sa):
                    for _ in range(reps_a):
                        ac[1](*ac[2], **ac[3])
                with torch.cuda.stream(sb):
                    for _ in range(reps_v):
                        o = first(vc[1](*vc[2], **vc[3]))
                        bad += (o != ref).sum()
                torch.cuda.synchronize()
                nb = int(bad.item())
                print(f"   aggressor {ai:3d} {d:90s} {ms:7.3f} ms x{reps_a:3d}  victim mismatching elements: {nb}" + ("   <<<<<<" if nb else ""))
            seen.clear()


if __name__ == "__main__":
    main()
