"""Level-0 mixer with LDS-DMA prefetch (pw_mlp_dma_kernel) against the one-tile-per-wave kernel: bit identity on ragged shapes, then time
per launch at the network's level-0 shape, alternating, with a sweep of workgroups per sample; `chunk`: the chunk-streamed mixer of the wide hidden
layers against the streaming / LDS-resident kernels.  python tools/r05_mlp_dma.py [check] [time] [chunk]
(The `lds` mode that produced the last section of profiles/r05_mlp_dma_prefetch.txt -- the same prefetch inside pw_mlp_lds_kernel -- went with that code.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_connectomics_amd import _native as nat  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16


def knob(k, v):
    nat.check(nat.lib().pytc_set_tuning(k.encode(), int(v)), "set_tuning")


def operands(N, rows, chid, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    t = torch.randn(N, rows, 32, generator=g).to(bf).to(dev)
    res = torch.randn(N, rows, 32, generator=g).to(bf).to(dev)
    w2n = torch.stack([ops.pw_pack_weight_paired((torch.randn(chid, 32, generator=g) / 32 ** 0.5).to(dev)) for _ in range(N)])
    b2n = torch.randn(N, chid, generator=g).to(dev)
    w3 = ops.pw_pack_weight_paired((torch.randn(32, chid, generator=g) / chid ** 0.5).to(dev), f16=True)
    b3 = torch.randn(32, generator=g).to(dev)
    return t, res, w2n, b2n, w3, b3


def run(t, res, w2n, b2n, w3, b3, add, y=None):
    N, rows, chid = t.shape[0], t.shape[1], b2n.shape[1]
    kw = dict(N=N, rows_per_sample=rows, c_in=32, c_hid=chid, c_out=32, y=y)
    if add:
        return ops.pw_mlp(t, None, w2n, b2n, w3, b3, res=res, res_mode=nat.RES_ADD, **kw)
    return ops.pw_mlp(t, None, w2n, b2n, w3, b3, **kw)


def run_stem(t, res, w2n, b2n, w3, b3, x0, sw, sb, y=None):
    N, rows, chid = t.shape[0], t.shape[1], b2n.shape[1]
    return ops.pw_mlp_stemres(t, None, w2n, b2n, w3, b3, x0, sw, sb, N=N, rows_per_sample=rows, c_in=32, c_hid=chid, c_out=32, y=y)


def run_head(t, res, w2n, b2n, w3, b3, hw, hb, add, store_y):
    N, rows, chid = t.shape[0], t.shape[1], b2n.shape[1]
    return ops.pw_mlp_head(t, None, w2n, b2n, w3, b3, hw, hb, N=N, rows_per_sample=rows, c_in=32, c_hid=chid, c_out=32,
                           res=res if add else None, store_y=store_y)


def extras(N, rows, n_head, seed):
    g = torch.Generator(device="cpu").manual_seed(seed + 7)
    x0 = torch.randn(N, rows, generator=g).to(dev)
    sw, sb = torch.randn(32, generator=g).to(dev), torch.randn(32, generator=g).to(dev)
    hw = ops.pack_head_fragment((torch.randn(n_head, 32, generator=g) / 32 ** 0.5).to(dev))
    hb = torch.randn(n_head, generator=g).to(dev)
    return x0, sw, sb, hw, hb


def same(a, b):
    if a is None or b is None:
        return a is None and b is None
    return torch.equal(a.view(torch.int16 if a.dtype == bf else torch.int32), b.view(torch.int16 if b.dtype == bf else torch.int32))


def check_kinds():
    knob("mlp_dma_rows", 0)
    for N, rows, chid, n_head in [(1, 1, 64, 1), (2, 63, 64, 3), (3, 1000, 96, 16), (2, 4097, 128, 5), (8, 28 ** 3, 64, 3), (1, 112 ** 3, 64, 2)]:
        o = operands(N, rows, chid, seed=rows)
        x0, sw, sb, hw, hb = extras(N, rows, n_head, rows)
        got = {}
        for dma in (0, 1):
            knob("mlp_dma", dma)
            got[dma] = [run_stem(*o, x0, sw, sb)]
            for add in (False, True):
                for store_y in (False, True):
                    y, logits = run_head(*o, hw, hb if add else None, add, store_y)
                    got[dma] += [y, logits]
        torch.cuda.synchronize()
        ok = all(same(a, b) for a, b in zip(got[0], got[1]))
        print(f"stem-residual / head kinds: N {N} rows {rows} hid {chid} heads {n_head}: {'bit-identical' if ok else 'MISMATCH'}", flush=True)
        assert ok


def check():
    knob("mlp_dma_rows", 0)
    for N, rows, chid in [(1, 64, 64), (1, 1, 64), (2, 63, 64), (3, 1000, 96), (2, 4097, 128), (8, 28 ** 3, 64), (1, 112 ** 3, 64), (5, 777, 64)]:
        for add in (False, True):
            for wgs in (0, 1, 3):
                o = operands(N, rows, chid, seed=rows)
                knob("mlp_dma", 0)
                want = run(*o, add)
                knob("mlp_dma", 1)
                if wgs:
                    knob("mlp_dma_wgs", wgs)
                got = run(*o, add)
                torch.cuda.synchronize()
                ok = torch.equal(want.view(torch.int16), got.view(torch.int16))
                print(f"N {N} rows {rows} hid {chid} add {int(add)} wgs {wgs or 'auto'}: {'bit-identical' if ok else 'MISMATCH'}", flush=True)
                assert ok
                if wgs:
                    knob("mlp_dma_wgs", 0)


def timeit(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def time_kinds():
    knob("mlp_dma_rows", 0)
    for N, D, chid, n_head in [(8, 112, 64, 3), (2, 160, 96, 3)]:
        rows = D ** 3
        o = operands(N, rows, chid)
        x0, sw, sb, hw, hb = extras(N, rows, n_head, 1)
        y = torch.empty(N, rows, 32, device=dev, dtype=bf)
        for label, fn, nbytes in [("stem residual", lambda: run_stem(*o, x0, sw, sb, y=y), N * rows * (128 + 4)),
                                  ("head (add, no y)", lambda: run_head(*o, hw, hb, True, False), N * rows * (128 + 4 * n_head))]:
            for rnd in range(2):
                for dma in (0, 1):
                    knob("mlp_dma", dma)
                    us = timeit(fn)
                    print(f"x{N} {D}^3 hid {chid} {label:18s} | {'dma' if dma else 'one tile per wave':18s} {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s", flush=True)


def time():
    knob("mlp_dma_rows", 0)
    for N, D, chid in [(8, 112, 64), (2, 160, 128)]:
        rows = D ** 3
        o = operands(N, rows, chid)
        y = torch.empty(N, rows, 32, device=dev, dtype=bf)
        for add in (False, True):
            nbytes = N * rows * 2 * 32 * (3 if add else 2)
            for rnd in range(2):
                for label, kn in [("one tile per wave", {"mlp_dma": 0}), ("dma auto", {"mlp_dma": 1, "mlp_dma_wgs": 0})] + \
                        [(f"dma wgs {w}", {"mlp_dma": 1, "mlp_dma_wgs": w}) for w in ((96, 384, 768) if rnd == 0 else ())]:
                    for k, v in kn.items():
                        knob(k, v)
                    us = timeit(lambda: run(*o, add, y=y))
                    print(f"x{N} {D}^3 hid {chid} {'add' if add else 'none'} | {label:18s} {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s", flush=True)
    knob("mlp_dma_wgs", 0)


def chunk_mixers():
    """the chunk-streamed mixer (wide hidden layers) against the streaming kernel: bits on ragged shapes, then time per variant at MedNeXt-L's"""
    shapes = [(128, 1024, 128, "add", 2, 7), (128, 1024, 128, "none", 1, 3), (256, 2048, 128, "up", 2, 6), (128, 512, 64, "up", 3, 4),
              (64, 512, 128, "none", 2, 9), (128, 96, 128, "add", 2, 5), (128, 64, 64, "add", 1, 4),
              (64, 128, 32, "up", 2, 6), (64, 256, 64, "add", 2, 7),
              (128, 1024, 128, "add", 2, 40), (256, 2048, 128, "up", 2, 40), (128, 512, 64, "up", 2, 80), (64, 512, 128, "none", 2, 40),
              (64, 256, 64, "add", 2, 80), (64, 128, 64, "add", 8, 56), (64, 128, 32, "up", 8, 112), (128, 256, 128, "add", 8, 28),
              (128, 256, 64, "up", 8, 56), (64, 192, 32, "up", 2, 160)]
    for folded in (True, False):
        for cin, chid, cout, mode, N, D in shapes:
            rows = D ** 3
            big = rows > 50000
            if big and not folded:
                continue
            g = torch.Generator(device="cpu").manual_seed(rows + cin)
            t = torch.randn(N, rows, cin, generator=g).to(bf).to(dev)
            res = torch.randn(N, rows, cout, generator=g).to(bf).to(dev)
            w3 = ops.pw_pack_weight_paired((torch.randn(cout, chid, generator=g) / chid ** 0.5).to(dev), f16=True)
            b3 = torch.randn(cout, generator=g).to(dev)
            if folded:
                ab = None
                w2 = torch.stack([ops.pw_pack_weight_paired((torch.randn(chid, cin, generator=g) / cin ** 0.5).to(dev)) for _ in range(N)])
                b2 = torch.randn(N, chid, generator=g).to(dev)
            else:
                ab = torch.stack([torch.rand(N, cin, generator=g) + 0.5, torch.randn(N, cin, generator=g) * 0.5], 1).contiguous().to(dev)
                w2 = ops.pw_pack_weight_paired((torch.randn(chid, cin, generator=g) / cin ** 0.5).to(dev))
                b2 = torch.randn(chid, generator=g).to(dev)
            kw = dict(N=N, rows_per_sample=rows, c_in=cin, c_hid=chid, c_out=cout)
            if mode == "add":
                kw.update(res=res, res_mode=nat.RES_ADD)
            elif mode == "up":
                low = torch.randn(N, rows // 8, cout, generator=g).to(bf).to(dev)
                kw.update(res=res, res_mode=nat.RES_UPSAMPLE, grid=(D, D, D), res_low=low, res_bias=b3)
            want = ops.pw_mlp(t, ab, w2, b2, w3, b3, **kw)
            y = torch.empty_like(want)
            flops = 2.0 * N * rows * chid * (cin + cout)
            if big:
                us = timeit(lambda: ops.pw_mlp(t, ab, w2, b2, w3, b3, y=y, **kw))
                print(f"streaming    {cin}->{chid}->{cout} {mode} x{N} {D}^3: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
            if big and ops.pw_mlp_lds_supported(cin, chid, cout):
                us = timeit(lambda: ops.pw_mlp(t, ab, w2, b2, w3, b3, y=y, lds=True, **kw))
                print(f"lds-resident {cin}->{chid}->{cout} {mode} x{N} {D}^3: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
            for variant in (1, 2, 3, 4):
                knob("mlp_chunk_variant", variant)
                y.fill_(float("nan"))
                ops.pw_mlp(t, ab, w2, b2, w3, b3, y=y, chunked=True, **kw)
                torch.cuda.synchronize()
                ok = torch.equal(y.view(torch.int16), want.view(torch.int16))
                us = timeit(lambda: ops.pw_mlp(t, ab, w2, b2, w3, b3, y=y, chunked=True, **kw)) if big else 0.0
                print(f"chunk mixer  {cin}->{chid}->{cout} {mode} x{N} {D}^3 folded {int(folded)} variant {variant}: {'bit-identical' if ok else 'MISMATCH'}"
                      + (f"  {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s" if big else ""), flush=True)
                assert ok
    knob("mlp_chunk_variant", 0)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    if "check" in what:
        check_kinds()
        check()
    if "time" in what:
        time_kinds()
        time()
    if "chunk" in what:
        chunk_mixers()
