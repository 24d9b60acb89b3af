"""Round-3 probe: stride-2 dense-conv weight gradient on the matrix cores (conv3d_wgrad_s2_mfma_kernel) against the VALU kernel it
replaces (knob conv_wgrad_s2_mfma) and against an fp64 einsum on a small case; time at the MONAI-style U-Net's shapes."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from pytorch_connectomics_amd import _native as nat  # noqa: E402
from pytorch_connectomics_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda")


def knob(v):
    nat.check(nat.lib().pytc_set_tuning(b"conv_wgrad_s2_mfma", v), "set")


def ref64(big, small):
    """dW[o][k][tz][ty][tx] = sum_r small[r][o] * big[2r + t - 1][k] in fp64 via unfold-free loops over taps"""
    N, Db, Hb, Wb, Ck = big.shape
    _, Ds, Hs, Ws, Co = small.shape
    bp = torch.nn.functional.pad(big.double().permute(0, 4, 1, 2, 3), (1, 1, 1, 1, 1, 1))
    sm = small.double()
    out = torch.zeros(Co, Ck, 3, 3, 3, dtype=torch.float64, device=big.device)
    for tz in range(3):
        for ty in range(3):
            for tx in range(3):
                sl = bp[:, :, tz:tz + 2 * Ds:2, ty:ty + 2 * Hs:2, tx:tx + 2 * Ws:2]          # (N, Ck, Ds, Hs, Ws)
                out[:, :, tz, ty, tx] = torch.einsum("nzyxo,nkzyx->ok", sm, sl)
    return out


torch.manual_seed(0)
for (N, sd, ck, co) in ((1, (5, 9, 37), 32, 48), (2, (3, 4, 70), 16, 16), (1, (6, 8, 33), 64, 32)):
    small = torch.randn(N, *sd, co, device=dev).bfloat16()
    big = torch.randn(N, 2 * sd[0], 2 * sd[1], 2 * sd[2], ck, device=dev).bfloat16()
    knob(1)
    got = ops.conv3d_wgrad_strided(big, small, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    got2 = ops.conv3d_wgrad_strided(big, small, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    knob(0)
    old = ops.conv3d_wgrad_strided(big, small, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    want = ref64(big, small)
    e_new = float((got.double() - want).abs().max() / want.abs().max())
    e_old = float((old.double() - want).abs().max() / want.abs().max())
    print(f"N={N} small={sd} {ck}x{co}: mfma rel err {e_new:.2e}  valu rel err {e_old:.2e}  deterministic {torch.equal(got, got2)}", flush=True)
    # odd big extents (conv with odd input: Db = 2 Ds - 1)
    big_odd = big[:, :2 * sd[0] - 1, :2 * sd[1] - 1, :2 * sd[2] - 1].contiguous()
    knob(1); g1 = ops.conv3d_wgrad_strided(big_odd, small, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    knob(0); g0 = ops.conv3d_wgrad_strided(big_odd, small, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    print(f"   odd big extents: mfma vs valu rel diff {float((g1 - g0).abs().max() / g0.abs().max()):.2e}")

for (ck, co, sd) in ((32, 64, (12, 128, 128)), (64, 128, (6, 64, 64)), (128, 256, (3, 32, 32)), (64, 384, (6, 64, 64)), (32, 128, (12, 128, 128))):
    small = torch.randn(2, *sd, co, device=dev).bfloat16()
    big = torch.randn(2, 2 * sd[0], 2 * sd[1], 2 * sd[2], ck, device=dev).bfloat16()
    res = {}
    for v in (0, 1):
        knob(v)
        ops.conv3d_wgrad_strided(big, small, (3, 3, 3), (2, 2, 2), (1, 1, 1)); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = ops.conv3d_wgrad_strided(big, small, (3, 3, 3), (2, 2, 2), (1, 1, 1))
        e1.record(); torch.cuda.synchronize()
        res[v] = (e0.elapsed_time(e1) / 5, out)
    d = float((res[0][1] - res[1][1]).abs().max() / res[0][1].abs().max())
    fl = 2 * 27 * 2 * sd[0] * sd[1] * sd[2] * ck * co
    print(f"big C {ck} x small C {co} at small grid {sd}: valu {res[0][0]:.3f} ms  mfma {res[1][0]:.3f} ms ({fl / res[1][0] / 1e9:.1f} TFLOP/s)  rel diff {d:.2e}", flush=True)
knob(1)
