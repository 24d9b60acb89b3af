"""Pin the MedNeXt oracle to upstream `nnunet_mednext` -- one command, for the day the package is importable.

    python tools/pin_mednext.py [--sizes S L] [--out tests/golden] [--checkpoint path.ckpt]

The reference delegates MedNeXt's arithmetic to the third-party, un-vendored, un-pinned `nnunet_mednext` package
(connectomics/models/architectures/mednext_models.py:24-25, 374-380, 449-479), which this image does not have: `oracle/mednext_oracle.py`
restates the published architecture and is anchored on parameter counts only ("parity unpinned").  Where the package
imports (any machine with `pip install git+https://github.com/MIC-DKFZ/MedNeXt`), this script builds upstream `create_mednext_v1` for
every requested size, loads THIS package's state dict into it with strict=True (the key / shape contract), runs a seeded forward +
backward on the CPU and writes `tests/golden/mednext_<size>.npz`:

    x, sd__<key>..., y, ds_0..ds_4 (deep-supervision run), grad__<key> for stem / first block / bottleneck / last block / head weights

`tests/test_oracle_golden.py::test_mednext_oracle_against_upstream_fixture` consumes the files when they exist (skips otherwise) and
compares oracle forward, deep-supervision outputs and those gradients against them: the pin.  `--checkpoint` additionally loads a
reference-trained Lightning checkpoint (keys `model.model.*`) into both sides first, so the pin covers real weights.
Nothing here runs on the GPU box or in the CPU suite; it only writes data (inputs + expected outputs).
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

GRAD_KEYS = ("stem.weight", "enc_block_0.0.conv1.weight", "enc_block_0.0.conv2.weight", "bottleneck.0.conv3.weight",
             "dec_block_0.1.norm.weight", "up_0.conv1.weight", "out_0.conv_out.weight")
PATCH = {"S": 32, "B": 32, "M": 32, "L": 32}          # every level of the trunk needs >= 2 voxels: 32 -> 2 at the bottleneck


def build_upstream(size: str, k: int, ds: bool):
    try:
        from nnunet_mednext import create_mednext_v1
    except ImportError as e:      # the expected outcome in this image
        raise SystemExit("nnunet_mednext is not importable here: nothing written.  Run this script where the package is installed "
                         f"(pip install git+https://github.com/MIC-DKFZ/MedNeXt).  [{e}]")
    return create_mednext_v1(num_input_channels=1, num_classes=2, model_id=size, kernel_size=k, deep_supervision=ds)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", nargs="+", default=["S", "L"])
    ap.add_argument("--kernel-size", type=int, default=3)
    ap.add_argument("--out", default=str(ROOT / "tests" / "golden"))
    ap.add_argument("--checkpoint", default=None)
    args = ap.parse_args()
    from pytorch_connectomics_amd.models.architectures.mednext import create_mednext_v1 as create_ours
    for size in args.sizes:
        torch.manual_seed(1234)
        ours = create_ours(1, 2, size, args.kernel_size, deep_supervision=True)
        sd = {k: v.detach().clone() for k, v in ours.state_dict().items()}
        if args.checkpoint:
            ck = torch.load(args.checkpoint, map_location="cpu", weights_only=True)
            src = {k[len("model.model."):]: v for k, v in ck["state_dict"].items() if k.startswith("model.model.")}
            missing = [k for k in sd if k not in src]
            if missing:
                raise SystemExit(f"checkpoint lacks {missing[:5]} ...: not a MedNeXt-{size} k{args.kernel_size} model")
            sd = {k: src[k].float() for k in sd}
        up = build_upstream(size, args.kernel_size, True)
        up.load_state_dict(sd, strict=True)          # THE key / shape contract: upstream accepts this package's state dict as is
        up.eval()
        p = PATCH[size]
        x = torch.rand(1, 1, p, p, p, generator=torch.Generator().manual_seed(77))
        outs = up(x)
        assert isinstance(outs, (list, tuple)) and len(outs) == 5, "deep supervision must return [out, out_1 .. out_4]"
        loss = sum((o.float() ** 2).mean() * (0.5 ** i) for i, o in enumerate(outs))
        loss.backward()
        grads = dict(up.named_parameters())
        arrs = {"x": x.numpy(), "loss": np.asarray([float(loss)], np.float64), "y": outs[0].detach().numpy()}
        for i, o in enumerate(outs):
            arrs[f"ds_{i}"] = o.detach().numpy()
        for k, v in sd.items():
            arrs["sd__" + k] = v.numpy()
        for k in GRAD_KEYS:
            if k in grads and grads[k].grad is not None:
                arrs["grad__" + k] = grads[k].grad.numpy()
        up_nods = build_upstream(size, args.kernel_size, False)
        arrs["n_params_without_ds"] = np.asarray([sum(q.numel() for q in up_nods.parameters())], np.int64)
        arrs["has_ds_heads_without_ds"] = np.asarray([int(any(k.startswith("out_1.") for k in up_nods.state_dict()))], np.int64)
        out = Path(args.out) / f"mednext_{size.lower()}_k{args.kernel_size}.npz"
        np.savez_compressed(out, **arrs)
        print(f"wrote {out}: {len(arrs)} arrays, loss {float(loss):.6f}")


if __name__ == "__main__":
    main()
