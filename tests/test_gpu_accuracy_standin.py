"""Accuracy stand-in for the north-star acceptance number (Lucchi++ Jaccard within 0.002 of the reference,
/root/reference/README.md:42): no dataset or checkpoint exists in this image, so the SAME synthetic segmentation task is trained
from the SAME initial weights on the HIP path (fp32 and bf16 storage) and on the CPU oracle (RSUNet: the oracle pinned to the
reference's own module by tests/golden/rsunet_*.npz), a held-out volume is predicted with the sliding-window engine on each
side, and the Jaccard indices of the thresholded predictions have to agree to 0.002."""
import numpy as np
import pytest
import torch

from oracle import rsunet_oracle as RO
from oracle import window_oracle as WO

pytestmark = pytest.mark.gpu

KW = dict(width=[8, 16], down_factors=[(2, 2, 2)], norm="group", num_groups=4, activation="elu")
ROI = (32, 32, 32)
STEPS, BATCH, LR = 200, 2, 5e-3


def _blob_volume(shape, seed):
    """Smooth random field thresholded into blobs (the label); image = blobs darker than a textured background + noise."""
    g = torch.Generator().manual_seed(seed)
    field = torch.randn((1, 1) + tuple(s // 4 + 2 for s in shape), generator=g)
    field = torch.nn.functional.interpolate(field, size=shape, mode="trilinear", align_corners=False)[0, 0]
    label = (field > 0.35).float()
    image = 0.65 - 0.35 * label + 0.08 * torch.randn(shape, generator=g) + 0.05 * field
    return image.clamp(0, 1), label


def _loss(out, y):
    from pytorch_connectomics_amd.training.module import dice_loss_sigmoid, weighted_bce_with_logits
    return weighted_bce_with_logits(out, y, None, None) + dice_loss_sigmoid(out, y)


def _jaccard(pred_logits, label):
    p, t = (pred_logits > 0), (label > 0.5)
    return float((p & t).sum()) / max(float((p | t).sum()), 1.0)


def test_hip_training_reaches_the_oracles_jaccard_on_a_synthetic_blob_task():
    from pytorch_connectomics_amd.inference.window import EagerSlidingWindowEngine
    from pytorch_connectomics_amd.models.architectures.rsunet import RSUNet
    train_img, train_lab = _blob_volume((64, 96, 96), 1)
    test_img, test_lab = _blob_volume((48, 80, 80), 2)
    g = torch.Generator().manual_seed(3)
    crops = []
    for _ in range(STEPS):
        xs, ys = [], []
        for _b in range(BATCH):
            z, y, x = (int(torch.randint(0, s - r + 1, (1,), generator=g)) for s, r in zip(train_img.shape, ROI))
            xs.append(train_img[z:z + 32, y:y + 32, x:x + 32])
            ys.append(train_lab[z:z + 32, y:y + 32, x:x + 32])
        crops.append((torch.stack(xs)[:, None], torch.stack(ys)[:, None]))
    torch.manual_seed(0)
    init = {k: v.detach().clone() for k, v in RSUNet(1, 1, **KW).state_dict().items()}

    # ---- CPU oracle: torch autograd through oracle/rsunet_oracle.py + torch.optim.AdamW
    params = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "up.weight" not in k) for k, v in init.items()}
    opt = torch.optim.AdamW([p for p in params.values() if p.requires_grad], lr=LR, weight_decay=1e-2)
    for x, y in crops:
        opt.zero_grad(set_to_none=True)
        _loss(RO.forward(params, x, **KW), y).backward()
        opt.step()
    with torch.no_grad():
        st = {k: v.detach() for k, v in params.items()}
        ref = WO.eager_sliding_window(test_img[None, None], lambda x: RO.forward(st, x, **KW), roi=ROI, overlap=0.5, mode="bump",
                                      sw_batch_size=4)
    j_ref = _jaccard(ref[0, 0], test_lab)

    # ---- HIP path, both storage widths, same init / batches / optimizer settings
    scores = {}
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        m = RSUNet(1, 1, **KW)
        m.load_state_dict(init)
        m = m.cuda().train()
        m.compute_dtype = dt
        o = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=LR, weight_decay=1e-2)
        for x, y in crops:
            o.zero_grad(set_to_none=True)
            _loss(m(x.cuda()), y.cuda()).backward()
            o.step()
        m.eval()
        eng = EagerSlidingWindowEngine(roi_size=ROI, sw_batch_size=4, overlap=0.5, mode="bump", padding_mode="constant", cval=0.0)
        with torch.no_grad():
            pred = eng(test_img[None, None].cuda(), m).cpu()
        scores[name] = _jaccard(pred[0, 0], test_lab)
    print(f"[accuracy stand-in] Jaccard: oracle {j_ref:.4f}, HIP fp32 {scores['fp32']:.4f}, HIP bf16 {scores['bf16']:.4f}")
    assert j_ref > 0.8, j_ref                                   # the task was learned
    assert abs(scores["fp32"] - j_ref) <= 0.002, (scores, j_ref)
    assert abs(scores["bf16"] - j_ref) <= 0.002, (scores, j_ref)
