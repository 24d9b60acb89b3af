"""CPU checks of the MONAI-style U-Net: registry / builder surface (reference monai_models.py:197-250), the module tree's
state-dict vocabulary, and the CPU oracle's shape contract (reference tests/unit/test_registry_basic.py:64-138)."""
from types import SimpleNamespace as NS

import pytest
import torch

from oracle import monai_unet_oracle as UO


def _cfg(filters, norm="batch", size=(16, 32, 32), out_ch=2, **extra):
    mon = dict(filters=list(filters), num_res_units=2, kernel_size=3, norm=norm, num_groups=2, dropout=0.0,
               upsample_mode="deconv")
    mon.update(extra)
    return NS(model=NS(arch=NS(type="monai_unet"), in_channels=1, out_channels=out_ch, input_size=list(size), monai=NS(**mon)))


def test_registered_and_builds_without_gpu():
    from pytorch_connectomics_amd.models import build_model, is_architecture_available, list_architectures
    assert is_architecture_available("monai_unet") and "monai_unet" in list_architectures()
    m = build_model(_cfg((32, 64, 128, 256)))
    assert type(m).__name__ == "MONAIModelWrapper" and m.supports_deep_supervision is False and m.output_scales == 1
    keys = list(m.state_dict())
    assert keys[0] == "model.model.0.conv.unit0.conv.weight"
    assert "model.model.1.submodule.1.submodule.1.submodule.conv.unit1.adn.A.weight" in keys       # bottom ResidualUnit
    assert "model.model.1.submodule.1.submodule.1.submodule.residual.weight" in keys               # 128 -> 256, 1x1x1
    assert m.state_dict()["model.model.1.submodule.1.submodule.1.submodule.residual.weight"].shape == (256, 128, 1, 1, 1)
    assert m.state_dict()["model.model.0.residual.weight"].shape == (32, 1, 3, 3, 3)               # strided: k3
    assert m.state_dict()["model.model.2.0.conv.weight"].shape == (64, 2, 3, 3, 3)                 # ConvTranspose3d [in][out]
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 1, 16, 32, 32))


def test_builder_errors():
    from pytorch_connectomics_amd.models import build_model
    with pytest.raises(NotImplementedError, match="upsample_mode"):
        build_model(_cfg((8, 16), upsample_mode="nontrainable"))
    with pytest.raises(NotImplementedError, match="3-D"):
        build_model(_cfg((8, 16), size=(64, 64)))
    with pytest.raises(ValueError, match="Unsupported MONAI norm"):
        build_model(_cfg((8, 16), norm="layer"))


@pytest.mark.parametrize("filters,norm,size", [((8, 16, 32), "batch", (16, 32, 32)), ((4, 8), "instance", (8, 16, 16)),
                                               ((8, 8, 16, 16), "group", (16, 16, 32))])
def test_oracle_shape_contract_and_determinism(filters, norm, size):
    from pytorch_connectomics_amd.models import build_model
    torch.manual_seed(0)
    st = build_model(_cfg(filters, norm, size, out_ch=4)).state_dict()
    x = torch.rand(2, 1, *size)
    with torch.no_grad():
        y = UO.forward(st, x, n_levels=len(filters), norm=norm, num_groups=2)
        y2 = UO.forward(st, x, n_levels=len(filters), norm=norm, num_groups=2)
    assert y.shape == (2, 4) + tuple(size) and torch.equal(y, y2) and torch.isfinite(y).all()
    if norm == "batch":      # train-mode statistics differ from the (0, 1) running buffers of a fresh model
        with torch.no_grad():
            yt = UO.forward(st, x, n_levels=len(filters), norm=norm, training=True)
        assert not torch.allclose(y, yt)
