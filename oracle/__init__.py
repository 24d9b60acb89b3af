"""CPU oracle for the PyTorch Connectomics hot path  --  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU (numpy + PyTorch-CPU fp32), the algorithms of the
reference's data-parallel hot path.  It exists so that the HIP path
in ``pytorch_connectomics_amd`` has something independent to be checked against.

Rules (enforced by tests/test_layout_rules.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
    import anything from here;
  * the product package never imports it and has no CPU fallback -- it raises when the
    HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * window planner / importance maps / eager engine / normalisation / chunk grid:
    PINNED against the reference's own modules run in the build container
    (tests/golden/*.npz, generator tests/golden/make_golden.py) and against the
    reference tests' closed-form properties.
  * RSUNet forward: PINNED against the reference's ``rsunet.py`` outputs (same fixtures).
  * MedNeXt forward: **parity unpinned** -- the arithmetic lives in the third-party,
    un-vendored, un-pinned ``nnunet_mednext`` package (reference call sites
    connectomics/models/architectures/mednext_models.py:24-25,374-380,449-479).  The
    restatement follows the published architecture (Roy et al., MICCAI 2023) and is
    anchored on the reference's parameter-count table (mednext_models.py:309-312) and
    the shape/consistency contracts of tests/unit/test_mednext_features.py.
"""
