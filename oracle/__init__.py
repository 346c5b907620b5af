"""CPU oracle for the CORE retrieval-time forward path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a plain PyTorch-CPU fp32 restatement of the reference's algorithm
(wangtong627/COR, files cited per function as ``ref: <file>:<lines>``), written
functionally over a flat ``state_dict`` so that the same function can be fed the
reference's parameters, the product's parameters, or seeded random ones.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / timed CPU baseline.
``cor_amd`` (the product) never imports it and has no CPU fallback.

Parity status
-------------
* SAM encoder, prompt encoder, two-way transformer, mask decoder, mask-adapter
  pooling, masked pooling, fuse module, region ``mask_pooling``: PINNED against
  the reference's own modules imported in the build container
  (``tools/make_golden.py`` -> ``tests/golden/*.npz``; checked by
  ``tests/test_oracle_golden.py``).
* Top-level glue (``SupportBranch.forward``, ``CirSegModel...forward``,
  ``build_model_with_query_support_feat``): PINNED by ``tests/golden/toplevel_*.npz``,
  outputs of the reference's own ``build_model...().forward`` on full SAM-B, run in the
  build container through an in-memory ``open_clip`` stand-in (2-block SigLIP) by
  ``tools/make_golden.py gen_toplevel``.
* SigLIP towers (third-party ``open_clip_torch==2.31.0`` / ``timm==1.0.15``,
  not vendored, not installed): PARITY UNPINNED by the reference; restated from
  the published architecture and cross-checked against
  ``transformers.models.siglip`` built from config (random init) in
  ``tests/test_oracle_siglip_hf.py``.
* Gallery similarity + top-k: not present in the reference at all
  (SURVEY.md fact 2); defined here from ``utils/loss_func.py:35-56,84``.
  PARITY UNPINNED by the reference; order is (score desc, index asc).
"""

from . import config, sam, siglip, support, model, retrieval  # noqa: F401
