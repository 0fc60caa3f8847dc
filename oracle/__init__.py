"""CPU oracle for the DOFA-ViT + UperNet segmentation hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch fp32 *restatement*
of the reference's algorithm for the hot path (SURVEY.md section 8a, rows
D1-D15, X2).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / CPU baseline --
never as the thing that is shipped or measured.  The product
(``geo-deep-learning_amd/``) must not import anything from here.

Pinning status
--------------
* In-repo reference code (DOFAv2 encoder, MultiLevelNeck, PPM, UperNetDecoder,
  FCNHead, SegmentationHead, DOFASegmentationModel, utils/tensors): PINNED.
  ``tools/make_goldens.py`` imports the real reference from /root/reference
  (this container only) and the committed fixtures under ``tests/golden/`` are
  outputs of the reference itself; ``tests/test_oracle_golden.py`` checks the
  restatement against them.  ``utils/tensors`` is also pinned by the
  reference's own known-answer tests (tests/test_utils_tensors.py:14-50).
* Third-party arithmetic that is NOT under /root/reference -- timm 1.0.24
  ``vision_transformer.Block`` and segmentation-models-pytorch 0.5.0
  ``DiceLoss`` -- "parity unpinned": restated from their published algorithm
  (SURVEY.md Appendix A.1 / A.5); the ViT block is cross-checked against the
  independent ``transformers`` Dinov2Layer implementation in
  ``tests/test_oracle_block_crosscheck.py``.
"""

from .model import (  # noqa: F401
    DOFASegmentationModel,
    SegmentationOutput,
    procedural_state_dict,
    synthetic_batch,
)
