"""oracle/ — CPU restatement of the reference's algorithm for the FE hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and only as the
checker.  Nothing under pets-face-recognition_amd/ imports it.

Pinning status:
  * arcface_ref / match_ref / swin: PINNED — checked bit-for-bit (≤1e-6) against the reference's own modules imported
    in the build container (losses/*.py, engine/controller.py:77-90 via PL stubs, models/swin.py) by
    oracle/make_golden.py, which also wrote the fixtures under tests/golden/.
  * resnet_ref: "parity unpinned" at the third-party boundary — the arithmetic lives in torchvision (absent from the
    reference tree and from this image, requirements.txt:3 pins only >=0.12); the restatement follows the published
    torchvision ResNet v1.5 definition and is self-checked by parameter counts / state-dict keys / op-level agreement
    with torch.nn.functional.
"""
