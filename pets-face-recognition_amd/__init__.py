"""pets-face-recognition_amd — MI355X-native feature-extractor hot path of MarQuisCheshire/Pets-Face-Recognition.

Layout
  csrc/      hand-written HIP kernels for gfx950 + the C-ABI (libpfr_hip.so, declared in include/pfr_hip.h)
  _hip/      ctypes binding of the C-ABI and thin tensor-level wrappers
  models/    registry mirroring the reference's `models/` (+ the torchvision-compatible ResNet the configs build)
  losses/    `SoftmaxBasedMetricLearning`, `ArcMarginProduct`, `AddMarginProduct`, `FocalLoss` (same signatures)
  engine/    `Controller`, `Trainer` (plain loop replacing the PyTorch-Lightning glue), evaluation (candR@K)
  utils/     `get_config`, `Config`, `configure_trainer`, device / DDP selection
  optim/     flat-buffer fused SGD / AdamW
"""
__version__ = "0.1.0"

import os as _os

# The train step uses up to four HIP streams at once (main, weight-gradient side stream, gradient all-reduce, RCCL's own).
# HIP multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; ask for 8 so that none of them share a queue.
# Only effective if set before the HIP runtime creates its queues, hence at package import.
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"
    import sys as _sys
    _t = _sys.modules.get("torch")
    if _t is not None and _t.cuda.is_available() and _t.cuda.is_initialized():
        # torch created the HIP queues before this import: the setting above came too late for this process
        import warnings as _w
        _w.warn("pets_face_recognition_amd imported after torch initialised HIP: export GPU_MAX_HW_QUEUES=8 before the first "
                "CUDA call, otherwise the side / comm streams share hardware queues with the main stream (~7 % slower DDP step)")


def install_reference_aliases():
    """Expose the sub-packages under the reference's top-level names (`losses`, `models`, `engine`, `utils`,
    `data_loading`) so that reference-style config files (`from losses import SoftmaxBasedMetricLearning`, …) run
    unchanged.  Called by main.py / eval scripts."""
    import importlib
    import sys
    for name in ("utils", "models", "losses", "data_loading", "engine", "match", "optim"):
        sys.modules.setdefault(name, importlib.import_module(__name__ + "." + name))
