# The fe_r50_mi355x workload with the input pipeline built for the device: the loader hands out raw HWC uint8 frames (what
# RecDataset holds before its transform), batches are pinned and copied one-two steps ahead on a copy stream
# (data_loading/prefetch.py), and the reference's train Compose pipeline (fe_dogs_config.py:17-32) runs on the GPU for the
# whole batch (data_loading/augment.py).  Used to report end-to-end img/s next to bench.py's resident-input number.
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import make as _make

_make(globals(), arch='resnet50', n_train_ids=10000, n_val_ids=int(os.environ.get('PFR_VAL_IDS', '200')), photos=4, image_size=224, train_bs=256, test_bs=64,
      device='cuda:0', n_epochs=1, limit_train_batches=int(os.environ.get('PFR_LIMIT_TRAIN_BATCHES', '60')),
      workers=int(os.environ.get('PFR_WORKERS', '16')), device_augment=True, noise_bank=64)
