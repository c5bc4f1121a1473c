# BASELINE.json configs[3]: Swin-T FE (models/swin.py) + ArcFace, bs=128/GPU, 8xMI355X data parallel (launch with torchrun, 8 procs)
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import make as _make

_make(globals(), arch='swin_t', n_train_ids=10000, n_val_ids=200, photos=4, image_size=224, train_bs=128, test_bs=64,
      device=[0, 1, 2, 3, 4, 5, 6, 7], n_epochs=1, limit_train_batches=int(os.environ.get('PFR_LIMIT_TRAIN_BATCHES', '20')),
      workers=8)
