# BASELINE.json configs[3] on one GPU: Swin-T FE (models/swin.py) + ArcFace, synthetic 10k ids, bs=128, MI355X bf16
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import make as _make

_make(globals(), arch='swin_t', n_train_ids=10000, n_val_ids=200, photos=4, image_size=224, train_bs=128, test_bs=64,
      device='cuda:0', n_epochs=1, limit_train_batches=int(os.environ.get('PFR_LIMIT_TRAIN_BATCHES', '20')), workers=8)
