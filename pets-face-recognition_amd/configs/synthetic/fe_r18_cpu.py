# BASELINE.json configs[0]: ResNet-18 FE + ArcFace, 100-class synthetic 224x224, bs=32, PyTorch CPU via main.py (plumbing)
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import make as _make

_make(globals(), arch='resnet18', n_train_ids=100, n_val_ids=12, photos=4, image_size=224, train_bs=32, test_bs=20,
      device='cpu', n_epochs=1, limit_train_batches=int(os.environ.get('PFR_LIMIT_TRAIN_BATCHES', '2')), n_pairs=40)
