# BASELINE.json configs[1]: ResNet-50 FE + ArcFace, Petfinder-dogs-scale synthetic (10k id), bs=256, 1xMI355X bf16
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import make as _make

_make(globals(), arch='resnet50', n_train_ids=10000, n_val_ids=200, photos=4, image_size=224, train_bs=256, test_bs=64,
      device='cuda:0', n_epochs=1, limit_train_batches=int(os.environ.get('PFR_LIMIT_TRAIN_BATCHES', '20')), workers=8)
