# fe_r18_cpu with the reference's two tuner switches on (main.py:79-89): batch-size finder, then the learning-rate finder
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _common import make as _make

_make(globals(), arch='resnet18', n_train_ids=24, n_val_ids=6, photos=4, image_size=64, train_bs=8, test_bs=8,
      device='cpu', n_epochs=1, limit_train_batches=int(os.environ.get('PFR_LIMIT_TRAIN_BATCHES', '2')), n_pairs=20)
find_max_batch_size = True
find_max_batch_size_kwargs = dict(steps_per_trial=1, init_val=4, max_trials=3)
find_optimal_init_lr = True
find_optimal_init_lr_kwargs = dict(min_lr=1e-5, max_lr=1.0, num_training=16)
