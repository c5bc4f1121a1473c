#!/usr/bin/env python
"""FE evaluation entry point — counterpart of /root/reference/eval_fe_dog_head_sgd.py:9-27 and
eval_fe_cat_head_sgd.py: load config, optionally a reference-format `state_dict` (strict=False: the margin weight may
be stripped, download_models.py:8-9), run `trainer.test(controller)` → Controller.test_epoch_end (ROC AUC, Accuracy,
Recall@K=10/100)."""
import argparse
import os
import sys
from pathlib import Path

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import pets_face_recognition_amd as pfr  # noqa: E402

pfr.install_reference_aliases()

import torch  # noqa: E402
from engine import Controller  # noqa: E402
from utils import configure_trainer, get_config  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('-c', '--config', required=True, type=Path)
    ap.add_argument('--checkpoint', type=Path, default=None, help='state_dict with keys model_loss.module.* (optional)')
    args = ap.parse_args(argv)
    config = get_config(args.config)
    controller = Controller(config)
    if args.checkpoint is not None:
        sd = torch.load(args.checkpoint, map_location='cpu')
        sd = sd.get('state_dict', sd)
        missing, unexpected = controller.load_state_dict(sd, strict=False)
        print('loaded', args.checkpoint, 'missing', len(missing), 'unexpected', len(unexpected))
    trainer = configure_trainer(config, None, None)
    return trainer.test(controller)


if __name__ == '__main__':
    main()
