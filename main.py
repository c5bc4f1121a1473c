#!/usr/bin/env python
"""FE training entry point — `python main.py --config <config.py>` — same CLI and side effects as the reference's
main.py (/root/reference/main.py:18-93): a run directory `config.output/<YYYYmmdd-HHMMSS>/{checkpoints,img}` is created
by the main process, the config file is copied into it, then Controller(config) → configure_trainer → trainer.fit.
MLflow is optional (absent here): metrics go to `<run>/metrics.jsonl`."""
import argparse
import os
import shutil
import sys
import warnings
from datetime import datetime
from pathlib import Path

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import pets_face_recognition_amd as pfr  # noqa: E402

pfr.install_reference_aliases()

from engine import Controller  # noqa: E402
from utils import is_main_process, configure_trainer, get_config, find_max_batch_size, find_optimal_init_lr  # noqa: E402


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('-c', '--config', required=True, type=Path, help='Path to config file')
    return parser.parse_args(argv)


def main(argv=None):
    warnings.simplefilter('ignore')
    args = parse_args(argv)
    config = get_config(args.config)
    checkpoint_path = None
    logger = None
    if is_main_process():
        run_root = Path(config.output) / datetime.now().strftime("%Y%m%d-%H%M%S")
        config.output = run_root
        checkpoint_path = run_root / 'checkpoints'
        config.checkpoint_path = checkpoint_path
        config.img_dir = run_root / 'img'
        checkpoint_path.mkdir(parents=True, exist_ok=True)
        config.img_dir.mkdir(exist_ok=True)
        shutil.copy2(args.config, run_root)
        if config.get('mlflow_target_uri') is not None:
            try:
                from engine.loggers import MLFlowLogger
                logger = MLFlowLogger(config.get('mlflow_target_uri'), config.get('experiment_name', 'default'),
                                      config.get('run_name', 'default'))
            except ImportError:
                print('mlflow is not installed: metrics are written to', run_root / 'metrics.jsonl')
    controller = Controller(config=config)
    trainer = configure_trainer(config, logger, checkpoint_path)
    if config.get('find_max_batch_size'):        # reference main.py:79-84
        new_batch_size = find_max_batch_size(trainer, controller)
        if new_batch_size:
            config.train_batch_size = new_batch_size
            config.test_batch_size = new_batch_size
        controller = Controller(config=config)
    if config.get('find_optimal_init_lr'):       # reference main.py:86-89 (its `opt_params` write is kept when the config has one)
        new_lr = find_optimal_init_lr(trainer, controller)
        if new_lr is not None:
            config.init_lr = new_lr
            if config.get('opt_params') is not None:
                config.opt_params['main']['lr'] = new_lr
        controller = Controller(config=config)
    trainer.fit(controller)
    if getattr(trainer, 'train_img_s', None) and getattr(trainer, 'rank', 0) == 0:
        import json
        rec = {'train_img_s': round(trainer.train_img_s, 1), 'config': str(args.config), 'prefetch_batches': trainer.prefetch_batches}
        print('THROUGHPUT ' + json.dumps(rec))
    print('Completed!')
    return controller, trainer


if __name__ == '__main__':
    main()
