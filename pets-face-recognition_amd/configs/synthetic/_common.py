"""Shared builder for the synthetic FE configs.  The config CONTRACT is the reference's
(/root/reference/configs/dog_fe/fe_dogs_config.py:67-162): model(), loss(config, model_), optimizer(model_),
train_dataloader(), val_dataloader(), pair_generator(idx), similarity_f(pairs), n_epochs, thrs, far_thr, k,
trainer_kwargs, output, device, distributed_train, world_size.  Data are synthetic (datasets need downloads)."""
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import DataLoader

import models
from data_loading import SyntheticRecDataset, RecSubset, PairGenerator
from losses import SoftmaxBasedMetricLearning



def _rank():
    """data-parallel rank of this process (torchrun), 0 otherwise: the reference's dataloader workers draw their augmentation
    decisions from per-worker / per-rank seeds, so the ranks must not share one decision stream"""
    import os
    return int(os.environ.get("RANK", "0"))

def _live(key, default):
    """value of `key` in the live Config object (main.py's batch-size / lr finders write there), else the config file's own"""
    from utils import Config, _SingletonBase
    inst = _SingletonBase._instances.get(Config)
    v = inst.get(key) if inst is not None else None
    return default if v is None else v


def make(ns, arch, n_train_ids, n_val_ids, photos, image_size, train_bs, test_bs, device, n_epochs=1, seed=0,
         fused_optimizer=True, compute_dtype=None, limit_train_batches=None, workers=0, n_pairs=200, device_augment=False, noise=0.15,
         noise_bank=0, limit_val_batches=None):
    torch.manual_seed(seed)
    dataset = SyntheticRecDataset(n_train_ids + n_val_ids, photos, image_size, seed=seed, noise=noise, raw_uint8=device_augment,
                                  noise_bank=noise_bank)
    train_users = list(range(n_train_ids))
    val_users = list(range(n_train_ids, n_train_ids + n_val_ids))
    labels = dataset.get_labels()
    train_idx = [i for i, u in enumerate(labels) if u < n_train_ids]
    val_idx = [i for i, u in enumerate(labels) if u >= n_train_ids]
    assert not (set(train_users) & set(val_users))
    train, val = RecSubset(dataset, train_idx), RecSubset(dataset, val_idx)
    n_pairs = min(n_pairs, n_val_ids * photos * (photos - 1))   # the reference asserts gen_number <= #ordered genuine pairs
    pair_gen = PairGenerator(dataset, n_pairs, 1, None, seed, val_users)

    def pair_generator(idx):
        if idx in (0, 1):
            return ('Val', 'Val 1')[idx], pair_gen
        raise Exception

    def similarity_f(pairs):
        t1 = torch.stack([p[0] for p in pairs])
        t2 = torch.stack([p[1] for p in pairs])
        return (F.cosine_similarity(t1, t2) + 1) / 2

    similarity_f._is_default_cosine = True   # lets the evaluator use the fused HIP pair-score kernel

    def model():
        kw = {} if compute_dtype is None else {'compute_dtype': compute_dtype}
        if arch.startswith('swin'):   # the reference's own backbone (models/swin.py:228-241): `mlp_head` IS the 512-d embedding layer
            return getattr(models, arch)(num_classes=512, **kw)
        model_ = getattr(models, arch)(**kw)
        model_.fc = torch.nn.Linear(model_.fc.in_features, 512)
        return model_

    def loss(config, model_):
        _ = config
        return SoftmaxBasedMetricLearning(model=model_, num_class=n_train_ids, embedding_size=512, is_focal=True,
                                          arc_margin=True)

    def optimizer(model_):
        params1 = [p for i, p in model_.module.named_parameters() if 'fc' not in i]
        params2 = [p for i, p in model_.module.named_parameters() if 'fc' in i]
        base = _live('init_lr', 10 ** -2)
        d = [{'lr': base / 2, 'params': params1},
             {'lr': base, 'params': params2},
             {'lr': base, 'params': list(model_.add_margin.parameters()), 'weight_decay': 1 * (10 ** -4)}]
        d = [g for g in d if len(g['params'])]   # (a backbone without an `fc` layer — Swin's embedding layer is `mlp_head` — has no second group)
        if fused_optimizer and device != 'cpu':
            from optim import FusedSGD
            optim = FusedSGD(d, 0.01, momentum=0.9)
        else:
            optim = torch.optim.SGD(d, 0.01, momentum=0.9)
        sched = torch.optim.lr_scheduler.MultiStepLR(optim, milestones=[35, 45], gamma=0.1)
        return [optim], [sched]

    def train_dataloader():
        # pinned batches for the copy stream (data_loading/prefetch.py); workers stay alive across epochs like the reference's
        # persistent_workers loaders (fe_dogs_config.py:135-143)
        # (batch size and the base rate are read from the config namespace at call time: main.py's find_max_batch_size /
        # find_optimal_init_lr write train_batch_size / init_lr there, reference main.py:79-89)
        return DataLoader(train, _live('train_batch_size', train_bs), shuffle=True, drop_last=True, num_workers=workers, pin_memory=device != 'cpu',
                          persistent_workers=workers > 0, prefetch_factor=4 if workers > 0 else None)

    def val_dataloader():
        return DataLoader(val, _live('test_batch_size', test_bs), num_workers=0)

    if device_augment:
        # the reference's train/val Compose pipelines (fe_dogs_config.py:17-32) applied on the device to uint8 batches
        from data_loading import DeviceAugmentation, val_augmentation
        ns['device_train_augmentation'] = DeviceAugmentation((image_size - 4, image_size - 4), (image_size, image_size), 0.1, 0.3,
                                                             5.0, torch.Generator().manual_seed(seed + _rank()))
        ns['device_val_augmentation'] = val_augmentation()
    output = Path('results')
    output.mkdir(exist_ok=True)
    ns.update(dict(
        n_epochs=n_epochs, train_batch_size=train_bs, test_batch_size=test_bs,
        thrs=np.linspace(0.5, 0.99, 6), far_thr=[0.1, 0.05, 0.03, 0.01, 0.005, 0.001], k=[5, 10, 100],
        pair_generator=pair_generator, similarity_f=similarity_f, model=model, loss=loss, optimizer=optimizer,
        train_dataloader=train_dataloader, val_dataloader=val_dataloader,
        trainer_kwargs=dict(benchmark=True, limit_train_batches=limit_train_batches, limit_val_batches=limit_val_batches),
        output=output, experiment_name='Synthetic', run_name=f'{arch} synthetic',
        device=device, distributed_train=not isinstance(device, str),
        world_size=len(device) if not isinstance(device, str) else None))
