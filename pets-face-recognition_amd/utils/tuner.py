"""Batch-size and learning-rate finders behind `find_max_batch_size` / `find_optimal_init_lr`.

The reference delegates both to PyTorch-Lightning's tuner (/root/reference/utils/__init__.py:137-148, called from
/root/reference/main.py:79-89 when the config sets `find_max_batch_size` / `find_optimal_init_lr`).  These are the two
procedures that tuner runs, restated for this package's plain Trainer:
  * scale_batch_size(mode='power'): double the batch size from `init_val`, `steps_per_trial` training steps per size, until
    a step runs out of memory, the dataset is smaller than the batch, or `max_trials` sizes have been tried; return the last
    size that ran.
  * lr_find(mode='exponential'): `num_training` steps with every group's learning rate swept from `min_lr` to `max_lr`
    geometrically, exponentially smoothed loss (beta 0.98, bias-corrected), stop when the loss exceeds
    `early_stop_threshold` x the best seen; the suggestion is the rate at the steepest descent of the smoothed curve
    (first 10 and last point skipped).
Model, optimizer-independent buffers and BN statistics are restored afterwards, as the tuner does."""
import copy

import numpy as np
import torch
from torch.utils.data import DataLoader


def _unpack_optimizers(opt):
    if isinstance(opt, (tuple, list)) and len(opt) == 2 and isinstance(opt[0], (list, tuple)):
        return list(opt[0])
    return [opt]


def _train_steps(trainer, controller, device, loader, optim, n_steps, lr_of_step=None, on_loss=None):
    from ..engine.trainer import _to_device
    controller.train()
    done = 0
    while done < n_steps:
        progressed = False
        for batch in loader:
            if lr_of_step is not None:
                for g in optim.param_groups:
                    g['lr'] = lr_of_step(done)
            batch = _to_device(batch, device)
            optim.zero_grad()
            loss = controller.training_step(batch, done)
            loss.backward()
            if trainer.ddp is not None:
                trainer.ddp.finish_backward()
            optim.step()
            done += 1
            progressed = True
            if on_loss is not None and on_loss(done - 1, float(loss.detach())):
                return done
            if done >= n_steps:
                break
        if not progressed:
            break
    return done


def _is_oom(e):
    msg = str(e).lower()
    return isinstance(e, (torch.cuda.OutOfMemoryError, MemoryError)) or 'out of memory' in msg or 'hiperroroutofmemory' in msg


def scale_batch_size(trainer, controller, mode='power', steps_per_trial=3, init_val=2, max_trials=25):
    if mode != 'power':
        raise ValueError("only mode='power' (the reference's default) is provided")
    device = trainer._setup(controller)
    saved = copy.deepcopy(controller.state_dict())
    dataset = controller.train_dataloader().dataset
    best, bs = None, int(init_val)
    for _ in range(max_trials):
        if bs > len(dataset):
            break
        optim = _unpack_optimizers(controller.configure_optimizers())[0]
        try:
            loader = DataLoader(dataset, bs, shuffle=False, drop_last=True)
            ran = _train_steps(trainer, controller, device, loader, optim, steps_per_trial)
            if device.type == 'cuda':
                torch.cuda.synchronize(device)
            if ran == 0:
                break
            best = bs
            bs *= 2
        except (RuntimeError, MemoryError) as e:
            if not _is_oom(e):
                raise
            if device.type == 'cuda':
                torch.cuda.empty_cache()
            break
    controller.load_state_dict(saved)
    return best


class LRFinderResult:
    def __init__(self, lrs, losses):
        self.results = {'lr': list(lrs), 'loss': list(losses)}

    def suggestion(self, skip_begin=10, skip_end=1):
        loss = np.asarray(self.results['loss'][skip_begin:len(self.results['loss']) - skip_end], dtype=np.float64)
        loss = loss[np.isfinite(loss)]
        if len(loss) < 2:
            return None
        return float(self.results['lr'][skip_begin + int(np.gradient(loss).argmin())])


def lr_find(trainer, controller, min_lr=1e-8, max_lr=1.0, num_training=100, mode='exponential', early_stop_threshold=4.0):
    if mode != 'exponential':
        raise ValueError("only mode='exponential' (the reference's default) is provided")
    device = trainer._setup(controller)
    saved = copy.deepcopy(controller.state_dict())
    optim = _unpack_optimizers(controller.configure_optimizers())[0]
    lrs, losses = [], []
    state = {'avg': 0.0, 'best': float('inf')}
    beta = 0.98

    def lr_of_step(i):
        return float(min_lr * (max_lr / min_lr) ** (i / max(1, num_training - 1)))

    def on_loss(i, value):
        state['avg'] = beta * state['avg'] + (1 - beta) * value
        smooth = state['avg'] / (1 - beta ** (i + 1))
        lrs.append(lr_of_step(i))
        losses.append(smooth)
        if early_stop_threshold is not None and i > 0 and smooth > early_stop_threshold * state['best']:
            return True
        state['best'] = min(state['best'], smooth)
        return not np.isfinite(smooth)

    _train_steps(trainer, controller, device, controller.train_dataloader(), optim, num_training, lr_of_step, on_loss)
    controller.load_state_dict(saved)
    return LRFinderResult(lrs, losses)
