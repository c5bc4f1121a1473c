"""Config loader and device / trainer selection with the reference's names and semantics
(/root/reference/utils/__init__.py:13-134): a config is a Python file executed once; every non-underscore global
becomes an attribute of a singleton `Config` with dict + attribute access; `configure_trainer(config, logger, dir)`
builds the trainer; `parse_gpus` / `get_strategy` turn `device` / `distributed_train` / `world_size` into a device list
and a data-parallel strategy.  No PyTorch-Lightning dependency: the trainer is engine.Trainer (a plain loop) and the
strategy is the RCCL flat-bucket DDP of engine/ddp.py."""
import inspect
import os
from importlib.util import module_from_spec, spec_from_file_location

import torch


class DictWrapper:
    def __init__(self, d=None):
        for k, v in (d or {}).items():
            if not inspect.ismodule(v):
                setattr(self, k, v)

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __iter__(self):
        return iter(self.__dict__)

    def __len__(self):
        return len(self.__dict__)

    def __contains__(self, key):
        return key in self.__dict__

    def __repr__(self):
        return 'DictWrapper: ' + repr(self.__dict__)

    def __getattr__(self, item):
        # missing attributes fall through to the dict API (get / items / keys / values …)
        return getattr(self.__dict__, item)


class _SingletonBase(type):
    _instances = {}

    def __call__(cls, *args, **kwargs):
        if cls not in cls._instances:
            cls._instances[cls] = super().__call__(*args, **kwargs)
        return cls._instances[cls]


class Config(DictWrapper, metaclass=_SingletonBase):
    def __repr__(self):
        return 'Config: ' + repr(self.__dict__)


def _exec_config(path):
    path = str(path)
    assert os.path.exists(path), path
    spec = spec_from_file_location('config', path)
    module = module_from_spec(spec)
    spec.loader.exec_module(module)
    return {k: getattr(module, k) for k in dir(module) if not k.startswith('_')}


def get_dict_wrapper(path) -> DictWrapper:
    return DictWrapper(_exec_config(path))


def get_config(path) -> Config:
    values = _exec_config(path)
    _SingletonBase._instances.pop(Config, None)
    return Config(values)


def get_gpus(world_size=1):
    n = torch.cuda.device_count()
    assert 0 <= world_size <= n, f"Only {n} are visible"
    return 0 if world_size == 0 else list(range(world_size))


def parse_gpus(cfg):
    """→ 0 (CPU), or a list of device indices (reference: utils/__init__.py:91-107)"""
    if cfg.get('distributed_train'):
        if isinstance(cfg.device, (list, tuple)):
            gpus = list(cfg.device)
            assert cfg.world_size == len(gpus), 'Not enough GPUs'
            return gpus
        return list(range(cfg.world_size))
    if cfg.device == 'cpu':
        return 0
    if cfg.device == 'cuda':
        return get_gpus()
    idx = int(str(cfg.device).split(':')[-1])
    if idx < torch.cuda.device_count():
        return [idx]
    return [0] if torch.cuda.is_available() else 0


def is_main_process() -> bool:
    return all(int(os.environ.get(k, 0)) == 0 for k in ('NODE_RANK', 'LOCAL_RANK', 'RANK'))


def get_strategy(config):
    """data-parallel settings (reference: DDPPlugin(find_unused_parameters, gradient_as_bucket_view), 114-119)"""
    if config.get('distributed_train', False):
        return dict(kind='ddp', bucket_mb=config.get('bucket_mb', 25),
                    find_unused_parameters=config.get('find_unused_parameters', False),
                    gradient_as_bucket_view=config.get('gradient_as_bucket_view', True))
    return None


def configure_trainer(config, lightning_logger=None, lightning_log_dir=None):
    from ..engine import Trainer
    return Trainer(gpus=parse_gpus(config), default_root_dir=lightning_log_dir, strategy=get_strategy(config),
                   max_epochs=config.n_epochs, logger=lightning_logger if lightning_logger is not None else False,
                   enable_checkpointing=True, callbacks=config.get('callbacks'), **config.get('trainer_kwargs', {}))


def find_max_batch_size(trainer, model):
    """reference utils/__init__.py:137-141 (PL tuner `scale_batch_size`) — see utils/tuner.py"""
    from .tuner import scale_batch_size
    new_batch_size = scale_batch_size(trainer, model, **model.config.get('find_max_batch_size_kwargs', {}))
    if new_batch_size is not None:
        print(f'Computed new batch size = {new_batch_size}')
        return new_batch_size


def find_optimal_init_lr(trainer, model):
    """reference utils/__init__.py:144-148 (PL tuner `lr_find(...).suggestion()`) — see utils/tuner.py"""
    from .tuner import lr_find
    new_lr = lr_find(trainer, model, **model.config.get('find_optimal_init_lr_kwargs', {})).suggestion()
    print(f'Computed new init lr = {new_lr}')
    return new_lr
