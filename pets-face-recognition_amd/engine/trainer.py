"""`Trainer` — a plain training / evaluation driver with the call surface the reference uses
(`Trainer(gpus=…, max_epochs=…, strategy=…, default_root_dir=…, logger=…, enable_checkpointing=…, callbacks=…,
**trainer_kwargs)`, `.fit(controller)`, `.test(controller)`), replacing the copy of PyTorch-Lightning's Trainer in
/root/reference/engine/trainer.py:66-652 and the custom loops of engine/loops/*.py.  Order of operations kept from the
reference: per batch zero_grad → training_step → backward → optimizer.step (PL automatic optimisation,
trainer.py:403-413); validation at the end of every epoch, then a barrier when distributed, then the LR-scheduler
step (loops/train_loop.py:13-38); evaluation outputs are handed over as List[dataloader][batch]
(loops/eval_loop.py:30-51); no sanity-validation steps, no sampler replacement, private BN statistics per rank
(trainer.py:105,110,118); one checkpoint (bare state_dict, reference key names) per epoch plus a `.trainer` sidecar
(epoch, global_step, optimizer_states, lr_schedulers — PL's checkpoint keys) that `resume_from_checkpoint=` /
`fit(ckpt_path=)` restart from (trainer.py:111,399; a PL-format checkpoint holding the same keys is read too).

Distributed = one process per GPU started by torchrun (RANK / LOCAL_RANK / WORLD_SIZE), RCCL all-reduce of the flat
gradient buffer in buckets overlapped with backward (engine/ddp.py)."""
import os
import time
from pathlib import Path

import torch


def _to_device(batch, device):
    if isinstance(batch, dict):
        return {k: _to_device(v, device) for k, v in batch.items()}
    if isinstance(batch, (list, tuple)):
        return type(batch)(_to_device(v, device) for v in batch)
    if torch.is_tensor(batch):
        return batch.to(device, non_blocking=True)
    return batch


def _batch_size(batch):
    if isinstance(batch, dict):
        for v in batch.values():
            if torch.is_tensor(v) and v.dim() > 0:
                return int(v.shape[0])
    if isinstance(batch, (list, tuple)) and batch and torch.is_tensor(batch[0]):
        return int(batch[0].shape[0])
    return 0


class Trainer:
    def __init__(self, gpus=0, default_root_dir=None, strategy=None, max_epochs=1, logger=False, enable_checkpointing=False,
                 callbacks=None, num_sanity_val_steps=0, limit_train_batches=None, limit_val_batches=None,
                 check_val_every_n_epoch=1, log_every_n_steps=50, benchmark=None, fast_dev_run=False, prefetch_batches=2,
                 resume_from_checkpoint=None, resume_weights_only=False, **_ignored):
        self.gpus, self.root, self.strategy = gpus, default_root_dir, strategy
        self.max_epochs = 1 if fast_dev_run else max_epochs
        self.logger = logger if logger else None
        self.enable_checkpointing = enable_checkpointing
        self.callbacks = callbacks or []
        self.limit_train_batches = 1 if fast_dev_run else limit_train_batches
        self.limit_val_batches = 1 if fast_dev_run else limit_val_batches
        self.check_val_every_n_epoch = check_val_every_n_epoch
        self.log_every_n_steps = log_every_n_steps
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.global_step = 0
        self.ddp = None
        self.resume_from_checkpoint = resume_from_checkpoint
        self.resume_weights_only = resume_weights_only   # a bare state dict without its .trainer sidecar restarts at epoch 0 instead of raising
        # batches copied to the device ahead of the step on a copy stream (data_loading/prefetch.py); 0: plain .to() per batch
        self.prefetch_batches = int(prefetch_batches)
        self.train_img_s = None      # end-to-end images/s of the last fit() (loader + copy + step), first 5 steps excluded

    # ------------------------------------------------------------------
    @property
    def is_distributed_run(self):
        return self.strategy is not None and self.world > 1

    def _device(self):
        if not self.gpus:
            return torch.device('cpu')
        if self.is_distributed_run:
            return torch.device('cuda', self.local_rank)
        return torch.device('cuda', self.gpus[0] if isinstance(self.gpus, (list, tuple)) else 0)

    def _setup(self, controller):
        device = self._device()
        if device.type == 'cuda':
            torch.cuda.set_device(device)
        p0 = next(controller.parameters(), None)
        if p0 is None or p0.device != device:
            controller.to(device)      # (a redundant .to() would make the HIP models drop and rebuild their engines)
        controller.logger = self.logger
        if self.is_distributed_run:
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
                dist.init_process_group('nccl' if device.type == 'cuda' else 'gloo')
            if device.type == 'cuda':
                from .ddp import FlatDDP
                eng = controller.model_loss.module.hip_engine(device)
                if self.ddp is None or self.ddp.eng is not eng:
                    # (re)bind: a new engine means new flat gradient buffers — the old reducer would all-reduce stale memory
                    if self.ddp is not None:
                        self.ddp.detach()
                    self.ddp = FlatDDP(controller.model_loss, bucket_mb=self.strategy.get('bucket_mb', 25))
            elif self.ddp is None or self.ddp.module is not controller.model_loss:
                # (re)bind: main.py builds a fresh Controller after the batch-size / lr finders — a reducer kept from the discarded
                # one would neither broadcast rank 0's new parameters nor all-reduce the new model's gradients
                from .ddp import GenericDDP
                self.ddp = GenericDDP(controller.model_loss)
        return device

    # ------------------------------------------------------------------ checkpoint / resume
    def _save_checkpoint(self, controller, optims, scheds, epoch):
        root = Path(self.root)
        root.mkdir(parents=True, exist_ok=True)
        torch.save(controller.state_dict(), root / f'epoch={epoch}.ckpt')
        torch.save({'epoch': epoch + 1, 'global_step': self.global_step,
                    'optimizer_states': [o.state_dict() for o in optims],
                    'lr_schedulers': [s.state_dict() for s in scheds]}, root / f'epoch={epoch}.ckpt.trainer')

    def _resume(self, controller, optims, scheds, path, device):
        """→ first epoch to run.  `path`: an `epoch=N.ckpt` of this trainer (state dict; loop state in its `.trainer` sidecar) or a
        pytorch-lightning checkpoint ({'state_dict', 'epoch', 'global_step', 'optimizer_states', 'lr_schedulers'}).  Like PL
        (trainer.py:306-308) a missing file raises, and training continues at the beginning of the next epoch."""
        path = Path(path)
        if not path.is_file():
            raise FileNotFoundError(f"resume_from_checkpoint: no checkpoint at {path}")
        # torch >= 2.6 unpickles with weights_only=True by default: fine for this trainer's own files (tensors, dicts, lists, numbers),
        # not for a pytorch-lightning checkpoint (callbacks, AttributeDict hyper-parameters, ...).  The checkpoint is the user's own file,
        # as it is for PL's `resume_from_checkpoint`: try the safe load first, fall back to the full unpickler and say so.
        def load(p_):
            import pickle
            try:
                return torch.load(str(p_), map_location='cpu', weights_only=True)
            except pickle.UnpicklingError as e:   # a global the safe unpickler does not allow; truncated files, I/O errors etc. propagate
                print(f'resume_from_checkpoint: {Path(p_).name} holds more than tensors ({e.__class__.__name__}: {str(e)[:120]}); '
                      f'loading it with the full unpickler (it can run code: only resume from files you wrote)')
                return torch.load(str(p_), map_location='cpu', weights_only=False)
        ckpt = load(path)
        if isinstance(ckpt, dict) and 'state_dict' in ckpt:
            sd, loop = ckpt['state_dict'], ckpt
            if 'optimizer_states' not in ckpt and 'epoch' not in ckpt:
                # {'state_dict': ...} alone / a PL `save_weights_only` file: the same policy as a bare state dict
                if not self.resume_weights_only:
                    raise ValueError(f"resume_from_checkpoint: {path.name} holds a state_dict but no loop state (epoch, optimizer, LR "
                                     f"schedule); pass Trainer(resume_weights_only=True) to restart at epoch 0 from these weights")
                loop = {}
        else:
            sd = ckpt
            side = Path(str(path) + '.trainer')
            if side.is_file():
                loop = load(side)
            elif self.resume_weights_only:
                loop = {}
            else:
                raise FileNotFoundError(f"resume_from_checkpoint: {path.name} is a bare state dict and its loop state {side.name} is missing "
                                        f"(epoch, optimizer, LR schedule); pass Trainer(resume_weights_only=True) to restart at epoch 0 from these weights")
        controller.load_state_dict(sd, strict=True)
        if loop and ('optimizer_states' in loop or 'lr_schedulers' in loop):
            ost, lst = loop.get('optimizer_states', []), loop.get('lr_schedulers', [])
            if len(ost) != len(optims) or len(lst) != len(scheds):
                raise ValueError(f"resume_from_checkpoint: the checkpoint holds {len(ost)} optimizer / {len(lst)} scheduler states, "
                                 f"configure_optimizers() built {len(optims)} / {len(scheds)}")
            for o, st in zip(optims, ost):
                o.load_state_dict(st)
            for s_, st in zip(scheds, lst):
                s_.load_state_dict(st)
        self.global_step = int(loop.get('global_step', 0))
        if 'epoch' not in loop:
            print(f'resume_from_checkpoint: {path.name} carries no loop state — weights only, starting at epoch 0')
        return int(loop.get('epoch', 0))

    # ------------------------------------------------------------------
    def fit(self, controller, ckpt_path=None):
        device = self._setup(controller)
        opt = controller.configure_optimizers()
        optims, scheds = (opt if isinstance(opt, (tuple, list)) and len(opt) == 2 and isinstance(opt[0], (list, tuple))
                          else ([opt], []))
        optim = optims[0]
        history = []
        first_epoch = 0
        ckpt_path = ckpt_path or self.resume_from_checkpoint
        if ckpt_path is not None:
            first_epoch = self._resume(controller, optims, scheds, ckpt_path, device)
            if self.ddp is not None and hasattr(self.ddp, 'broadcast_parameters'):
                self.ddp.broadcast_parameters()
        for epoch in range(first_epoch, self.max_epochs):
            controller.current_epoch = epoch
            controller.train()
            loader = controller.train_dataloader()
            if device.type == 'cuda' and self.prefetch_batches > 0:
                from ..data_loading.prefetch import DevicePrefetcher
                loader = DevicePrefetcher(loader, device, self.prefetch_batches, self.limit_train_batches)
            t_mark, n_mark = None, 0
            for bi, batch in enumerate(loader):
                if self.limit_train_batches is not None and bi >= self.limit_train_batches:
                    break
                if bi == 5:      # throughput clock: after the first steps (plan build, allocator warm-up)
                    if device.type == 'cuda':
                        torch.cuda.synchronize(device)
                    t_mark, n_mark = time.perf_counter(), 0
                batch = _to_device(batch, device)
                n_mark += _batch_size(batch)
                optim.zero_grad()
                loss = controller.training_step(batch, bi)
                loss.backward()
                if self.ddp is not None:
                    self.ddp.finish_backward()
                optim.step()
                self.global_step += 1
                if self.global_step % self.log_every_n_steps == 0 or bi == 0:
                    lv = float(loss.detach())
                    history.append(lv)
                    if self.rank == 0:
                        print(f'epoch {epoch} step {self.global_step} loss {lv:.5f}')
            if t_mark is not None and n_mark:
                if device.type == 'cuda':
                    torch.cuda.synchronize(device)
                self.train_img_s = n_mark * self.world / (time.perf_counter() - t_mark)
                if self.rank == 0:
                    print(f'epoch {epoch} train throughput {self.train_img_s:.1f} img/s (loader + copy + step, {self.world} process(es))')
            if (epoch + 1) % self.check_val_every_n_epoch == 0:
                self._run_eval(controller, device, 'val')
            if self.is_distributed_run:
                import torch.distributed as dist
                dist.barrier()
            for s in scheds:
                s.step()
            if self.enable_checkpointing and self.root is not None and self.rank == 0:
                self._save_checkpoint(controller, optims, scheds, epoch)
        self.loss_history = history
        return controller

    def _run_eval(self, controller, device, kind):
        if self.ddp is not None:
            self.ddp.sync_buffers()   # every rank evaluates with rank 0's BN statistics (torch DDP's broadcast_buffers)
        controller.eval()
        loaders = controller.val_dataloader() if kind == 'val' else controller.test_dataloader()
        if not isinstance(loaders, (list, tuple)):
            loaders = [loaders]
        outputs = []
        with torch.no_grad():
            for di, loader in enumerate(loaders):
                outs = []
                for bi, batch in enumerate(loader):
                    if self.limit_val_batches is not None and bi >= self.limit_val_batches:
                        break
                    batch = _to_device(batch, device)
                    step = controller.validation_step if kind == 'val' else controller.test_step
                    outs.append(step(batch, bi, di))
                outputs.append(outs)
            res = controller.validation_epoch_end(outputs) if kind == 'val' else controller.test_epoch_end(outputs)
        controller.train()
        return res

    def test(self, controller):
        device = self._setup(controller)
        return self._run_eval(controller, device, 'test')

    def validate(self, controller):
        device = self._setup(controller)
        return self._run_eval(controller, device, 'val')
