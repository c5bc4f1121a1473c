"""`Controller` — the model + loss + evaluation unit, with the method names of the reference's LightningModule
(/root/reference/engine/controller.py:14-246) but no PyTorch-Lightning dependency:

  Controller(config)                      model_loss = config.loss(config, config.model())        (17-22)
  training_step(batch, idx) -> loss       model_loss(batch['x'], batch['label'])['loss']           (27-29)
  validation_step / test_step -> dict     {'emb', 'label', 'index'}                                (31-35, 42-46)
  validation_epoch_end / test_epoch_end   outputs is List[dataloader][batch] (engine/loops/eval_loop.py:30-51)
  _evaluate, compute_accuracy, *_dataloader, configure_optimizers                                   (95-246)

The O(N²) python pair loop of the reference (77-90, 143-160: ≈ 8.4 µs per scored pair) is replaced by the match module
(one MFMA GEMM per gallery chunk + running top-K); the metrics dict has the reference's keys and threshold rules (pinned to the
reference's own `_evaluate` run, tests/golden/evaluate.npz).
Unlike the reference (39: `self.logger.run_id` unconditionally), a missing logger is fine."""
import json
import time
from pathlib import Path

import torch
import torch.nn as nn

from . import metrics as M
from ..match import recall_at_k, pair_similarity


class Controller(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        model = self.config.model()
        self.model_loss = self.config.loss(config, model)
        self.hparams = {k: repr(v) for k, v in config.items()}
        self.logger = None
        self.current_epoch = 0
        self.last_metrics = {}

    def forward(self, *args, **kwargs):
        return self.model_loss(*args, **kwargs)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, config=None, map_location='cpu', strict=True, **_):
        """`Controller.load_from_checkpoint(path, config=get_dict_wrapper(cfg.py))` — how the reference's inference consumers get
        their FE models (generate_tsv.py:158-176, eval_fe_*.py).  Reads this package's checkpoints (a plain state dict with the
        reference's key names) and pytorch-lightning ones ({'state_dict': …}); the result is in eval()-ready train mode like PL's."""
        if config is None:
            raise ValueError("load_from_checkpoint needs config=… (the module is rebuilt from config.model() / config.loss())")
        ckpt = torch.load(str(checkpoint_path), map_location=map_location)
        sd = ckpt.get('state_dict', ckpt) if isinstance(ckpt, dict) else ckpt
        self = cls(config)
        self.load_state_dict(sd, strict=strict)
        return self

    # ------------------------------------------------------------------ steps
    def _images(self, x, train):
        """uint8 [N, H, W, 3] batches are raw frames: the config's device-side Compose pipeline (the reference's
        train_augmentation / val_augmentation, fe_dogs_config.py:17-32) turns them into the float NCHW batch"""
        if x.dtype != torch.uint8:
            return x
        aug = self.config.get('device_train_augmentation' if train else 'device_val_augmentation')
        if aug is None:
            raise ValueError("uint8 image batch but the config defines no device_train_augmentation / device_val_augmentation")
        return aug(x)

    def training_step(self, batch, batch_idx=0):
        return self.model_loss(self._images(batch['x'], True), batch['label'])['loss']

    def validation_step(self, batch, batch_idx=0, dataset_idx=0):
        return {'emb': self.model_loss(self._images(batch['x'], False)), 'label': batch['label'], 'index': batch['index']}

    def test_step(self, batch, batch_idx=0, dataset_idx=0):
        return self.validation_step(batch, batch_idx, dataset_idx)

    # ------------------------------------------------------------------ evaluation
    @staticmethod
    def _gather(outputs_i):
        emb = torch.cat([o['emb'] for o in outputs_i], dim=0)
        classes = torch.cat([o['label'] for o in outputs_i], dim=0)
        indices = torch.cat([o['index'] for o in outputs_i], dim=0)
        order = torch.argsort(indices)
        return emb[order].float(), classes[order]

    def _pair_scores(self, emb, pair_generator):
        ci = pair_generator.corrected_indices
        ia = [a for a, _ in ci]
        ib = [b for _, b in ci]
        sim = getattr(self.config, 'similarity_f', None)
        if sim is not None and not getattr(sim, '_is_default_cosine', False):
            # a custom similarity is always honoured (the reference calls config.similarity_f unconditionally,
            # controller.py:62) — on CUDA embeddings too; only the flagged default (cos+1)/2 takes the fused kernel
            scores = sim([(emb[a], emb[b]) for a, b in ci])
        else:
            scores = pair_similarity(emb, ia, ib)                       # (cos + 1) / 2, fe_dogs_config.py:89-93
        # CUDA scores stay on the device: the sort / scan of the metric suite runs there (engine/metrics.py, pfr_pair_curve)
        return scores.float(), torch.as_tensor(pair_generator.labels)

    def _recall(self, emb, classes, ks):
        dt = torch.float32 if not emb.is_cuda else getattr(self.config, 'match_dtype', torch.bfloat16)
        return recall_at_k(emb, classes, tuple(ks), compute_dtype=dt)

    def test_epoch_end(self, outputs):
        all_metrics = {}
        for i in range(len(outputs)):
            emb, classes = self._gather(outputs[i])
            name, pair_generator = self.config.pair_generator(i)
            scores, labels = self._pair_scores(emb, pair_generator)
            ps = M.PairStats(scores, labels)
            metrics = {'ROC AUC': ps.auroc(), 'Accuracy': ps.best_threshold_accuracy()[0]}
            rk = self._recall(emb, classes, [10, 100])
            metrics.update({f'Recall@K={k}': (x / y if y else float('nan')) for k, (x, y) in rk.items()})
            print('', *[f'{name} {k}\t{v}' for k, v in metrics.items()], sep='\n')
            all_metrics[name] = metrics
        self.last_metrics = all_metrics
        return all_metrics

    def validation_epoch_end(self, outputs):
        m = self._evaluate(outputs)
        if self.logger is not None and hasattr(self.logger, 'log_artifacts'):
            self.logger.log_artifacts(str(self.config.output))
        return m

    def _evaluate(self, outputs):
        """The metrics dict of the reference's `_evaluate` (controller.py:95-183): the same keys in the same order
        ('ROC AUC', 'AveragePrecision', 'Accuracy', 'Opt thr', f'Accuracy thr={thr}' / 'Precision thr=…' / 'Recall thr=…' over
        config.thrs, f'Recall@K={k}' over config.k, f'TAR@FAR={far}' + f'TH@FAR={far}', f'TRR@FRR={frr}' + f'TH@FRR={frr}') and the
        same threshold rules (engine/metrics.py) — pinned key for key to the reference's own run, tests/golden/evaluate.npz."""
        cfg = self.config
        all_metrics = {}
        self.last_confmat = {}
        rocs = []
        for i in range(len(outputs)):
            emb, classes = self._gather(outputs[i])
            name, pair_generator = cfg.pair_generator(i)
            scores, labels = self._pair_scores(emb, pair_generator)
            ps = M.PairStats(scores, labels)
            fpr, tpr, thr = ps.roc()
            opt_thr = ps.opt_threshold()
            cm = ps.stats_at(opt_thr)
            confmat = [[cm['tn'], cm['fp']], [cm['fn'], cm['tp']]]          # [target][prediction], torchmetrics.ConfusionMatrix
            print(name, f'\nConf Mat thr = {opt_thr}', confmat)
            metrics = {'ROC AUC': ps.auroc(), 'AveragePrecision': ps.average_precision(),
                       'Accuracy': ps.best_threshold_accuracy(thr, fpr, 1 - tpr)[0], 'Opt thr': opt_thr}
            thrs = cfg.get('thrs', ())
            at = [ps.stats_at(float(t)) for t in thrs]
            metrics.update((f'Accuracy thr={t}', st['accuracy']) for t, st in zip(thrs, at))
            metrics.update((f'Precision thr={t}', st['precision']) for t, st in zip(thrs, at))
            metrics.update((f'Recall thr={t}', st['recall']) for t, st in zip(thrs, at))
            ks = list(cfg.get('k', ()))
            if len(ks):
                rk = self._recall(emb, classes, ks)
                metrics.update({f'Recall@K={k}': (x / y if y else float('nan')) for k, (x, y) in rk.items()})
            for far, (tar, th) in ps.far_points(cfg.get('far_thr', ())).items():
                metrics[f'TAR@FAR={far}'] = tar
                metrics[f'TH@FAR={far}'] = th
            for frr, (trr, th) in ps.frr_points(cfg.get('frr_thr', ())).items():
                metrics[f'TRR@FRR={frr}'] = trr
                metrics[f'TH@FRR={frr}'] = th
            all_metrics[name] = metrics
            self.last_confmat[name] = confmat
            print(*[f'{name} {k}\t{v}' for k, v in metrics.items()], sep='\n')
            self._plot_confmat(name, cm)
            self._log(name, metrics)
            rocs.append((fpr.numpy(), tpr.numpy(), metrics['ROC AUC'], name))
        self._plot_rocs(rocs)
        self.last_metrics = all_metrics
        return all_metrics

    # the two figures the reference's _evaluate writes per validation epoch (controller.py:185-203); skipped when matplotlib
    # is not installed
    def _img_dir(self):
        d = Path(self.config.get('img_dir', '.'))
        d.mkdir(parents=True, exist_ok=True)
        return d

    def _plot_confmat(self, name, cm):
        try:   # an Agg canvas of its own: library code must not switch the process-wide matplotlib backend
            from matplotlib.figure import Figure
            from matplotlib.backends.backend_agg import FigureCanvasAgg
        except ImportError:
            return
        mat = [[cm['tn'], cm['fp']], [cm['fn'], cm['tp']]]
        fig = Figure()
        FigureCanvasAgg(fig)
        ax = fig.subplots()
        ax.imshow(mat, cmap='viridis')
        for r in range(2):
            for c in range(2):
                ax.text(c, r, str(mat[r][c]), ha='center', va='center', color='w')
        ax.set_xticks([0, 1]); ax.set_yticks([0, 1])
        ax.set_xlabel('Predicted label'); ax.set_ylabel('True label')
        fig.savefig(self._img_dir() / f' {name}_confmat_{self.current_epoch}.png')

    def _plot_rocs(self, rocs):
        try:
            from matplotlib.figure import Figure
            from matplotlib.backends.backend_agg import FigureCanvasAgg
        except ImportError:
            return
        fig = Figure(figsize=(10, 10))
        FigureCanvasAgg(fig)
        ax = fig.subplots()
        for fpr, tpr, auc, name in rocs:
            ax.plot(fpr, tpr, label=f'{name} AUC = {auc}', linewidth=3)
        ax.plot([0, 1], [0, 1], 'k--', linewidth=3)
        ax.set_xlabel('False positive rate'); ax.set_ylabel('True positive rate')
        ax.set_title('ROC curves'); ax.grid(); ax.legend()
        fig.savefig(self._img_dir() / f'roc_{self.current_epoch}.png')

    def _log(self, name, metrics):
        if self.logger is not None and hasattr(self.logger, 'log_metrics'):
            self.logger.log_metrics({f'{name} {k}': v for k, v in metrics.items()}, step=self.current_epoch)
        out = self.config.get('output', None)
        if out is not None:
            try:
                Path(out).mkdir(parents=True, exist_ok=True)
                with open(Path(out) / 'metrics.jsonl', 'a') as f:
                    f.write(json.dumps({'time': time.time(), 'epoch': self.current_epoch, 'set': name, **metrics}) + '\n')
            except OSError:
                pass

    @staticmethod
    def compute_accuracy(scores, labels, thresholds, fpr, fnr):
        return M.best_threshold_accuracy(scores, labels, thresholds, fpr, fnr)[0]

    # ------------------------------------------------------------------ delegation to the config
    def train_dataloader(self):
        return self.config.train_dataloader()

    def val_dataloader(self):
        return self.config.val_dataloader()

    def test_dataloader(self):
        return self.config.test_dataloader() if 'test_dataloader' in self.config else self.config.val_dataloader()

    def configure_optimizers(self):
        return self.config.optimizer(self.model_loss)
