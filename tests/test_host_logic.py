"""Host-side logic: config loader, metrics, evaluator, fused-optimizer bookkeeping, trainer loop on CPU (config 1),
and the data-parallel gradient exchange with the gloo backend (world_size 2)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_loader_semantics(tmp_path):
    import pets_face_recognition_amd as pfr
    from pets_face_recognition_amd.utils import get_config, get_dict_wrapper, Config, parse_gpus, get_strategy
    f = tmp_path / "cfg.py"
    f.write_text("import os\n_hidden = 1\nn_epochs = 3\ndevice = 'cpu'\ndef model():\n    return 'm'\n")
    cfg = get_config(f)
    assert isinstance(cfg, Config) and cfg.n_epochs == 3 and cfg['device'] == 'cpu' and cfg.model() == 'm'
    assert '_hidden' not in cfg and 'os' not in cfg                      # underscore names and modules are dropped
    assert cfg.get('missing') is None and cfg.get('missing', 5) == 5     # dict API falls through
    assert dict(cfg.items())['n_epochs'] == 3
    cfg.output = 'x'
    assert cfg['output'] == 'x'
    cfg2 = get_config(f)                                                 # singleton is re-created per load
    assert 'output' not in cfg2
    assert parse_gpus(cfg2) == 0 and get_strategy(cfg2) is None
    w = get_dict_wrapper(f)
    assert w.n_epochs == 3 and not isinstance(w, Config)


def test_metrics_against_sklearn():
    sk = pytest.importorskip("sklearn.metrics")
    from pets_face_recognition_amd.engine import metrics as M
    g = torch.Generator().manual_seed(0)
    y = torch.randint(0, 2, (500,), generator=g)
    s = torch.rand(500, generator=g) * 0.5 + y * 0.3
    s[10] = s[11]
    assert abs(M.auroc(s, y) - sk.roc_auc_score(y.numpy(), s.numpy())) < 1e-9
    assert abs(M.average_precision(s, y) - sk.average_precision_score(y.numpy(), s.numpy())) < 1e-9
    fpr, tpr, thr = M.roc_curve(s, y)
    f2, t2, _ = sk.roc_curve(y.numpy(), s.numpy(), drop_intermediate=False)
    assert np.allclose(fpr.numpy(), f2) and np.allclose(tpr.numpy(), t2)
    acc, t = M.best_threshold_accuracy(s, y, thr, fpr, 1 - tpr)
    brute = max((((s > c) == (y == 1)).float().mean().item()) for c in s.tolist())
    assert acc <= brute + 1e-9 and acc > 0.5


def test_fused_optimizer_span_detection():
    from pets_face_recognition_amd.optim.fused import _FusedBase
    flat = torch.zeros(4096)
    a = flat[0:576].view(4, 3, 3, 16).permute(0, 3, 1, 2)      # channels-last view (engine parameter layout)
    assert _FusedBase._storage_span(a) == (a.data_ptr(), 576)
    assert _FusedBase._storage_span(flat[0:100:2]) is None       # strided: not dense
    # run building: neighbours merge across alignment padding (< 64 floats) only — a 64-element parameter of
    # ANOTHER group lying between two weights of this group (ResNet layer1: conv / bn.bias / conv) must not be swallowed
    from pets_face_recognition_amd.optim import FusedSGD
    master, grad = torch.zeros(1024), torch.zeros(1024)

    def par(lo, n):
        q = torch.nn.Parameter(master[lo:lo + n])
        q.grad = grad[lo:lo + n]
        return q
    w1, b, w2, w3 = par(0, 128), par(128, 64), par(192, 100), par(320, 64)     # w2 is followed by 28 floats of padding
    opt = FusedSGD([{"params": [w1, w2, w3], "lr": 0.1}, {"params": [b], "lr": 0.2}], 0.1, momentum=0.9)
    runs = opt._build_runs([w1, w2, w3])
    assert [(r["ps"] - master.data_ptr()) // 4 for r in runs] == [0, 192] and [r["n"] for r in runs] == [128, 192]
    assert [(q is w2, off) for q, off in runs[1]["members"]] == [(True, 0), (False, 128)]
    assert len(opt._build_runs([b])) == 1


def test_main_cpu_config1_runs(tmp_path):
    """BASELINE config 1: ResNet-18 + ArcFace, 100 ids, 224x224, bs=32, PyTorch CPU through main.py --config"""
    env = dict(os.environ, PFR_LIMIT_TRAIN_BATCHES="1")
    cfg = os.path.join(ROOT, "pets-face-recognition_amd", "configs", "synthetic", "fe_r18_cpu.py")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--config", cfg], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Completed!" in r.stdout and "Val Recall@K=10" in r.stdout and "Val ROC AUC" in r.stdout
    runs = list((tmp_path / "results").iterdir())
    assert len(runs) == 1 and (runs[0] / "checkpoints" / "epoch=0.ckpt").exists() and (runs[0] / "fe_r18_cpu.py").exists()
    imgs = sorted(p.name for p in (runs[0] / "img").iterdir())     # the reference's per-epoch figures (controller.py:185-203)
    assert "roc_0.png" in imgs and any(n.endswith("_confmat_0.png") for n in imgs), imgs
    sd = torch.load(runs[0] / "checkpoints" / "epoch=0.ckpt")
    assert "model_loss.module.layer1.0.conv1.weight" in sd and "model_loss.add_margin.weight" in sd   # reference key names


_DDP_SCRIPT = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, {root!r})
    from pets_face_recognition_amd.engine.ddp import BucketReducer, GenericDDP
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    # 1) bucketed flat reduction: suffixes become ready back to front, result = mean over ranks
    n = 1000
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = BucketReducer(flat, bucket_elems=256)
    for off in (900, 700, 650, 300, 0):
        red.ready(off)
    launched = list(red.launched)
    red.finish()
    expect = torch.arange(n, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    assert torch.allclose(flat, expect), (flat[:5], expect[:5])
    assert launched == [(744, 1000), (488, 744), (232, 488), (0, 232)], launched
    # a mark with at most bucket / 8 elements below it flushes what is final: the last collective (at mark 0) carries only that tail
    flat2 = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = BucketReducer(flat2, bucket_elems=256)
    for off in (900, 700, 300, 20, 0):
        red.ready(off)
    launched = list(red.launched)
    red.finish()
    assert torch.allclose(flat2, expect)
    assert launched == [(744, 1000), (488, 744), (232, 488), (20, 232), (0, 20)], launched
    # 2) 2 ranks x bs 4 with different data == mean of the per-rank gradients (replicated model, private BN stats)
    import pets_face_recognition_amd.models as M
    from pets_face_recognition_amd.losses import SoftmaxBasedMetricLearning
    torch.manual_seed(100 + rank)                     # different init per rank: DDP must broadcast rank 0's
    m = M.resnet18(); m.fc = torch.nn.Linear(512, 512)
    ml = SoftmaxBasedMetricLearning(m, 10, 512, is_focal=True, arc_margin=True)
    ddp = GenericDDP(ml)
    g = torch.Generator().manual_seed(7 + rank)
    x = torch.rand(4, 3, 32, 32, generator=g); y = torch.randint(0, 10, (4,), generator=g)
    ml(x, y)['loss'].backward()
    local = [p.grad.clone() for p in ml.parameters()]
    ddp.finish_backward()
    for p, l in zip(ml.parameters(), local):
        gathered = [torch.zeros_like(l) for _ in range(world)]
        dist.all_gather(gathered, l)
        assert torch.allclose(p.grad, sum(gathered) / world, atol=1e-6)
    w0 = [torch.zeros_like(ml.add_margin.weight) for _ in range(world)]
    dist.all_gather(w0, ml.add_margin.weight.data)
    assert torch.equal(w0[0], w0[1])
    dist.destroy_process_group()
    print('rank', rank, 'ok')
""")


def test_ddp_gloo_world2(tmp_path):
    script = tmp_path / "ddp_check.py"
    script.write_text(_DDP_SCRIPT.format(root=ROOT))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29631", str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.count("ok") == 2


_SHARD_SCRIPT = textwrap.dedent("""
    import sys, torch, torch.distributed as dist
    sys.path.insert(0, {root!r})
    from pets_face_recognition_amd.match import cosine_topk, cosine_topk_sharded
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    g = torch.Generator().manual_seed(3)
    q = torch.randn(40, 64, generator=g); gal = torch.randn(1001, 64, generator=g)
    gal[900] = gal[17]                                   # duplicate row across shards: tie → lower global index
    bounds = [0, 430, 1001]
    sc, idx = cosine_topk_sharded(q, gal[bounds[rank]:bounds[rank + 1]], 25, bounds[rank])
    ref_sc, ref_idx = cosine_topk(q, gal, 25)
    assert torch.allclose(sc, ref_sc, atol=1e-6) and torch.equal(idx, ref_idx.long()), (rank)
    dist.destroy_process_group()
    print('rank', rank, 'ok')
""")


def test_gallery_sharded_match_gloo_world2(tmp_path):
    script = tmp_path / "shard_check.py"
    script.write_text(_SHARD_SCRIPT.format(root=ROOT))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29633", str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.count("ok") == 2


_HOOK_SCRIPT = textwrap.dedent("""
    import sys, torch, torch.distributed as dist
    sys.path.insert(0, {root!r})
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    import pets_face_recognition_amd.models as M
    from pets_face_recognition_amd.models._fe_engine import flat_layout, grad_ready_marks
    from pets_face_recognition_amd.losses import SoftmaxBasedMetricLearning
    from pets_face_recognition_amd.engine.ddp import FlatDDP

    torch.manual_seed(50 + rank)
    m = M.resnet50(); m.fc = torch.nn.Linear(2048, 512)
    ml = SoftmaxBasedMetricLearning(m, 300, 512, is_focal=True, arc_margin=True)
    offs, total = flat_layout(m)
    marks = grad_ready_marks(m)
    assert marks[0] == offs['fc.weight'] and marks[-1] == 0 and all(a > b for a, b in zip(marks, marks[1:])), marks
    assert len(marks) == 1 + 16 + 1

    class FakeEngine:                       # what FlatDDP touches of an FEEngine: flat buffers + the grad-ready hook
        def __init__(self):
            self.master = torch.full((total,), float(rank + 1))
            self.grad = torch.zeros(total)
            self.stats = torch.full((64,), float(rank))
            self.nbt = torch.zeros(1, dtype=torch.int64)
            self.grad_ready_hook = None
    eng = FakeEngine()
    m.hip_engine = lambda device=None: eng
    ddp = FlatDDP(ml, bucket_mb=25)
    assert torch.all(eng.master == 1.0) and torch.all(eng.stats == 0.0)           # rank 0's replica everywhere
    w_all = [torch.zeros_like(ml.add_margin.weight) for _ in range(world)]
    dist.all_gather(w_all, ml.add_margin.weight.data)
    assert torch.equal(w_all[0], w_all[1])
    # event log: every collective with the gradient-final mark that was current when it was launched
    log, cur = [], [total]
    r = ddp.reducer
    orig = r.reduce_tensor
    def spy(view, lo=0, hi=0):
        log.append(('head',) if (lo == 0 and hi == 0) else ('bucket', lo, hi, cur[0]))
        return orig(view, lo, hi)
    r.reduce_tensor = spy
    for step in range(2):
        log.clear(); cur[0] = total
        eng.grad.zero_()
        ml.add_margin.weight.grad = None
        # head first (its gradient is complete before the backbone's backward starts) ...
        (ml.add_margin.weight * float(rank + 1)).sum().backward()
        # ... then the backbone: the engine finishes the flat buffer back to front and reports each mark
        hi = total
        for off in marks:
            eng.grad[off:hi] = torch.arange(off, hi, dtype=torch.float32) * (rank + 1)
            cur[0] = off
            eng.grad_ready_hook(off)
            hi = off
        ddp.finish_backward()
        assert log[0] == ('head',), log[:2]
        buckets = [e for e in log if e[0] == 'bucket']
        # only final data is sent; ranges tile [0, total) exactly once, back to front
        assert all(lo >= mark for _, lo, _, mark in buckets), buckets
        assert buckets[0][2] == total and buckets[-1][1] == 0
        assert all(a[1] == b[2] for a, b in zip(buckets, buckets[1:])), buckets
        assert 4 <= len(buckets) <= 7, len(buckets)             # ~25 MB buckets of a 98 MB buffer (+ the small tail collective)
        mean = sum(range(1, world + 1)) / world
        assert torch.allclose(eng.grad, torch.arange(total, dtype=torch.float32) * mean)
        assert torch.allclose(ml.add_margin.weight.grad, torch.full_like(ml.add_margin.weight, mean))
    # a model that got a new engine must be detected, not silently skipped
    eng2 = FakeEngine(); m.hip_engine = lambda device=None: eng2
    try:
        ddp.finish_backward(); raise SystemExit('stale engine not detected')
    except RuntimeError:
        pass
    dist.destroy_process_group()
    print('rank', rank, 'ok')
""")


def test_flat_ddp_hook_ordering_gloo_world2(tmp_path):
    """FlatDDP (the class that runs on the MI355X node) driven on CPU/gloo with world size 2 by a fake engine that reports the
    REAL ResNet-50 grad-ready marks: head all-reduce first, buckets cover every element exactly once with final data only,
    result = mean over ranks, stale-engine detection.  Reference: /root/reference/utils/__init__.py:114-119."""
    script = tmp_path / "hook_check.py"
    script.write_text(_HOOK_SCRIPT.format(root=ROOT))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29637", str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.count("ok") == 2


def test_device_prefetcher_order_limit_and_errors():
    """data_loading/prefetch.py on the CPU path and its contract: same batches in the same order as the loader, `limit` respected,
    a loader exception reaches the consumer, early exit of the consumer stops the feeder"""
    from pets_face_recognition_amd.data_loading import DevicePrefetcher
    batches = [{'x': torch.full((4, 3), float(i)), 'label': torch.arange(4) + i, 'tag': i} for i in range(7)]
    got = list(DevicePrefetcher(batches, 'cpu', depth=2))
    assert [b['tag'] for b in got] == list(range(7)) and all(torch.equal(a['x'], b['x']) for a, b in zip(got, batches))
    assert len(list(DevicePrefetcher(batches, 'cpu', depth=2, limit=3))) == 3
    assert len(DevicePrefetcher(batches, 'cpu', limit=3)) == 3

    def bad():
        yield batches[0]
        raise ValueError("decode failed")
    with pytest.raises(ValueError):
        list(DevicePrefetcher(bad(), 'cpu'))


def test_synthetic_dataset_noise_bank_is_deterministic_and_labelled():
    from pets_face_recognition_amd.data_loading import SyntheticRecDataset
    a = SyntheticRecDataset(5, 3, 32, seed=1, raw_uint8=True, noise_bank=4)
    b = SyntheticRecDataset(5, 3, 32, seed=1, raw_uint8=True, noise_bank=4)
    assert torch.equal(a[7]['x'], b[7]['x']) and a[7]['x'].dtype == torch.uint8 and a[7]['x'].shape == (32, 32, 3)
    assert int(a[7]['label']) == 2 and not torch.equal(a[7]['x'], a[8]['x'])


def test_main_runs_batch_size_and_lr_finders(tmp_path):
    """reference main.py:79-89: config.find_max_batch_size / find_optimal_init_lr -> utils.find_max_batch_size (power scaling) and
    utils.find_optimal_init_lr (exponential sweep, steepest-descent suggestion); the found values reach the loaders / optimizer"""
    cfg = os.path.join(ROOT, "pets-face-recognition_amd", "configs", "synthetic", "fe_r18_cpu_tuned.py")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--config", cfg], cwd=tmp_path, env=dict(os.environ),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Computed new batch size = 16" in r.stdout      # 4 -> 8 -> 16: three trials
    lr = float(r.stdout.split("Computed new init lr = ")[1].split()[0])
    assert 1e-5 <= lr <= 1.0
    assert "Completed!" in r.stdout


def test_lr_finder_suggestion_is_the_steepest_descent():
    from pets_face_recognition_amd.utils.tuner import LRFinderResult
    lrs = list(np.geomspace(1e-6, 1, 40))
    loss = [5.0] * 15 + list(np.linspace(5.0, 1.0, 10)) + [1.0] * 5 + list(np.linspace(1.0, 9.0, 10))
    loss[18] -= 0.6
    s = LRFinderResult(lrs, loss).suggestion()
    assert s in (lrs[17], lrs[18]) and LRFinderResult(lrs[:5], loss[:5]).suggestion() is None


def _tiny_config(lr_milestones=(1, 2)):
    """a config object with the reference's contract on a tiny CPU model (deterministic loaders)"""
    from pets_face_recognition_amd.losses import SoftmaxBasedMetricLearning
    from pets_face_recognition_amd.utils import get_dict_wrapper  # noqa: F401
    g = torch.Generator().manual_seed(0)
    xs = torch.rand(24, 3, 8, 8, generator=g)
    ys = torch.randint(0, 6, (24,), generator=g)
    data = [{"x": xs[i:i + 8], "label": ys[i:i + 8], "index": torch.arange(i, i + 8)} for i in range(0, 24, 8)]

    class Cfg(dict):
        __getattr__ = dict.get

    cfg = Cfg()

    def model():
        return torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(192, 512))

    def loss(config, m):
        return SoftmaxBasedMetricLearning(m, 6, 512, is_focal=True, arc_margin=True)

    def optimizer(ml):
        o = torch.optim.SGD([{"params": list(ml.module.parameters()), "lr": 0.05},
                             {"params": list(ml.add_margin.parameters()), "lr": 0.1, "weight_decay": 1e-4}], 0.01, momentum=0.9)
        return [o], [torch.optim.lr_scheduler.MultiStepLR(o, milestones=list(lr_milestones), gamma=0.1)]

    cfg.update(model=model, loss=loss, optimizer=optimizer, train_dataloader=lambda: data, val_dataloader=lambda: data, n_epochs=3)
    return cfg


def test_resume_from_checkpoint_continues_the_run(tmp_path):
    """reference engine/trainer.py:111,399 (`resume_from_checkpoint` → CheckpointConnector): a run restarted from the epoch-0
    checkpoint ends with the weights, momentum buffers, scheduler state and step counter of the uninterrupted 3-epoch run; a
    missing file raises like PL's; a pytorch-lightning-format checkpoint is read too."""
    from pets_face_recognition_amd.engine import Trainer
    from pets_face_recognition_amd.engine.controller import Controller
    kw = dict(gpus=0, max_epochs=3, enable_checkpointing=True, check_val_every_n_epoch=100, prefetch_batches=0)
    torch.manual_seed(1)
    a = Controller(_tiny_config())
    ta = Trainer(default_root_dir=tmp_path / "a", **kw)
    ta.fit(a)
    assert ta.global_step == 9 and (tmp_path / "a" / "epoch=0.ckpt.trainer").exists()
    sd0 = torch.load(tmp_path / "a" / "epoch=0.ckpt")
    assert "model_loss.add_margin.weight" in sd0                  # the weights file stays a bare reference-named state dict
    torch.manual_seed(2)                                           # a different initialisation: everything must come from the file
    b = Controller(_tiny_config())
    tb = Trainer(default_root_dir=tmp_path / "b", resume_from_checkpoint=tmp_path / "a" / "epoch=0.ckpt", **kw)
    tb.fit(b)
    assert tb.global_step == 9
    assert not (tmp_path / "b" / "epoch=0.ckpt").exists() and (tmp_path / "b" / "epoch=2.ckpt").exists()   # epochs 1, 2 ran
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(va, vb), k
    sa = torch.load(tmp_path / "a" / "epoch=2.ckpt.trainer"); sb = torch.load(tmp_path / "b" / "epoch=2.ckpt.trainer")
    assert sa["epoch"] == sb["epoch"] == 3 and sa["lr_schedulers"] == sb["lr_schedulers"]
    for x, y in zip(sa["optimizer_states"][0]["state"].values(), sb["optimizer_states"][0]["state"].values()):
        assert torch.equal(x["momentum_buffer"], y["momentum_buffer"])
    # pytorch-lightning layout: one file with state_dict + loop state; fit(ckpt_path=) form
    st = torch.load(tmp_path / "a" / "epoch=1.ckpt.trainer")
    torch.save(dict(st, state_dict=torch.load(tmp_path / "a" / "epoch=1.ckpt")), tmp_path / "pl.ckpt")
    c = Controller(_tiny_config())
    tc = Trainer(default_root_dir=tmp_path / "c", **kw)
    tc.fit(c, ckpt_path=tmp_path / "pl.ckpt")
    for (k, va), (_, vc) in zip(a.state_dict().items(), c.state_dict().items()):
        assert torch.equal(va, vc), k
    with pytest.raises(FileNotFoundError):
        Trainer(default_root_dir=tmp_path / "d", resume_from_checkpoint=tmp_path / "nope.ckpt", **kw).fit(Controller(_tiny_config()))
    # {'state_dict': ...} without loop state (PL `save_weights_only`): the bare-state-dict policy — refused unless resume_weights_only
    torch.save({"state_dict": torch.load(tmp_path / "a" / "epoch=2.ckpt")}, tmp_path / "wonly.ckpt")
    with pytest.raises(ValueError, match="no loop state"):
        Trainer(default_root_dir=tmp_path / "e", resume_from_checkpoint=tmp_path / "wonly.ckpt", **kw).fit(Controller(_tiny_config()))
    e = Controller(_tiny_config())
    te = Trainer(default_root_dir=tmp_path / "e", resume_from_checkpoint=tmp_path / "wonly.ckpt", resume_weights_only=True,
                 **dict(kw, max_epochs=0))
    te.fit(e)
    assert te.global_step == 0
    for (k, va), (_, ve) in zip(a.state_dict().items(), e.state_dict().items()):
        assert torch.equal(va, ve), k
    # a truncated file is an error of its own, not a reason to try the full unpickler
    blob = (tmp_path / "a" / "epoch=0.ckpt").read_bytes()
    (tmp_path / "trunc.ckpt").write_bytes(blob[: len(blob) // 2])
    with pytest.raises(Exception) as ei:
        Trainer(default_root_dir=tmp_path / "f", resume_from_checkpoint=tmp_path / "trunc.ckpt", **kw).fit(Controller(_tiny_config()))
    assert not isinstance(ei.value, (ValueError, FileNotFoundError)) or "loop state" not in str(ei.value)


_REBIND_SCRIPT = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))
    from test_host_logic import _tiny_config
    from pets_face_recognition_amd.engine import Trainer
    from pets_face_recognition_amd.engine.controller import Controller
    rank = int(os.environ['RANK'])
    tr = Trainer(gpus=0, max_epochs=1, strategy=dict(kind='ddp'), check_val_every_n_epoch=100, prefetch_batches=0)
    torch.manual_seed(10 + rank)
    old = Controller(_tiny_config())
    tr._setup(old)                                   # what the batch-size / lr finders do (utils/tuner.py) ...
    first = tr.ddp
    torch.manual_seed(20 + rank)                     # ... then main.py builds a FRESH controller (different init per rank here)
    new = Controller(_tiny_config())
    tr.fit(new)
    assert tr.ddp is not first and tr.ddp.module is new.model_loss
    for k, v in new.state_dict().items():
        both = [torch.zeros_like(v) for _ in range(2)]
        dist.all_gather(both, v)
        assert torch.equal(both[0], both[1]), k      # rank 0's parameters were broadcast and the gradients averaged
    dist.destroy_process_group()
    print('rank', rank, 'ok')
""")


def test_generic_ddp_rebinds_to_the_controller_built_after_the_tuners_gloo_world2(tmp_path):
    """ADVICE r3 (main.py:59): after find_max_batch_size / find_optimal_init_lr main.py builds a fresh Controller; on the torch / gloo
    path the trainer must bind its reducer to THAT module (broadcast + all-reduce), not keep the one of the discarded controller."""
    script = tmp_path / "rebind_check.py"
    script.write_text(_REBIND_SCRIPT.format(root=ROOT))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29641", str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.count("ok") == 2


def test_bench_launches_its_own_ranks_when_no_launcher_is_around_it():
    """`python bench.py --gpus N` (the driver's N = 1 form with another N) must start N ranks itself: the launcher path is
    exercised here with the GPU-free self test — two ranks under torch.distributed.run on 127.0.0.1, ONE JSON line from rank 0."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--launcher-selftest"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec == {"launcher_selftest": True, "n_gpus": 2, "rank_sum": 1.0, "steps": 3, "warmup": 1}
    # a rank count that disagrees with the launcher's world size is an error, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], capture_output=True, text=True, timeout=300, env=env2)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_hand_counted_vmcnt_kernels_use_no_scratch(tmp_path):
    """pfr_slin.hip counts its vector-memory operations by hand (inline-asm buffer loads behind `s_waitcnt vmcnt(n)`); a
    compiler-inserted scratch spill / reload between them would shift the count silently.  The ISA of the SHIPPED flags (csrc/build.sh)
    must therefore show no private segment and no scratch instruction for any slin_kernel instantiation (the launch path checks the same
    through hipFuncGetAttributes and falls back to the tile kernel otherwise)."""
    import re
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    csrc = os.path.join(ROOT, "pets-face-recognition_amd", "csrc")
    flags = re.search(r'^FLAGS="([^"]*)"', open(os.path.join(csrc, "build.sh")).read(), re.M).group(1)
    flags = flags.replace("$ARCH", "gfx950").replace("${PFR_EXTRA_FLAGS}", "").split()
    out = tmp_path / "pfr_slin.s"
    subprocess.run(["hipcc", *flags, "--cuda-device-only", "-S", os.path.join(csrc, "pfr_slin.hip"), "-o", str(out)], check=True,
                   stderr=subprocess.DEVNULL)
    text = out.read_text()
    kernels = re.findall(r"\.name:\s+(\S*slin_kernel\S*)\s+\.private_segment_fixed_size:\s+(\d+)", text)
    assert len(kernels) >= 15, kernels
    assert all(int(sz) == 0 for _, sz in kernels), [k for k in kernels if int(k[1])]
    assert not re.search(r"^\s+scratch_(load|store)", text, re.M)
