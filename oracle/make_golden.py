"""Generates tests/golden/*.npz by running the REFERENCE's own Python modules (imported from /root/reference in the
build container) and checks the oracle restatement against them.  Run: `python oracle/make_golden.py`.

The reference tree does not exist on the GPU box: only the resulting data fixtures (inputs + expected outputs) and
this script are committed.  What is imported / executed from the reference:
  losses/__init__.py, losses/large_margin.py, losses/losses.py   (as-is)
  models/swin.py                                                 (as-is)
  engine/controller.py  Controller.test_epoch_end                (with sys.modules stubs for pytorch_lightning /
                                                                  torchmetrics, which are not installed)
  configs/dog_fe/fe_dogs_config.py  similarity_f                 (that one function, extracted with ast)
The ResNet backbone is third-party to the reference (torchvision, not installable here); `resnet_hf.npz` pins its
restatement against the independent implementation in Hugging Face transformers (`python oracle/make_golden.py resnet_hf`).
"""
import ast
import contextlib
import importlib.util
import io
import os
import re
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import arcface_ref, match_ref, resnet_ref  # noqa: E402


def load_ref_module(name, relpath, package_path=None):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath),
                                                  submodule_search_locations=package_path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def ref_losses():
    return load_ref_module("ref_losses", "losses/__init__.py", [os.path.join(REF, "losses")])


def ref_similarity_f():
    src = open(os.path.join(REF, "configs/dog_fe/fe_dogs_config.py")).read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "similarity_f")
    ns = {"torch": torch, "F": torch.nn.functional}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "fe_dogs_config.similarity_f", "exec"), ns)
    return ns["similarity_f"]


def _tm_binary_clf_curve(preds, target):
    """torchmetrics.functional.classification.precision_recall_curve._binary_clf_curve (0.7 - 0.11, the releases contemporary with
    the reference's pytorch-lightning==1.5.9 pin; torchmetrics itself is unpinned and not installable here), restated from its
    published algorithm: descending sort, one operating point per run of equal scores, integer cumulative counts."""
    order = torch.argsort(preds, descending=True)
    preds, target = preds[order], target[order]
    distinct = torch.where(preds[1:] - preds[:-1])[0]
    idx = torch.nn.functional.pad(distinct, [0, 1], value=target.size(0) - 1)
    target = (target == 1).to(torch.long)
    tps = torch.cumsum(target, dim=0)[idx]
    fps = 1 + idx - tps
    return fps, tps, preds[idx]


def torchmetrics_standins():
    """The eight torchmetrics classes engine/controller.py:12 imports, as test-side stand-ins that follow torchmetrics' published
    binary-input semantics (float scores in [0, 1] against 0/1 targets, `preds >= threshold`, float32 curves).  Each one is
    cross-checked against scikit-learn by gen_evaluate()."""
    class ROC:
        def __call__(self, s, l):
            fps, tps, thr = _tm_binary_clf_curve(s, l)
            tps = torch.cat([torch.zeros(1, dtype=tps.dtype), tps])
            fps = torch.cat([torch.zeros(1, dtype=fps.dtype), fps])
            thr = torch.cat([thr[0][None] + 1, thr])
            return fps / fps[-1], tps / tps[-1], thr

    class AUROC:
        def __call__(self, s, l):
            fpr, tpr, _ = ROC()(s, l)
            return torch.trapz(tpr, fpr)

    class AveragePrecision:
        def __call__(self, s, l):
            fps, tps, _ = _tm_binary_clf_curve(s, l)
            precision = tps / (tps + fps)
            recall = tps / tps[-1]
            last = int(torch.where(tps == tps[-1])[0][0])
            precision = torch.cat([precision[:last + 1].flip(0), torch.ones(1)])
            recall = torch.cat([recall[:last + 1].flip(0), torch.zeros(1)])
            return -torch.sum((recall[1:] - recall[:-1]) * precision[:-1])

    def stat(s, l, thr):
        pred = (s >= thr).int()
        tp = ((pred == 1) & (l == 1)).sum()
        fp = ((pred == 1) & (l == 0)).sum()
        tn = ((pred == 0) & (l == 0)).sum()
        fn = ((pred == 0) & (l == 1)).sum()
        return tp, fp, tn, fn

    class _Thr:
        def __init__(self, *a, threshold=0.5, **k):
            self.threshold = threshold

    class StatScores(_Thr):
        def __call__(self, s, l):
            tp, fp, tn, fn = stat(s, l, self.threshold)
            return torch.stack([tp, fp, tn, fn, tp + fn])

    class Accuracy(_Thr):
        def __call__(self, s, l):
            tp, fp, tn, fn = stat(s, l, self.threshold)
            return (tp + tn) / (tp + tn + fp + fn)

    class Precision(_Thr):
        def __call__(self, s, l):
            tp, fp, tn, fn = stat(s, l, self.threshold)
            return tp / (tp + fp) if int(tp + fp) else torch.tensor(0.0)     # zero_division = 0

    class Recall(_Thr):
        def __call__(self, s, l):
            tp, fp, tn, fn = stat(s, l, self.threshold)
            return tp / (tp + fn) if int(tp + fn) else torch.tensor(0.0)

    class ConfusionMatrix(_Thr):
        def __init__(self, num_classes, threshold=0.5):
            super().__init__(threshold=threshold)

        def __call__(self, s, l):
            tp, fp, tn, fn = stat(s, l, self.threshold)
            return torch.stack([torch.stack([tn, fp]), torch.stack([fn, tp])])     # [target][prediction]

    return dict(ROC=ROC, AUROC=AUROC, AveragePrecision=AveragePrecision, StatScores=StatScores, Accuracy=Accuracy,
                Precision=Precision, Recall=Recall, ConfusionMatrix=ConfusionMatrix)


def ref_controller(full=False):
    """engine/controller.py with stand-ins for the two missing third-party packages (test-side stubs only).  full=True: all
    eight torchmetrics classes (what Controller._evaluate needs), else the AUROC / ROC pair test_epoch_end needs."""
    from sklearn.metrics import roc_auc_score, roc_curve

    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = torch.nn.Module
    lg = types.ModuleType("pytorch_lightning.loggers")
    lg.MLFlowLogger = object
    ut = types.ModuleType("pytorch_lightning.utilities")
    ty = types.ModuleType("pytorch_lightning.utilities.types")
    for n in ("STEP_OUTPUT", "EPOCH_OUTPUT", "TRAIN_DATALOADERS", "EVAL_DATALOADERS"):
        setattr(ty, n, object)
    tm = types.ModuleType("torchmetrics")

    class AUROC:
        def __call__(self, s, l):
            return torch.tensor(roc_auc_score(l.numpy(), s.numpy()))

    class ROC:
        def __call__(self, s, l):
            fpr, tpr, thr = roc_curve(l.numpy(), s.numpy())
            return torch.tensor(fpr), torch.tensor(tpr), torch.tensor(thr)

    tm.AUROC, tm.ROC = AUROC, ROC
    for n in ("AveragePrecision", "Recall", "Precision", "StatScores", "Accuracy", "ConfusionMatrix"):
        setattr(tm, n, object)
    if full:
        for n, c in torchmetrics_standins().items():
            setattr(tm, n, c)
    sys.modules.update({"pytorch_lightning": pl, "pytorch_lightning.loggers": lg, "pytorch_lightning.utilities": ut,
                        "pytorch_lightning.utilities.types": ty, "torchmetrics": tm})
    return load_ref_module("ref_controller", "engine/controller.py")


def gen_arcface(L):
    out = {}
    g = torch.Generator().manual_seed(123)
    cases = [("arc_hard", dict(arc_margin=True, easy_margin=False), 16, 100),
             ("arc_easy", dict(arc_margin=True, easy_margin=True), 16, 100),
             ("cosface", dict(arc_margin=False), 16, 100),
             ("arc_hard_400", dict(arc_margin=True, easy_margin=False), 32, 400)]
    for name, kw, B, C in cases:
        for gamma in ((0,) if C > 100 else (0, 2)):
            x = torch.randn(B, 512, generator=g)
            label = torch.randint(0, C, (B,), generator=g)
            torch.manual_seed(1000 + 7 * len(out))   # the head weight is drawn from the GLOBAL generator (xavier_uniform_): pin it
            wrap = L.SoftmaxBasedMetricLearning(torch.nn.Identity(), C, 512, is_focal=True, loss_kwargs=dict(gamma=gamma), **kw)
            w = wrap.add_margin.weight.detach().clone()
            if name == "arc_hard" and gamma == 0:
                # edge rows: target cosine beyond the threshold cos(pi - m), and an almost-aligned pair
                x[0] = -w[label[0]] * 3.0 + 2e-3 * torch.randn(512, generator=g)
                x[1] = w[label[1]] * 2.0 + 1e-3 * torch.randn(512, generator=g)
            xr = x.clone().requires_grad_(True)
            res = wrap(xr, label)
            res["loss"].backward()
            key = f"{name}_g{gamma}"
            out[key + "_x"] = x.numpy(); out[key + "_w"] = w.numpy(); out[key + "_label"] = label.numpy()
            out[key + "_logits"] = res["logits"].detach().numpy(); out[key + "_loss"] = res["loss"].detach().numpy()
            out[key + "_dx"] = xr.grad.numpy(); out[key + "_dw"] = wrap.add_margin.weight.grad.numpy()
            # oracle check
            xo = x.clone().requires_grad_(True); wo = w.clone().requires_grad_(True)
            s, m = wrap.add_margin.s, wrap.add_margin.m
            if kw.get("arc_margin"):
                lo = arcface_ref.arc_margin_logits(xo, wo, label, s, m, kw.get("easy_margin", False))
            else:
                lo = arcface_ref.add_margin_logits(xo, wo, label, s, m)
            loss = arcface_ref.focal_loss(lo, label, gamma)
            loss.backward()
            assert torch.allclose(lo, res["logits"], rtol=1e-6, atol=1e-5), key
            assert abs(loss.item() - res["loss"].item()) < 1e-5, key
            assert torch.allclose(xo.grad, xr.grad, rtol=1e-4, atol=1e-6), key
            assert torch.allclose(wo.grad, wrap.add_margin.weight.grad, rtol=1e-4, atol=1e-6), key
            out[key + "_s"] = np.float32(s); out[key + "_m"] = np.float32(m)
    np.savez_compressed(os.path.join(OUT, "arcface.npz"), **out)
    print("arcface.npz:", len(out), "arrays; oracle == reference")


def gen_arcface_alpha(L):
    """The adaptive-alpha branch of the reference's FocalLoss (losses/losses.py:13-24: `input = self.alpha * input` with a learnable
    per-class alpha) inside SoftmaxBasedMetricLearning — no FE config turns it on, but it is part of the class: loss, logits and the
    gradients of the embedding, the head weight and alpha, for non-trivial alpha values."""
    out = {}
    g = torch.Generator().manual_seed(321)
    for name, kw, gamma in [("arc_hard_alpha", dict(arc_margin=True, easy_margin=False), 2), ("cosface_alpha", dict(arc_margin=False), 0)]:
        B, C = 16, 100
        x = torch.randn(B, 512, generator=g)
        label = torch.randint(0, C, (B,), generator=g)
        torch.manual_seed(4000 + len(out))
        wrap = L.SoftmaxBasedMetricLearning(torch.nn.Identity(), C, 512, is_focal=True, loss_kwargs=dict(gamma=gamma, alpha=True), **kw)
        assert wrap.focal_loss.adaptive_flag
        with torch.no_grad():
            wrap.focal_loss.alpha.copy_(0.5 + torch.rand(C, generator=g))
        w = wrap.add_margin.weight.detach().clone()
        alpha = wrap.focal_loss.alpha.detach().clone()
        xr = x.clone().requires_grad_(True)
        res = wrap(xr, label)
        res["loss"].backward()
        out[name + "_x"] = x.numpy(); out[name + "_w"] = w.numpy(); out[name + "_label"] = label.numpy(); out[name + "_alpha"] = alpha.numpy()
        out[name + "_logits"] = res["logits"].detach().numpy(); out[name + "_loss"] = res["loss"].detach().numpy()
        out[name + "_dx"] = xr.grad.numpy(); out[name + "_dw"] = wrap.add_margin.weight.grad.numpy()
        out[name + "_dalpha"] = wrap.focal_loss.alpha.grad.numpy()
        out[name + "_gamma"] = np.float32(gamma)
        # oracle check: the restated logits, scaled per class, through the restated focal loss
        xo = x.clone().requires_grad_(True); wo = w.clone().requires_grad_(True); ao = alpha.clone().requires_grad_(True)
        s_, m_ = wrap.add_margin.s, wrap.add_margin.m
        lo = arcface_ref.arc_margin_logits(xo, wo, label, s_, m_, False) if kw.get("arc_margin") else arcface_ref.add_margin_logits(xo, wo, label, s_, m_)
        loss = arcface_ref.focal_loss(ao * lo, label, gamma)
        loss.backward()
        assert torch.allclose(lo, res["logits"], rtol=1e-6, atol=1e-5) and abs(loss.item() - res["loss"].item()) < 1e-5, name
        assert torch.allclose(ao.grad, wrap.focal_loss.alpha.grad, rtol=1e-4, atol=1e-6), name
    np.savez_compressed(os.path.join(OUT, "arcface_alpha.npz"), **out)
    print("arcface_alpha.npz:", len(out), "arrays; oracle == reference")


def gen_recall(ctrl_mod, sim_f):
    out = {}
    for name, N, ncls, noise, ties in [("n256", 256, 40, 1.7, False), ("n400", 400, 80, 2.1, False), ("ties", 96, 12, 1.5, True)]:
        g = torch.Generator().manual_seed(7 + N)
        centers = torch.randn(ncls, 512, generator=g)
        classes = torch.randint(0, ncls, (N,), generator=g)
        emb = centers[classes] + noise * torch.randn(N, 512, generator=g) * (512 ** 0.5) / 8
        if ties:
            emb[10] = emb[3]; emb[11] = emb[3]; emb[40] = 2.0 * emb[41]  # exact duplicates / colinear rows
        idx = torch.randperm(N, generator=g)
        pairs = [(int(a), int(b)) for a, b in torch.randint(0, N, (300, 2), generator=g).tolist()]
        plabels = [int(classes[a] == classes[b]) for a, b in pairs]

        class PG:
            corrected_indices = pairs
            labels = plabels

        class Cfg(dict):
            def pair_generator(self, i):
                return "Val", PG

            similarity_f = staticmethod(sim_f)

            def items(self):
                return []

        c = ctrl_mod.Controller.__new__(ctrl_mod.Controller)
        torch.nn.Module.__init__(c)
        c.config = Cfg()
        # outputs: List[dataloader][batch] of dicts, shuffled order, 'index' restores it (controller.py:51-56)
        emb_s, cls_s = emb[idx], classes[idx]
        batches = [{"emb": emb_s[i:i + 20], "label": cls_s[i:i + 20], "index": idx[i:i + 20]} for i in range(0, N, 20)]
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            c.test_epoch_end([batches])
        txt = buf.getvalue()
        vals = {k: float(v) for k, v in re.findall(r"Val ([^\t]+)\t([0-9.eE+-]+)", txt)}
        ours = match_ref.recall_at_k_loop(emb, classes, (10, 100))
        ours_m = match_ref.recall_at_k_matrix(emb, classes, (10, 100))
        for k in (10, 100):
            ref_ratio = vals[f"Recall@K={k}"]
            assert abs(ours[k][0] / ours[k][1] - ref_ratio) < 1e-12 or ties, (name, k, ours[k], ref_ratio)
            assert ours[k] == ours_m[k] or ties, (name, k, ours[k], ours_m[k])
            out[f"{name}_recall{k}_ref"] = np.float64(ref_ratio)
            out[f"{name}_recall{k}_counts"] = np.array(ours[k])
        out[f"{name}_emb"] = emb.numpy().astype(np.float32); out[f"{name}_classes"] = classes.numpy()
        out[f"{name}_auc_ref"] = np.float64(vals["ROC AUC"]); out[f"{name}_acc_ref"] = np.float64(vals["Accuracy"])
        out[f"{name}_pairs"] = np.array(pairs); out[f"{name}_plabels"] = np.array(plabels)
        sc = sim_f([(emb[a], emb[b]) for a, b in pairs])
        assert torch.allclose(sc, arcface_ref.similarity_f(emb[[a for a, _ in pairs]], emb[[b for _, b in pairs]]), atol=1e-7)
        out[f"{name}_pair_scores"] = sc.numpy()
        print(f"recall {name}: ref R@10={vals['Recall@K=10']:.4f} R@100={vals['Recall@K=100']:.4f} counts={ours}")
    np.savez_compressed(os.path.join(OUT, "recall.npz"), **out)


def gen_evaluate(sim_f):
    """The reference's own Controller._evaluate (engine/controller.py:95-203) run on the three seeded sets of recall.npz (same
    embeddings, classes and pairs) with config.k = [5, 10, 100], two thrs, far_thr / frr_thr lists incl. values whose index
    rule lands on int(...) == 0 — the printed metrics dict is the golden, key for key."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    import tempfile
    from sklearn.metrics import roc_auc_score, average_precision_score
    plt.pause = lambda *_a, **_k: None
    ctrl_mod = ref_controller(full=True)
    rec = np.load(os.path.join(OUT, "recall.npz"))
    out = {}
    thrs = np.array([0.5, 0.598])          # config.thrs is a numpy array in the reference's configs (fe_dogs_config.py:71)
    far = [0.1, 0.05, 0.01, 0.001]
    frr = [0.1, 0.03, 0.001]
    ks = [5, 10, 100]
    # "edge": impostor pairs of identical rows (score exactly 1) and genuine pairs of opposite rows (score exactly 0), so that
    # the far / frr index rule lands on thr in (0, 1) and the reference SKIPS those keys (controller.py:170,177)
    g = torch.Generator().manual_seed(77)
    e_emb = torch.randn(64, 512, generator=g)
    e_cls = torch.arange(64) // 8
    e_emb[8] = e_emb[0]; e_emb[16] = e_emb[0]; e_emb[24] = e_emb[0]          # different identities, same vector
    e_emb[2] = -e_emb[1]; e_emb[4] = -e_emb[3]                                # same identity, opposite vectors
    e_pairs = [(0, 8), (0, 16), (8, 16), (0, 24), (8, 24), (16, 24), (1, 2), (3, 4), (2, 1)] + \
              [(int(a), int(b)) for a, b in torch.randint(0, 64, (111, 2), generator=g).tolist()]
    e_lab = [int(e_cls[a] == e_cls[b]) for a, b in e_pairs]
    out["edge_emb"] = e_emb.numpy(); out["edge_classes"] = e_cls.numpy()
    out["edge_pairs"] = np.array(e_pairs); out["edge_plabels"] = np.array(e_lab)
    for name in ("n256", "n400", "ties", "edge"):
        src = out if name == "edge" else rec
        emb = torch.from_numpy(src[f"{name}_emb"])
        classes = torch.from_numpy(src[f"{name}_classes"])
        pairs = [(int(a), int(b)) for a, b in src[f"{name}_pairs"]]
        plabels = [int(v) for v in src[f"{name}_plabels"]]
        N = emb.shape[0]
        idx = torch.randperm(N, generator=torch.Generator().manual_seed(5 + N))

        class PG:
            corrected_indices = pairs
            labels = plabels

        with tempfile.TemporaryDirectory() as td:
            class Cfg(dict):
                def pair_generator(self, i):
                    return "Val", PG

                similarity_f = staticmethod(sim_f)

            cfg = Cfg(far_thr=far, frr_thr=frr, img_dir=td)
            cfg.thrs, cfg.k = thrs, ks
            c = ctrl_mod.Controller.__new__(ctrl_mod.Controller)
            torch.nn.Module.__init__(c)
            c.config = cfg
            c.current_epoch = 0
            c.logger = None
            emb_s, cls_s = emb[idx], classes[idx]
            batches = [{"emb": emb_s[i:i + 20], "label": cls_s[i:i + 20], "index": idx[i:i + 20]} for i in range(0, N, 20)]
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                c._evaluate([batches])
        vals = {k: float(v) for k, v in re.findall(r"^Val ([^\t\n]+)\t([0-9.eE+-]+|nan|inf)$", buf.getvalue(), flags=re.M)}
        cm = re.search(r"Conf Mat thr = [^ ]+ tensor\(\[\[ *(\d+), *(\d+)\],\s*\[ *(\d+), *(\d+)\]\]\)", buf.getvalue())
        sc = sim_f([(emb[a], emb[b]) for a, b in pairs])
        lab = np.array(plabels)
        # the stand-ins against scikit-learn (an independent implementation of the same definitions)
        assert abs(vals["ROC AUC"] - roc_auc_score(lab, sc.numpy())) < 1e-6, name
        assert abs(vals["AveragePrecision"] - average_precision_score(lab, sc.numpy())) < 1e-6, name
        keys = list(vals.keys())
        out[f"{name}_keys"] = np.array(keys)
        out[f"{name}_values"] = np.array([vals[k] for k in keys], dtype=np.float64)
        out[f"{name}_confmat"] = np.array([int(v) for v in cm.groups()]).reshape(2, 2)
        out[f"{name}_pair_scores"] = sc.numpy()
        print(f"evaluate {name}:", {k: round(v, 5) for k, v in vals.items()})
    out["thrs"] = thrs; out["far_thr"] = np.array(far); out["frr_thr"] = np.array(frr); out["k"] = np.array(ks)
    np.savez_compressed(os.path.join(OUT, "evaluate.npz"), **out)


def gen_pairs():
    """The reference's PairGenerator (data_loading/pairs.py:10-108) run on seeded identity tables: pairs + correction table.
    `data_loading/dataset.py` needs PIL / torchvision-free imports only for file scanning; a stub module stands in for it."""
    ds_stub = types.ModuleType("ref_data_loading.dataset")
    ds_stub.RecDataset = object
    pkg = types.ModuleType("ref_data_loading")
    pkg.__path__ = [os.path.join(REF, "data_loading")]
    sys.modules["ref_data_loading"] = pkg
    sys.modules["ref_data_loading.dataset"] = ds_stub
    ref = load_ref_module("ref_data_loading.pairs", "data_loading/pairs.py")
    out = {}
    for name, n_id, seed, gen_number, ratio in [("a", 30, 3, 60, 1), ("b", 57, 11, None, 0.5), ("c", 120, 42, 400, 2)]:
        rs = np.random.RandomState(100 + seed)
        counts = rs.randint(1, 6, size=n_id)
        perm = rs.permutation(int(counts.sum()))          # dataset indices of an identity are not contiguous in general
        u2i, o = {}, 0
        for u in range(n_id):
            u2i[u] = sorted(int(v) for v in perm[o:o + counts[u]])
            o += counts[u]
        users = [u for u in range(n_id) if rs.rand() < 0.6]

        class DS:
            uid_to_indices = u2i

            def __len__(self):
                return int(counts.sum())

        pg = ref.PairGenerator(DS(), gen_number, ratio, None, seed, users)
        out[f"{name}_counts"] = counts; out[f"{name}_perm"] = perm; out[f"{name}_users"] = np.array(users)
        out[f"{name}_args"] = np.array([-1 if gen_number is None else gen_number, seed], dtype=np.int64)
        out[f"{name}_ratio"] = np.float64(ratio)
        out[f"{name}_pairs"] = np.array(pg.pairs, dtype=np.int64)
        ck = sorted(pg.correction)
        out[f"{name}_corr_keys"] = np.array(ck, dtype=np.int64)
        out[f"{name}_corr_vals"] = np.array([pg.correction[k] for k in ck], dtype=np.int64)
        out[f"{name}_corrected"] = np.array(pg.corrected_indices, dtype=np.int64)
        out[f"{name}_labels"] = np.asarray(pg.labels)
        print(f"pairs {name}: {len(pg.pairs)} pairs ({int(np.sum(pg.labels))} genuine), {len(ck)} indices")
    np.savez_compressed(os.path.join(OUT, "pairs.npz"), **out)


def ref_calc_scores():
    """generate_tsv.py's `similarity_f`, `mean_strategy_cal_scores`, `calc_scores` (:63-125), extracted by name — the
    module itself imports the whole inference stack (mmdet pipelines, pandas CLI) at import time."""
    import typing
    from pathlib import Path
    src = open(os.path.join(REF, "generate_tsv.py")).read()
    tree = ast.parse(src)
    want = ("similarity_f", "mean_strategy_cal_scores", "calc_scores")
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert len(fns) == len(want)
    ns = {"torch": torch, "F": torch.nn.functional, "np": np, "Path": Path, "List": typing.List, "Dict": typing.Dict,
          "Any": typing.Any, "tqdm": lambda it, **kw: it}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "generate_tsv.calc_scores", "exec"), ns)
    return ns["calc_scores"]


def gen_calc_scores():
    from pathlib import Path
    from oracle.match_ref import calc_scores_case
    calc = ref_calc_scores()
    (qh, qhs, qb, qbs, qt), (gh, ghs, gb, gbs, gt) = calc_scores_case()

    def db(prefix, h, hs, b, bs, t):
        return {Path(f"/cards/{prefix}{c:04d}"): {"head_vectors": [torch.from_numpy(v) for v in h[hs[c]:hs[c + 1]]],
                                                  "body_vectors": [torch.from_numpy(v) for v in b[bs[c]:bs[c + 1]]],
                                                  "type": int(t[c])} for c in range(len(t))}
    rows = calc(db("q", qh, qhs, qb, qbs, qt), db("g", gh, ghs, gb, gbs, gt))
    qrow = np.array([int(r[0][1:]) for r in rows])
    ans = np.full((len(rows), 100), -1, np.int64)
    for i, r in enumerate(rows):
        a = [int(n[1:]) for n in r[4].split(",")]
        ans[i, :len(a)] = a
    np.savez_compressed(os.path.join(OUT, "calc_scores.npz"), q_head=qh, q_head_seg=qhs, q_body=qb, q_body_seg=qbs, q_type=qt,
                        g_head=gh, g_head_seg=ghs, g_body=gb, g_body_seg=gbs, g_type=gt, rows_query=qrow,
                        top1=np.array([r[1] for r in rows]), mean3=np.array([r[2] for r in rows]),
                        mean10=np.array([r[3] for r in rows]), answer=ans)
    print("calc_scores.npz:", len(rows), "rows of", len(qt), "queries; answers/row",
          sorted(set((ans >= 0).sum(1).tolist()))[:4], "...")


def gen_augment():
    """tests/golden/augment.npz: the train augmentation of fe_dogs_config.py:17-26 executed with PILLOW's own functions
    (the calls torchvision's PIL backend makes; torchvision itself is not installed here) for fixed random decisions."""
    from PIL import Image, ImageEnhance, ImageOps
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_augment_oracle import _images
    out = {}
    for tag, n, H, W, crop, size in (("full", 4, 224, 224, 220, 224), ("small", 10, 48, 52, 44, 48)):
        x = np.stack(_images(17, n, H, W))
        g = torch.Generator().manual_seed(100 + n)
        flags = np.stack([(torch.rand(n, generator=g) < 0.5).int().numpy(), (torch.rand(n, generator=g) < 0.5).int().numpy(),
                          torch.randint(0, H - crop + 1, (n,), generator=g).int().numpy(),
                          torch.randint(0, W - crop + 1, (n,), generator=g).int().numpy()], 1).astype(np.int32)
        flags[0, :2] = (1, 1)
        flags[1, :2] = (0, 0)
        flags[2, :2] = (1, 0)
        flags[3, :2] = (0, 1)
        angles = torch.empty(n).uniform_(-5, 5, generator=g).numpy()
        res = []
        for i in range(n):
            im = Image.fromarray(x[i])
            if flags[i, 0]:
                im = ImageEnhance.Sharpness(im).enhance(0)
            if flags[i, 1]:
                im = ImageOps.autocontrast(im)
            t, l = int(flags[i, 2]), int(flags[i, 3])
            im = im.crop((l, t, l + crop, t + crop))
            im = im.resize((size, size), Image.BILINEAR)
            im = im.rotate(float(angles[i]), Image.NEAREST, expand=False, center=None, fillcolor=(0, 0, 0))
            res.append(np.asarray(im))
        out.update({f"{tag}_x": x, f"{tag}_flags": flags, f"{tag}_angles": angles, f"{tag}_out": np.stack(res),
                    f"{tag}_crop": crop, f"{tag}_size": size})
    np.savez_compressed(os.path.join(OUT, "augment.npz"), **out)
    print("augment.npz:", {k: getattr(v, "shape", v) for k, v in out.items()})


def gen_swin():
    ref = load_ref_module("ref_swin", "models/swin.py")
    torch.manual_seed(1234)
    m = ref.swin_t(num_classes=512)
    g = torch.Generator().manual_seed(99)
    x = torch.rand(2, 3, 224, 224, generator=g)
    with torch.no_grad():
        s1 = m.stage1(x)
        y = m(x)
    np.savez_compressed(os.path.join(OUT, "swin_t.npz"), seed=1234, x_seed=99, emb=y.numpy(),
                        stage1_sample=s1[:, :8, :6, :6].numpy(), n_params=sum(p.numel() for p in m.parameters()),
                        keys=np.array(sorted(m.state_dict().keys())))
    print("swin_t.npz: emb", tuple(y.shape))


def gen_train_trace(L):
    """5 SGD steps of ResNet-18 (oracle restatement; torchvision is absent) inside the REFERENCE's
    SoftmaxBasedMetricLearning (ArcFace s=64 m=0.5 + FocalLoss γ=0), optimiser groups as fe_dogs_config.py:123-133."""
    arch, C, B, HW = "resnet18", 100, 8, 96
    sd = resnet_ref.init_state_dict(arch, 512, seed=5)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.ParameterDict({k.replace(".", "__"): torch.nn.Parameter(v.clone()) for k, v in sd.items()
                                             if v.dtype.is_floating_point and "running" not in k})
            self.stats = {k: v.clone() for k, v in sd.items() if "running" in k}

        def forward(self, x):
            cur = {k.replace("__", "."): v for k, v in self.p.items()}
            cur.update(self.stats)
            new = {}
            y = resnet_ref.forward(cur, x, arch, train=True, new_stats=new)
            self.stats.update(new)
            return y

    torch.manual_seed(11)
    net = Net()
    wrap = L.SoftmaxBasedMetricLearning(net, C, 512, is_focal=True, arc_margin=True)
    w0 = wrap.add_margin.weight.detach().clone()
    fc = [p for n, p in net.p.items() if n.startswith("fc")]
    rest = [p for n, p in net.p.items() if not n.startswith("fc")]
    opt = torch.optim.SGD([{"lr": 5e-3, "params": rest}, {"lr": 1e-2, "params": fc},
                           {"lr": 1e-2, "params": wrap.add_margin.parameters(), "weight_decay": 1e-4}], 0.01, momentum=0.9)
    g = torch.Generator().manual_seed(321)
    x8 = torch.randint(0, 256, (5, B, 3, HW, HW), generator=g, dtype=torch.uint8)  # ToTensor(): uint8 / 255
    xs = x8.float() / 255.0
    ys = torch.randint(0, C, (5, B), generator=g)
    losses, emb0 = [], None
    for i in range(5):
        opt.zero_grad()
        r = wrap(xs[i], ys[i])
        if i == 0:
            emb0 = r["emb"].detach().clone()
        r["loss"].backward()
        opt.step()
        losses.append(r["loss"].item())
    np.savez_compressed(os.path.join(OUT, "train_trace_r18.npz"), arch=arch, C=C, B=B, HW=HW, init_seed=5, x_u8=x8.numpy(),
                        y=ys.numpy(), head_w0=w0.numpy(), losses=np.array(losses), emb0=emb0.numpy(),
                        rm_bn1=net.stats["bn1.running_mean"].numpy(), rv_bn1=net.stats["bn1.running_var"].numpy(),
                        fc_bias_final=net.p["fc__bias"].detach().numpy())
    print("train_trace_r18.npz: losses", [round(v, 5) for v in losses])


def hf_resnet(arch, sd):
    """the state dict `sd` (torchvision names) loaded into Hugging Face transformers' ResNetModel"""
    from transformers import ResNetConfig, ResNetModel
    if arch == "resnet50":
        cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3],
                           layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False,
                           downsample_in_bottleneck=False)
        nconv = 3
    else:
        cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[2, 2, 2, 2],
                           layer_type="basic", hidden_act="relu", downsample_in_first_stage=False)
        nconv = 2
    hf = ResNetModel(cfg)
    new = {}

    def put(dst, src_conv, src_bn):
        new[dst + ".convolution.weight"] = sd[src_conv + ".weight"]
        for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            new[dst + ".normalization." + k] = sd[src_bn + "." + k]

    put("embedder.embedder", "conv1", "bn1")
    for Lx, depth in enumerate(cfg.depths):
        for b in range(depth):
            tv, h = f"layer{Lx + 1}.{b}", f"encoder.stages.{Lx}.layers.{b}"
            for i in range(nconv):
                put(f"{h}.layer.{i}", f"{tv}.conv{i + 1}", f"{tv}.bn{i + 1}")
            if f"{tv}.downsample.0.weight" in sd:
                put(f"{h}.shortcut", f"{tv}.downsample.0", f"{tv}.downsample.1")
    hf.load_state_dict(new, strict=True)
    return hf


def resnet_hf_inputs(arch):
    """seeded weights (non-trivial running statistics) + input of the ResNet cross-check fixture"""
    sd = resnet_ref.init_state_dict(arch, 512, seed=4)
    g = torch.Generator().manual_seed(9)
    for k in list(sd):
        if k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
    x = torch.rand(3, 3, 96, 96, generator=g)
    return sd, x


def gen_resnet_hf():
    """torchvision (the reference's backbone provider) cannot be installed here; Hugging Face transformers ships an
    independent implementation of the same published architecture (ResNet v1.5).  Its outputs for seeded weights — which the
    restatement regenerates from the seed — pin the backbone oracle: eval mode, train mode, updated running statistics."""
    out = {}
    for arch in ("resnet18", "resnet50"):
        sd, x = resnet_hf_inputs(arch)
        hf = hf_resnet(arch, sd)
        for train in (False, True):
            hf.train(train)
            with torch.no_grad():
                pooled = hf(x).pooler_output.flatten(1)
                emb = pooled @ sd["fc.weight"].t() + sd["fc.bias"]
            ref = resnet_ref.forward(sd, x, arch, train=train)
            assert torch.allclose(ref, emb, rtol=2e-4, atol=2e-5), (arch, train)
            out[f"{arch}_{'train' if train else 'eval'}_emb"] = emb.numpy()
            out[f"{arch}_{'train' if train else 'eval'}_pooled"] = pooled.numpy()
        out[f"{arch}_bn1_running_mean_after"] = hf.state_dict()["embedder.embedder.normalization.running_mean"].numpy()
    np.savez_compressed(os.path.join(OUT, "resnet_hf.npz"), **out)
    print("resnet_hf.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "resnet_hf":
        gen_resnet_hf()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pairs":
        gen_pairs()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "augment":
        gen_augment()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "calc_scores":
        gen_calc_scores()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "evaluate":
        gen_evaluate(ref_similarity_f())
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "arcface":
        gen_arcface(ref_losses())
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "arcface_alpha":
        gen_arcface_alpha(ref_losses())
        sys.exit(0)
    L = ref_losses()
    gen_arcface(L)
    gen_arcface_alpha(L)
    gen_recall(ref_controller(), ref_similarity_f())
    gen_evaluate(ref_similarity_f())
    gen_pairs()
    gen_calc_scores()
    gen_augment()
    gen_swin()
    gen_train_trace(L)
    gen_resnet_hf()
