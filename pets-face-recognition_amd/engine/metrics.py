"""Verification metrics of the evaluation epoch (host logic over ≤ 20 000 pair scores; torch CPU, no torchmetrics /
sklearn dependency).  Mirrors what /root/reference/engine/controller.py:67-75,114-183 gets from torchmetrics:
ROC (fpr, tpr, thresholds), AUROC, accuracy at the threshold minimising fpr+fnr, AP, confusion matrix, P/R@thr,
TAR@FAR, TRR@FRR."""
import torch


def roc_curve(scores, labels):
    """→ fpr, tpr, thresholds (descending; first point (0,0) at threshold max+1 like torchmetrics.ROC)"""
    scores = scores.detach().double().cpu().flatten()
    labels = labels.detach().cpu().flatten().long()
    order = torch.argsort(scores, descending=True, stable=True)
    s, y = scores[order], labels[order]
    distinct = torch.nonzero(s[1:] != s[:-1]).flatten()
    ends = torch.cat([distinct, torch.tensor([s.numel() - 1])])
    tps = torch.cumsum(y, 0)[ends].double()
    fps = (ends + 1).double() - tps
    P, N = float(y.sum()), float((1 - y).sum())
    tpr = torch.cat([torch.zeros(1, dtype=torch.double), tps / max(P, 1.0)])
    fpr = torch.cat([torch.zeros(1, dtype=torch.double), fps / max(N, 1.0)])
    thr = torch.cat([s[ends[:1]] + 1, s[ends]])
    return fpr, tpr, thr


def auroc(scores, labels):
    fpr, tpr, _ = roc_curve(scores, labels)
    return float(torch.trapz(tpr, fpr))


def average_precision(scores, labels):
    """AP = Σ (R_n − R_{n−1})·P_n over the distinct score thresholds (ties share one operating point)"""
    scores = scores.detach().double().cpu().flatten()
    labels = labels.detach().cpu().flatten().long()
    order = torch.argsort(scores, descending=True, stable=True)
    s, y = scores[order], labels[order].double()
    distinct = torch.nonzero(s[1:] != s[:-1]).flatten()
    ends = torch.cat([distinct, torch.tensor([s.numel() - 1])])
    tp = torch.cumsum(y, 0)[ends]
    prec = tp / (ends + 1).double()
    rec = tp / max(float(y.sum()), 1.0)
    prev = torch.cat([torch.zeros(1, dtype=torch.double), rec[:-1]])
    return float(((rec - prev) * prec).sum())


def best_threshold_accuracy(scores, labels, thresholds, fpr, fnr):
    """accuracy at t = thresholds[argmin(fpr+fnr)] (reference: Controller.compute_accuracy, controller.py:206-211)"""
    t = thresholds[torch.argmin(fpr + fnr)]
    scores = scores.detach().double().cpu().flatten()
    labels = labels.detach().cpu().flatten().long()
    gen, imp = scores[labels == 1], scores[labels == 0]
    return float(((gen > t).sum() + (imp <= t).sum()).item() / labels.numel()), float(t)


def stats_at_threshold(scores, labels, thr):
    scores = scores.detach().double().cpu().flatten()
    labels = labels.detach().cpu().flatten().long()
    pred = (scores >= thr).long()
    tp = int(((pred == 1) & (labels == 1)).sum())
    fp = int(((pred == 1) & (labels == 0)).sum())
    tn = int(((pred == 0) & (labels == 0)).sum())
    fn = int(((pred == 0) & (labels == 1)).sum())
    acc = (tp + tn) / max(1, labels.numel())
    prec = tp / max(1, tp + fp)
    rec = tp / max(1, tp + fn)
    return dict(tp=tp, fp=fp, tn=tn, fn=fn, accuracy=acc, precision=prec, recall=rec)


def tar_at_far(fpr, tpr, far):
    """largest TPR whose FPR does not exceed `far`"""
    ok = fpr <= far
    return float(tpr[ok].max()) if ok.any() else 0.0
