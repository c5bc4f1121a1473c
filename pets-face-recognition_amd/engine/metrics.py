"""Verification metrics of the evaluation epoch.  Mirrors what /root/reference/engine/controller.py:67-75,114-183 gets from
torchmetrics: ROC (fpr, tpr, thresholds), AUROC, accuracy at the threshold minimising fpr+fnr, AP, confusion matrix, P/R@thr,
TAR@FAR, TRR@FRR (no torchmetrics / sklearn dependency).

The sort + running-count part runs on the device when the scores are CUDA tensors (`pfr_pair_curve`: one
workgroup bitonic-sorts the ≤ 20 000 pair scores with their labels and scans the genuine-pair count); what is left is a handful
of operations on the distinct-threshold operating points.  CPU tensors take the torch path below (BASELINE config 1)."""
import torch


def _curve(scores, labels):
    """→ (score, genuine-pair count, 0-based position) at the end of every run of equal scores in descending order, n_pos, n_neg;
    CPU double / long tensors"""
    if scores.is_cuda:
        from .._hip import lib
        sc = scores.detach().float().contiguous().flatten()
        lb = labels.detach().to(sc.device).to(torch.int32).contiguous().flatten()
        P = sc.numel()
        st = torch.cuda.current_stream().cuda_stream
        ws = torch.empty(lib.pfr_pair_curve_ws_bytes(P), dtype=torch.uint8, device=sc.device)
        ss = torch.empty(P, dtype=torch.float32, device=sc.device)
        ct = torch.empty(P, dtype=torch.int32, device=sc.device)
        re = torch.empty(P, dtype=torch.uint8, device=sc.device)
        lib.pfr_pair_curve(sc.data_ptr(), lb.data_ptr(), P, ws.data_ptr(), ss.data_ptr(), ct.data_ptr(), re.data_ptr(), st)
        ends = torch.nonzero(re).flatten()
        npos = float(ct[-1].item())
        return ss[ends].double().cpu(), ct[ends].double().cpu(), ends.cpu(), npos, float(P) - npos
    scores = scores.detach().double().cpu().flatten()
    labels = labels.detach().cpu().flatten().long()
    order = torch.argsort(scores, descending=True, stable=True)
    s, y = scores[order], labels[order]
    distinct = torch.nonzero(s[1:] != s[:-1]).flatten()
    ends = torch.cat([distinct, torch.tensor([s.numel() - 1])])
    return s[ends], torch.cumsum(y, 0)[ends].double(), ends, float(y.sum()), float((1 - y).sum())


def roc_curve(scores, labels):
    """→ fpr, tpr, thresholds (descending; first point (0,0) at threshold max+1 like torchmetrics.ROC)"""
    s, tps, ends, P, N = _curve(scores, labels)
    fps = (ends + 1).double() - tps
    tpr = torch.cat([torch.zeros(1, dtype=torch.double), tps / max(P, 1.0)])
    fpr = torch.cat([torch.zeros(1, dtype=torch.double), fps / max(N, 1.0)])
    thr = torch.cat([s[:1] + 1, s])
    return fpr, tpr, thr


def auroc(scores, labels):
    fpr, tpr, _ = roc_curve(scores, labels)
    return float(torch.trapz(tpr, fpr))


def average_precision(scores, labels):
    """AP = Σ (R_n − R_{n−1})·P_n over the distinct score thresholds (ties share one operating point)"""
    _, tp, ends, P, _ = _curve(scores, labels)
    prec = tp / (ends + 1).double()
    rec = tp / max(P, 1.0)
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
