"""Verification metrics of the evaluation epoch: what /root/reference/engine/controller.py:67-75,112-183 gets from torchmetrics
(ROC, AUROC, AveragePrecision, ConfusionMatrix, Accuracy / Precision / Recall(threshold=), StatScores(threshold=)) and the
index rules the reference applies itself (`Opt thr`, TAR@FAR / TH@FAR, TRR@FRR / TH@FRR), with no torchmetrics / sklearn
dependency.  Pinned to the reference's own `_evaluate` output: tests/golden/evaluate.npz.

Arithmetic follows the reference where it decides a discrete outcome: binary predictions are `score >= threshold` in the
scores' float32; fpr / tpr are integer counts divided in float32 (torchmetrics' `fps / fps[-1]`), so `argmin(fpr + 1 - tpr)`
(`Opt thr`, controller.py:120) and `argmin(fpr + fnr)` (`compute_accuracy`, 206-211) see the same ties.  The areas (AUROC, AP)
are accumulated in float64 — the reference prints their float32 values (tolerance 1e-6 in the tests).

The sort + running-count part runs on the device when the scores are CUDA tensors (`pfr_pair_curve`: one
workgroup bitonic-sorts the ≤ 20 000 pair scores with their labels and scans the genuine-pair count); every metric below is
a handful of operations on that one sorted list.  CPU tensors take the torch sort (BASELINE config 1)."""
import torch


class PairStats:
    """The pair scores of one validation set, sorted once (descending): `s` float32 scores, `y` 0/1 labels, both on the host."""

    def __init__(self, scores, labels):
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
            self.s = ss.cpu()
            self.tp_cum = ct.cpu().long()                       # genuine pairs among the first i+1 scores
            self.ends = torch.nonzero(re.cpu()).flatten()       # last position of every run of equal scores
            self.y = torch.diff(self.tp_cum, prepend=torch.zeros(1, dtype=torch.long))
        else:
            s = scores.detach().float().cpu().flatten()
            y = labels.detach().cpu().flatten().long()
            order = torch.argsort(s, descending=True, stable=True)
            self.s, self.y = s[order], y[order]
            self.tp_cum = torch.cumsum(self.y, 0)
            distinct = torch.nonzero(self.s[1:] != self.s[:-1]).flatten()
            self.ends = torch.cat([distinct, torch.tensor([self.s.numel() - 1])])
        self.n = self.s.numel()
        self.n_pos = int(self.tp_cum[-1]) if self.n else 0
        self.n_neg = self.n - self.n_pos

    # ---- curves: one operating point per distinct score (torchmetrics' _binary_clf_curve)
    def _counts(self):
        tps = self.tp_cum[self.ends]
        fps = 1 + self.ends - tps
        return tps, fps

    def roc(self):
        """→ fpr, tpr, thresholds (float32, descending thresholds, first point (0, 0) at max + 1: torchmetrics.ROC)"""
        tps, fps = self._counts()
        z = torch.zeros(1, dtype=torch.long)
        tps, fps = torch.cat([z, tps]), torch.cat([z, fps])
        thr = torch.cat([self.s[self.ends][:1] + 1, self.s[self.ends]])
        # (a set without impostor / without genuine pairs: rates 0 instead of torchmetrics' 0/0 = nan)
        return fps / fps[-1].clamp_min(1), tps / tps[-1].clamp_min(1), thr

    def auroc(self):
        tps, fps = self._counts()
        z = torch.zeros(1, dtype=torch.double)
        tpr = torch.cat([z, tps.double() / max(self.n_pos, 1)])
        fpr = torch.cat([z, fps.double() / max(self.n_neg, 1)])
        return float(torch.trapz(tpr, fpr))

    def average_precision(self):
        """AP = Σ (R_n − R_{n−1})·P_n over the distinct score thresholds (ties share one operating point)"""
        tps, fps = self._counts()
        prec = tps.double() / (tps + fps).double()
        rec = tps.double() / max(self.n_pos, 1)
        prev = torch.cat([torch.zeros(1, dtype=torch.double), rec[:-1]])
        return float(((rec - prev) * prec).sum())

    def opt_threshold(self):
        """`thresholds[argmin(fpr + 1 - tpr)]` (controller.py:120), float32 arithmetic in the reference's operand order"""
        fpr, tpr, thr = self.roc()
        return float(thr[torch.argmin(fpr + 1 - tpr)])

    # ---- counts at a threshold: prediction = score >= thr in float32 (torchmetrics' binary input handling)
    def stats_at(self, thr):
        t = torch.as_tensor(thr).to(torch.float32)
        npred = int((self.s >= t).sum())            # s is descending: the predicted-positive scores are a prefix
        tp = int(self.tp_cum[npred - 1]) if npred else 0
        fp = npred - tp
        fn = self.n_pos - tp
        tn = self.n_neg - fp
        return dict(tp=tp, fp=fp, tn=tn, fn=fn,
                    accuracy=(tp + tn) / max(1, self.n),
                    precision=tp / (tp + fp) if tp + fp else 0.0,     # torchmetrics: zero_division = 0
                    recall=tp / (tp + fn) if tp + fn else 0.0)

    def best_threshold_accuracy(self, thresholds=None, fpr=None, fnr=None):
        """accuracy at t = thresholds[argmin(fpr + fnr)] with `score > t` as the decision (Controller.compute_accuracy,
        controller.py:206-211) → (accuracy, t)"""
        if thresholds is None:
            fpr, tpr, thresholds = self.roc()
            fnr = 1 - tpr
        t = thresholds[torch.argmin(fpr + fnr)].to(torch.float32)
        gt = self.s > t
        n_true = int((gt & (self.y == 1)).sum()) + int((~gt & (self.y == 0)).sum())
        return n_true / max(1, self.n), float(t)

    # ---- the reference's own index rules (controller.py:162-181)
    def far_points(self, far_thrs):
        """for every far: thr = neg_scores[-int(len(neg) * far)] on the ascending impostor scores (index −0 is index 0, as in the
        reference); skipped when thr is exactly 0 or 1; TAR = #(genuine >= thr) / #genuine → {far: (tar, thr)}"""
        neg = self.s[self.y == 0].flip(0)
        out = {}
        for far in far_thrs:
            thr = neg[-int(len(neg) * far)]
            if float(thr) not in (0.0, 1.0):
                out[far] = (self.stats_at(thr)['tp'] / self.n_pos, float(thr))
        return out

    def frr_points(self, frr_thrs):
        """thr = pos_scores[int(len(pos) * frr)] on the ascending genuine scores; TRR = #(impostor < thr) / #impostor"""
        pos = self.s[self.y == 1].flip(0)
        out = {}
        for frr in frr_thrs:
            thr = pos[int(len(pos) * frr)]
            if float(thr) not in (0.0, 1.0):
                out[frr] = (self.stats_at(thr)['tn'] / self.n_neg, float(thr))
        return out


# ---- function forms (one sort per call)
def roc_curve(scores, labels):
    return PairStats(scores, labels).roc()


def auroc(scores, labels):
    return PairStats(scores, labels).auroc()


def average_precision(scores, labels):
    return PairStats(scores, labels).average_precision()


def best_threshold_accuracy(scores, labels, thresholds, fpr, fnr):
    return PairStats(scores, labels).best_threshold_accuracy(thresholds, fpr, fnr)


def stats_at_threshold(scores, labels, thr):
    return PairStats(scores, labels).stats_at(thr)
