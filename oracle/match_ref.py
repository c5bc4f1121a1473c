"""CPU restatement of the reference's embedding match / candR@K (Recall@K) evaluation.

  recall_at_k_loop        ← /root/reference/engine/controller.py:77-90 (and 143-160): per query, score every OTHER
                             embedding with similarity_f, sort descending, hit if any same-class item is in the top K;
                             denominator = queries that have at least one same-class other.
  recall_at_k_matrix      — the same quantity from one score matrix (used at sizes the loop cannot reach)
  topk_query_gallery      — query-vs-gallery top-K (BASELINE config 5; the shape of generate_tsv.py:91-125)
Tie rule (the reference's argsort is unstable, so its own result is undefined on ties at the K boundary): score
descending, then LOWER index first."""
import numpy as np
import torch

from .arcface_ref import similarity_f


def recall_at_k_loop(emb, classes, ks=(10, 100)):
    """Literal restatement: python loop over queries, exclusion of self by index, stable tie-break by index."""
    n = emb.shape[0]
    counts = {k: [0, 0] for k in ks}
    for j in range(n):
        others = [i for i in range(n) if i != j]
        sc = similarity_f(emb[j].unsqueeze(0).expand(n - 1, -1), emb[others])
        oc = classes[others]
        order = torch.argsort(sc, descending=True, stable=True)
        oc = oc[order]
        has = int((oc == classes[j]).any().item())
        for k in ks:
            counts[k][0] += int((oc[:k] == classes[j]).any().item())
            counts[k][1] += has
    return counts


def recall_at_k_matrix(emb, classes, ks=(10, 100)):
    e = emb / emb.norm(dim=1, keepdim=True).clamp_min(1e-8)
    sc = e @ e.t()
    n = emb.shape[0]
    sc.fill_diagonal_(-float("inf"))
    kmax = min(max(ks), n - 1)
    # stable descending order with lower index first on ties
    order = torch.argsort(sc, dim=1, descending=True, stable=True)[:, :kmax]
    same = classes[order] == classes[:, None]
    eq = classes[:, None] == classes[None, :]
    has = eq.sum(dim=1) > 1
    counts = {}
    for k in ks:
        counts[k] = [int((same[:, :k].any(dim=1) & has).sum().item()), int(has.sum().item())]
    return counts


def topk_query_gallery(q, g, k):
    """→ (scores [Q,k] fp32 cosine, indices [Q,k] int64), descending, ties → lower gallery index"""
    qn = q / q.norm(dim=1, keepdim=True).clamp_min(1e-8)
    gn = g / g.norm(dim=1, keepdim=True).clamp_min(1e-8)
    sc = qn @ gn.t()
    order = torch.argsort(sc, dim=1, descending=True, stable=True)[:, :k]
    return torch.gather(sc, 1, order), order


def cand_recall_query_gallery(idx, q_cls, g_cls, ks=(10, 100)):
    hit = g_cls[idx] == q_cls[:, None]
    present = (q_cls[:, None] == torch.unique(g_cls)[None, :]).any(dim=1)
    return {k: [int((hit[:, :k].any(dim=1) & present).sum().item()), int(present.sum().item())] for k in ks}


def recall_loop_reference_cost(emb, classes, queries, ks=(10, 100)):
    """COST-faithful restatement of the per-query body of /root/reference/engine/controller.py:77-90 + similarity_f
    (configs/dog_fe/fe_dogs_config.py:89-93) for the queries in `queries` only: tuple index list, a python list of per-row
    tensors, `torch.cat` of unsqueezed rows, cosine similarity, (unstable) argsort — the operations whose interpreter overhead
    makes the reference's evaluation O(N²) at ~8 µs per scored pair.  Used as the timed CPU baseline of the match (bench.py)."""
    import torch.nn.functional as F

    def sim(pairs):
        t1 = torch.cat([p[0].unsqueeze(0) for p in pairs], dim=0)
        t2 = torch.cat([p[1].unsqueeze(0) for p in pairs], dim=0)
        return (F.cosine_similarity(t1, t2) + 1) / 2

    n = classes.shape[0]
    out = {k: [0, 0] for k in ks}
    for j in queries:
        cur, cls = emb[j], classes[j]
        other = emb[tuple(jj for jj in range(n) if jj != j), :]
        sc = sim(list(zip([cur] * (n - 1), [other[jj] for jj in range(len(other))])))
        oc = classes[torch.as_tensor(tuple(jj for jj in range(n) if jj != j))]
        oc = oc[torch.argsort(sc, descending=True)]
        for k in ks:
            out[k][0] += int((cls == oc[:k]).sum().item() != 0)
            out[k][1] += int((cls == oc).sum().item() != 0)
    return out
