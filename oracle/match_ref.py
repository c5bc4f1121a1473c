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


def calc_scores_reference(q, g, k=100, thresholds=(0.9069641, 0.985643)):
    """Restatement of /root/reference/generate_tsv.py:63-125 (`similarity_f`, `mean_strategy_cal_scores`, `calc_scores`)
    on ragged arrays.  q, g: tuples (head [P,D], head_seg [n+1], body [P,D], body_seg [n+1], type [n]) of float32 /
    int arrays.  → list of (query card index, top1, mean3, mean10, [gallery card indices, best first]); a query whose
    candidate list is empty yields no row (:114).  Pinned by tests/golden/calc_scores.npz (written by running the
    reference's own functions, oracle/make_golden.py:gen_calc_scores)."""
    qh, qhs, qb, qbs, qt = q
    gh, ghs, gb, gbs, gt = g

    def mean_strategy(v1, v2):                      # :71-78 — every photo pair, (cos+1)/2, mean, clamp(min=0)
        a = torch.from_numpy(np.repeat(v1, len(v2), axis=0))
        b = torch.from_numpy(np.tile(v2, (len(v1), 1)))
        return torch.mean(similarity_f(a, b)).clamp(min=0.0).item()

    rows = []
    for i in range(len(qt)):
        v1, v1_body, type_ = qh[qhs[i]:qhs[i + 1]], qb[qbs[i]:qbs[i + 1]], int(qt[i])
        cand = []
        for j in range(len(gt)):
            if int(gt[j]) != type_:                 # :100-101
                continue
            v2, v2_body = gh[ghs[j]:ghs[j + 1]], gb[gbs[j]:gbs[j + 1]]
            s0 = mean_strategy(v1, v2) if len(v1) and len(v2) else 0
            s1 = mean_strategy(v1_body, v2_body) if len(v1_body) and len(v2_body) else 0
            if s0 + s1 == 0:                        # :107-108
                continue
            cand.append((j, s1 if len(v1) == 0 or (s0 == 0 and s1 > thresholds[type_ - 1]) else s0))   # :109
        cand.sort(key=lambda x: x[1], reverse=True)  # stable: ties keep gallery order
        if cand:
            sc = [c[1] for c in cand]
            rows.append((i, sc[0], float(np.mean(sc[:3])), float(np.mean(sc[:10])), [c[0] for c in cand[:k]]))
    return rows


def calc_scores_case(seed=11, Q=24, G=260, D=32, n_id=40):
    """ragged synthetic cards: identities shared between query and gallery cards so that some body scores clear the
    fusion thresholds; cards without head vectors, without body vectors, and without either"""
    rs = np.random.RandomState(seed)
    ident_h = rs.randn(n_id, D).astype(np.float32)
    ident_b = rs.randn(n_id, D).astype(np.float32)

    def cards(n, first_ids):
        head, body, hs, bs, types = [], [], [0], [0], []
        for c in range(n):
            i = first_ids[c] if c < len(first_ids) else rs.randint(n_id)
            types.append(1 + i % 2)
            u = rs.rand()
            nh = 0 if u < 0.25 else rs.randint(1, 5)
            nb = 0 if 0.2 < u < 0.4 or u > 0.93 else rs.randint(1, 4)     # u in (0.2, 0.25): neither modality
            noise = [0.05, 0.15, 0.6][rs.randint(3)]
            head.extend(ident_h[i] + noise * rs.randn(D).astype(np.float32) for _ in range(nh))
            body.extend(ident_b[i] + noise * rs.randn(D).astype(np.float32) for _ in range(nb))
            hs.append(len(head))
            bs.append(len(body))
        return (np.stack(head).astype(np.float32), np.array(hs), np.stack(body).astype(np.float32), np.array(bs),
                np.array(types))

    return cards(Q, list(range(Q))), cards(G, list(range(n_id)) * 2)
