"""Embedding match: cosine scores, running top-K, candR@K (Recall@K) — the evaluation half of the FE hot path.

Reference call sites: /root/reference/engine/controller.py:60-65,77-90,143-160 (pair scores + per-query ranking),
configs/dog_fe/fe_dogs_config.py:89-93 (`similarity_f`), generate_tsv.py:91-125 (query-vs-gallery top-100).

CUDA tensors run on the gfx950 kernels (normalise → MFMA GEMM per gallery chunk → radix-select/threshold top-K merge,
optional exact fp32 re-scoring of the candidate lists); CPU tensors use plain torch (the reference's own device
behaviour, used by the CPU plumbing config).  Ordering everywhere: score descending, ties → lower index.
"""
import ctypes

import torch

from .._hip import lib, dtype_id, PfrError
from .._hip import ops


def _stream():
    return torch.cuda.current_stream().cuda_stream


class PreparedGallery:
    """An L2-normalised gallery kept on the device for MANY query batches (generate_tsv.py:91-125 loops over query cards; a served
    gallery answers query batch after query batch): the bf16 GEMM operand and, for the exact re-scoring, the fp32 copy.  Reuse is
    EXPLICIT — the caller builds the handle with `prepare_gallery(g)` and passes it to `cosine_topk` in place of `g`; nothing is keyed
    on tensor identity or version counters (writes through raw pointers — every kernel of this library — are invisible to them), so a
    gallery buffer that was refilled needs a new handle.  1 M x 512: 1 GB (bf16) + 2 GB (fp32)."""

    def __init__(self, gn, gn32, compute_dtype, rescore, normalized):
        self.gn, self.gn32, self.gscale = gn, gn32, None
        self.compute_dtype, self.rescore, self.normalized = compute_dtype, rescore, normalized
        self.shape, self.device = tuple(gn.shape), gn.device

    def __len__(self):
        return self.shape[0]


def _prep_rows(x32, T, rescore, normalize, borrow=False):
    """rows of an fp32 matrix -> (GEMM operand in T, fp32 rows for the re-scoring or None[, their per-row scale]).
    borrow=True (one-shot match): the re-scoring reads the caller's RAW rows times 1 / |row| instead of a normalised fp32 copy
    (1 M x 512: 2 GB less written); -> (operand, raw rows, inverse norms)"""
    D = x32.shape[1]
    if normalize and T == torch.bfloat16 and D % 4 == 0 and D <= 2048:
        # one pass per matrix: bf16 GEMM operand + (when re-scoring) the fp32 copy or the inverse norms
        xb = torch.empty(x32.shape, dtype=torch.bfloat16, device=x32.device)
        if borrow and rescore:
            inv = torch.empty(x32.shape[0], dtype=torch.float32, device=x32.device)
            lib.pfr_l2norm_dual(x32.data_ptr(), xb.data_ptr(), 0, inv.data_ptr(), x32.shape[0], D, 1e-12, _stream())
            return xb, x32, inv
        xf = torch.empty(x32.shape, dtype=torch.float32, device=x32.device) if rescore else None
        lib.pfr_l2norm_dual(x32.data_ptr(), xb.data_ptr(), 0 if xf is None else xf.data_ptr(), 0, x32.shape[0], D, 1e-12, _stream())
        return (xb, xf, None) if borrow else (xb, xf)
    if borrow:
        return _prep_rows(x32, T, rescore, normalize) + (None,)
    if normalize:
        xn, _, _ = ops.l2norm_fwd(x32, T)
        xf = ops.l2norm_fwd(x32, torch.float32)[0] if rescore else None
        return xn, xf
    # rows are used as given (card centroids)
    return (x32 if T == torch.float32 else ops.cast(x32, T)), x32


def prepare_gallery(g, compute_dtype=torch.bfloat16, rescore=None, normalize=True):
    """Normalise a gallery once for many `cosine_topk(q, handle, k)` calls (same compute_dtype / rescore / normalize as those calls)."""
    if not g.is_cuda:
        raise PfrError("prepare_gallery: the gallery handle is a device object (CPU tensors go to cosine_topk directly)")
    T = compute_dtype
    if rescore is None:
        rescore = T != torch.float32
    gn, gn32 = _prep_rows(g.float().contiguous(), T, rescore, normalize)
    return PreparedGallery(gn, gn32, T, bool(rescore), bool(normalize))


#: what the last certified `cosine_topk` call on this process did: queries, candidates per query, the largest selection-score error measured,
#: queries re-matched with a wider candidate list, queries re-matched exactly in fp32
last_match_stats = {}
_CERT_SAFETY = 2.0     # a query is certified when its gap is at least this many times the LARGEST selection error measured on any candidate


def cosine_topk(q, g, k, compute_dtype=torch.bfloat16, chunk=131072, exclude_self=False, rescore=None, slack=None,
                normalize=True, fused_filter=True, merge_every=2, seed_cols="auto", certify=None):
    """Top-k gallery rows per query by cosine similarity.  q [Q,D], g [G,D] (any scale; rows are L2-normalised here) or a
    `prepare_gallery(g)` handle (the gallery's normalisation is then not repeated per call).
    → (scores [Q,k] fp32 cosine, idx [Q,k] int32, −1 / −inf padded when fewer than k exist).
    exclude_self: q and g are the same set, the diagonal is skipped (the reference excludes the query itself).
    rescore (default: True for bf16): candidates are selected on bf16-input scores with `slack` extra entries, then
    re-scored exactly in fp32 and re-sorted, so that the final order is the fp32 order.
    certify (default: with rescore): the selection is CHECKED instead of trusted.  Every gallery row outside a query's candidate list has a
    selection score of at most the list's last one, so it can belong to the exact top-k only if its selection error exceeds
    gap = (k-th best fp32 score) − (last selection score).  The re-scoring measures the selection error on every candidate (Q x kc samples of
    the same data); a query whose gap is below twice the largest error seen is matched again with a four times wider slack, and in fp32
    (exactly the reference's arithmetic) if it fails again.  With the check the default slack is max(28, k / 4) instead of max(28, 1.5 k):
    1.5 ms less at 10 k x 1 M x 512 (13.5 instead of 15.0 ms on the box that measured it), same results; `last_match_stats` records what the
    check did."""
    prepared = g if isinstance(g, PreparedGallery) else None
    if not q.is_cuda:
        if prepared is not None:
            raise PfrError("cosine_topk: a PreparedGallery is a device object; pass the CPU gallery tensor itself")
        return _cosine_topk_torch(q, g, k, exclude_self, normalize)
    Q, D = q.shape
    G = g.shape[0]
    T = compute_dtype
    if rescore is None:
        rescore = T != torch.float32
    if certify is None:
        certify = bool(rescore)
    certify = bool(certify and rescore)
    if prepared is not None and (prepared.compute_dtype != T or prepared.rescore != bool(rescore) or prepared.normalized != bool(normalize)
                                 or prepared.shape[1] != D or prepared.device != q.device):
        raise PfrError("cosine_topk: the PreparedGallery was built for other settings "
                       f"({prepared.compute_dtype}, rescore={prepared.rescore}, normalize={prepared.normalized}, D={prepared.shape[1]}, {prepared.device})")
    if k > 512:
        # the running lists of the top-K kernels hold at most 512 entries per query (pfr_match.hip); silently returning
        # fewer columns, or re-scoring past the candidate list, would be wrong answers
        raise PfrError(f"cosine_topk: k={k} exceeds the 512-entry running list of the gfx950 top-K kernels")
    if slack is None:
        slack = max(28, (k + 3) // 4) if certify else max(28, k // 2 + k)
    kc = max(k, min(512, k + slack)) if rescore else k
    q32 = q.float().contiguous()
    qn, qn32 = _prep_rows(q32, T, rescore, normalize)
    if prepared is not None:
        gn, gn32, gscale = prepared.gn, prepared.gn32, None      # (a handle owns a normalised copy: the caller may refill its buffer)
    else:
        gn, gn32, gscale = _prep_rows(g.float().contiguous(), T, rescore, normalize, borrow=True)
    sched = dict(chunk=chunk, exclude_self=exclude_self, fused_filter=fused_filter, merge_every=merge_every, seed_cols=seed_cols)
    if not rescore:
        return _topk_pass(qn, None, gn, None, None, k, k, T, False, **sched)[:2]
    sc, idx, cert = _topk_pass(qn, qn32, gn, gn32, gscale, k, kc, T, certify, **sched)
    if not certify:
        return sc, idx
    # ---- the certificate (see the docstring).  One host synchronisation; the re-matches below run only for the queries that fail.
    eps = cert[:, 0].max()
    bad = (cert[:, 1] < _CERT_SAFETY * eps).nonzero().flatten()
    stats = {"queries": Q, "candidates": kc, "max_selection_error": float(eps), "widened": 0, "exact": 0}
    if bad.numel():
        kc2 = max(k, min(512, k + 4 * slack))
        if kc2 > kc:
            # (with exclude_self the fused filter identifies "self" by row == column: the whole set is matched again, not a subset)
            rows = None if exclude_self else bad
            stats["widened"] = Q if rows is None else int(rows.numel())
            s2, i2, c2 = _topk_pass(qn if rows is None else qn[rows], qn32 if rows is None else qn32[rows], gn, gn32, gscale, k, kc2, T, True, **sched)
            eps = torch.maximum(eps, c2[:, 0].max())
            still = c2[:, 1] < _CERT_SAFETY * eps
            if rows is None:
                sc, idx, bad = s2, i2, still.nonzero().flatten()
            else:
                sc[rows], idx[rows] = s2, i2
                bad = rows[still]
        if bad.numel():
            # fp32 scores of every pair for these queries: the reference's own arithmetic (utils/calc_scores.py: torch.mm on fp32)
            rows = None if exclude_self else bad
            stats["exact"] = Q if rows is None else int(rows.numel())
            if gscale is None:      # gn32 holds the rows the scores are defined on (normalised, or as given with normalize=False)
                s3, i3 = cosine_topk(qn32 if rows is None else qn32[rows], gn32, k, compute_dtype=torch.float32, chunk=chunk,
                                     exclude_self=exclude_self, normalize=False)
            else:                   # gn32 = the caller's raw rows
                s3, i3 = cosine_topk(q32 if rows is None else q32[rows], gn32, k, compute_dtype=torch.float32, chunk=chunk,
                                     exclude_self=exclude_self, normalize=normalize)
            if rows is None:
                sc, idx = s3, i3
            else:
                sc[rows], idx[rows] = s3, i3
    stats["max_selection_error"] = float(eps)
    last_match_stats.clear()
    last_match_stats.update(stats)
    return sc, idx


def _topk_pass(qn, qn32, gn, gn32, gscale, k, kc, T, certify, chunk=131072, exclude_self=False, fused_filter=True, merge_every=2,
               seed_cols="auto"):
    """One pass of prepared query rows over the prepared gallery: running top-kc lists per query (selection scores in T), then — with
    qn32 / gn32 — the exact fp32 re-scoring that keeps k.  → (scores [Q,k], idx [Q,k], certificate [Q,2] or None)."""
    Q, D = qn.shape
    G = gn.shape[0]
    rescore = qn32 is not None
    chunk = min(chunk, G)
    state = torch.empty(lib.pfr_topk_state_bytes(Q, kc), dtype=torch.uint8, device=qn.device)
    self_idx = torch.arange(Q, dtype=torch.int32, device=qn.device) if exclude_self else None
    # Chunks after the first (every running list is full by then) use the GEMM with the top-K filter in its epilogue: the
    # fp32 score chunk is never written.  A candidate-list overflow (adversarially ordered gallery) is flagged on the
    # device and the whole match is redone on the unfused path.
    fused = fused_filter and G > chunk and chunk >= kc + 1 and D % (64 if T == torch.bfloat16 else 32) == 0
    cap = 1536
    cand = torch.empty((Q, cap), dtype=torch.int64, device=qn.device) if fused else None
    sbuf = None
    while True:
        lib.pfr_topk_reset(state.data_ptr(), Q, kc, _stream())
        # gallery segments (first column, columns, filter fused into the GEMM?): the first chunk goes through a materialised fp32 score
        # chunk (its running lists start empty), every later one through the GEMM whose epilogue is the top-K filter.
        # (Round 4 measured a seeded variant — an 8 k-column unfused seed segment, then growing fused segments of the first chunk too:
        #  same results, 20.8 / 19.3 / 18.7 ms for seeds of 8 k / 16 k / 32 k columns against 18.5 ms: the early segments carry several times
        #  the candidates per query and their merges cost more than the unfused first chunk they replace; profiles/r04_match_seed_ab.txt.)
        segs = [(c0, min(chunk, G - c0), bool(fused and c0 > 0)) for c0 in range(0, G, chunk)]
        if seed_cols == "auto":     # (a seed pays once the gallery is several chunks long)
            # (round 6, tools/match_sched_ab.py with the certified 128-entry lists: 131 072-column chunks after a 16 384-column seed 13.4 ms,
            #  65 536 / 32 768 — the round-5 schedule — 13.8, 65 536 / 16 384 13.4-13.7, 262 144 / 32 768 13.7)
            seed_cols = 16384 if (chunk >= 65536 and G >= 262144) else None
        if fused and seed_cols and kc + 1 <= seed_cols < chunk:
            # round 5 (tools/match_seed_ab.py, profiles/r05_ab.txt): an unfused seed of `seed_cols` columns, then fused segments that double
            # up to `chunk`, each of the early ones merged at once (their thresholds are loose).  Round 4 had measured this form SLOWER
            # (20.8 / 19.3 / 18.7 ms for seeds of 8 k / 16 k / 32 k against 18.5): the per-winner atomics of the filter epilogue made the
            # candidate-rich early segments expensive; with the counted epilogue it is 15.2 against 15.5 ms, same results.
            segs, c0, n = [(0, seed_cols, False)], seed_cols, seed_cols
            while c0 < G:
                n = min(max(n, seed_cols) if c0 + n <= chunk else chunk, G - c0)
                segs.append((c0, n, True))
                c0 += n
                n = min(2 * n, chunk)
        ld = (max(n for _, n, f in segs if not f) + 3) // 4 * 4
        if sbuf is None or sbuf.shape[-1] < ld:
            sbuf = torch.empty((Q, 1, 1, ld), dtype=torch.float32, device=qn.device)
        ld = sbuf.shape[-1]
        # the candidate lists of TWO fused chunks are folded into the running lists by one merge (expected candidates per query for two
        # chunks: <= 2 * kc, cap = 1536; an overflow is flagged and the match redone unfused): 18.4 -> 17.9 ms at 10 k x 1 M, same results;
        # every 3 / 4 chunks: 18.3 / 18.7 (the stale threshold lets more candidates through the filter epilogue)
        pending = 0
        for si, (c0, n, seg_fused) in enumerate(segs):
            if seg_fused:
                lib.pfr_match_scores_filter(qn.data_ptr(), gn[c0:c0 + n].data_ptr(), dtype_id(T), Q, n, D, c0, kc, state.data_ptr(),
                                            cand.data_ptr(), cap, int(exclude_self), _stream())
                pending += 1
                if pending >= merge_every or si + 1 == len(segs) or (seed_cols and c0 < 4 * chunk):
                    lib.pfr_topk_merge(cand.data_ptr(), cap, Q, kc, state.data_ptr(), _stream())
                    pending = 0
                continue
            ops.conv2d_fwd(qn.view(Q, 1, 1, D), gn[c0:c0 + n].view(n, 1, 1, D), out=sbuf)
            lib.pfr_topk_update(sbuf.data_ptr(), Q, ld, n, c0, kc, state.data_ptr(), 0 if self_idx is None else self_idx.data_ptr(),
                                _stream())
        if not fused:
            break
        flag = ctypes.c_int(0)
        lib.pfr_topk_flags(state.data_ptr(), Q, kc, ctypes.addressof(flag), _stream())
        if not (flag.value & 2):
            break
        fused = False   # candidate buffer overflow: redo without the fused filter
    sc = torch.empty((Q, kc), dtype=torch.float32, device=qn.device)
    idx = torch.empty((Q, kc), dtype=torch.int32, device=qn.device)
    lib.pfr_topk_finish(state.data_ptr(), Q, kc, sc.data_ptr(), idx.data_ptr(), _stream())
    if rescore:
        sc2 = torch.empty((Q, k), dtype=torch.float32, device=qn.device)
        idx2 = torch.empty((Q, k), dtype=torch.int32, device=qn.device)
        gs = 0 if gscale is None else gscale.data_ptr()
        if certify:
            cert = torch.empty((Q, 2), dtype=torch.float32, device=qn.device)
            lib.pfr_topk_rescore_cert(qn32.data_ptr(), gn32.data_ptr(), gs, Q, D, idx.data_ptr(), sc.data_ptr(), kc, k,
                                      sc2.data_ptr(), idx2.data_ptr(), cert.data_ptr(), _stream())
            return sc2, idx2, cert
        lib.pfr_topk_rescore(qn32.data_ptr(), gn32.data_ptr(), gs, Q, D, idx.data_ptr(), kc, k, sc2.data_ptr(), idx2.data_ptr(), _stream())
        return sc2, idx2, None
    return sc[:, :k].contiguous(), idx[:, :k].contiguous(), None


def _cosine_topk_torch(q, g, k, exclude_self, normalize=True):
    qn = q / q.norm(dim=1, keepdim=True).clamp_min(1e-8) if normalize else q
    gn = g / g.norm(dim=1, keepdim=True).clamp_min(1e-8) if normalize else g
    sc = qn @ gn.t()
    if exclude_self:
        sc.fill_diagonal_(-float("inf"))
    kk = min(k, sc.shape[1] - (1 if exclude_self else 0))
    order = torch.argsort(sc, dim=1, descending=True, stable=True)[:, :kk]
    return torch.gather(sc, 1, order), order.int()


def recall_at_k(emb, classes, ks=(10, 100), compute_dtype=torch.float32):
    """candR@K of the reference's validation protocol (controller.py:77-90): every embedding queries all the OTHERS;
    a hit at K = some same-class item among the K best; denominator = queries that have a same-class other.
    → {k: [hits, denom]}"""
    kmax = min(max(ks), emb.shape[0] - 1)
    _, idx = cosine_topk(emb, emb, kmax, compute_dtype=compute_dtype, exclude_self=True)
    idx = idx.long()
    cls = classes.to(idx.device)
    valid = idx >= 0
    same = (cls[idx.clamp_min(0)] == cls[:, None]) & valid
    counts = torch.bincount(cls - cls.min())
    has = counts[cls - cls.min()] > 1
    out = {}
    for k in ks:
        out[k] = [int((same[:, :k].any(dim=1) & has).sum().item()), int(has.sum().item())]
    return out


def pair_similarity(emb, idx_a, idx_b, eps=1e-8):
    """(cos(emb[a], emb[b]) + 1) / 2 for index pairs — `similarity_f` over `pair_generator.corrected_indices`."""
    ia = torch.as_tensor(idx_a, dtype=torch.int64, device=emb.device).contiguous()
    ib = torch.as_tensor(idx_b, dtype=torch.int64, device=emb.device).contiguous()
    if not emb.is_cuda:
        # CPU tensors: the reference's own call (fe_dogs_config.py:93), bit for bit
        return (torch.nn.functional.cosine_similarity(emb[ia], emb[ib], eps=eps) + 1) / 2
    if ia.numel() and (int(torch.stack([ia.min(), ib.min()]).min()) < 0 or int(torch.stack([ia.max(), ib.max()]).max()) >= emb.shape[0]):
        # the torch indexing of the reference raises here; a device gather would read outside the embedding matrix
        raise IndexError(f"pair index out of range for {emb.shape[0]} embeddings (evaluation truncated by limit_val_batches?)")
    e = emb.float().contiguous()
    out = torch.empty(ia.numel(), dtype=torch.float32, device=emb.device)
    lib.pfr_pair_similarity(e.data_ptr(), e.shape[1], ia.data_ptr(), ib.data_ptr(), ia.numel(), float(eps), out.data_ptr(), _stream())
    return out


def card_centroids(emb, seg, compute_dtype=torch.float32):
    """Mean of the L2-normalised photo embeddings per card.  emb [P, D]; seg: int64 [ncards+1] row offsets."""
    seg = torch.as_tensor(seg, dtype=torch.int64, device=emb.device).contiguous()
    n = seg.numel() - 1
    if not emb.is_cuda:
        e = emb.float() / emb.float().norm(dim=1, keepdim=True).clamp_min(1e-8)
        return torch.stack([e[seg[i]:seg[i + 1]].mean(0) if seg[i + 1] > seg[i] else torch.zeros(e.shape[1]) for i in range(n)])
    e = emb.float().contiguous()
    out = torch.empty((n, e.shape[1]), dtype=torch.float32, device=emb.device)
    lib.pfr_card_centroids(e.data_ptr(), seg.data_ptr(), n, e.shape[1], 1e-8, out.data_ptr(), 0, 0, _stream())
    return out


def card_match(q_emb, q_seg, g_emb, g_seg, k=100, compute_dtype=torch.bfloat16):
    """Mean-strategy card-vs-card ranking of the reference's inference pipeline (generate_tsv.py:71-78, 91-125, one
    modality): score(card_q, card_g) = clamp(mean over photo pairs of (cos+1)/2, min=0); → (scores [Q,k], idx [Q,k]),
    best first.  The mean over the photo cross product equals (⟨centroid_q, centroid_g⟩ + 1)/2, so this is ONE GEMM of
    card centroids with the running top-k."""
    qc = card_centroids(q_emb, q_seg)
    gc = card_centroids(g_emb, g_seg)
    dots, idx = cosine_topk(qc, gc, k, compute_dtype=compute_dtype, normalize=False)
    return ((dots + 1) / 2).clamp_min(0), idx


FUSION_THRESHOLDS = (0.9069641, 0.985643)   # generate_tsv.py:107, indexed by species `type` - 1


def _card_flags(head_seg, body_seg, types, device):
    """flags byte per card: bit 0 = has head vectors, bit 1 = has body vectors, bits 2..7 = species type"""
    hs = torch.as_tensor(head_seg, dtype=torch.int64)
    bs = torch.as_tensor(body_seg, dtype=torch.int64)
    t = torch.as_tensor(types, dtype=torch.int64)
    if not (hs.numel() == bs.numel() == t.numel() + 1):
        raise PfrError("calc_scores: head_seg / body_seg need ncards+1 offsets, types ncards entries")
    if int(t.min()) < 1 or int(t.max()) > 63:
        raise PfrError("calc_scores: species type must be in 1..63")
    f = (hs[1:] > hs[:-1]).long() | ((bs[1:] > bs[:-1]).long() << 1) | (t << 2)
    return f.to(torch.uint8).to(device).contiguous()


def calc_scores(q_head, q_head_seg, q_body, q_body_seg, q_type, g_head, g_head_seg, g_body, g_body_seg, g_type, k=100,
                thresholds=FUSION_THRESHOLDS, chunk=32768):
    """Card-vs-card ranking of the reference's inference pipeline WITH its head/body fusion rule
    (generate_tsv.py:91-125 `calc_scores`), on the device.

    Query cards (the reference's `init_db`) and gallery cards (`extra_db`) are given per modality as ragged photo-embedding
    matrices: `*_head [P, D]` + `*_head_seg [ncards+1]` row offsets (an empty range = the card has no head vectors),
    likewise `*_body`; `*_type [ncards]` is the species id (1 or 2).  Per (query card, gallery card) pair:
        skip when types differ                                                               (:100-101)
        s0 = mean-strategy head score if both have head vectors else 0                        (:103-104, :71-78)
        s1 = mean-strategy body score if both have body vectors else 0                        (:105-106)
        skip when s0 + s1 == 0                                                                (:107-108)
        score = s1 if (query has no head vectors or (s0 == 0 and s1 > thresholds[type-1])) else s0   (:109)
    then descending sort, best `k` (=100) gallery cards, and the (top-1, mean of best 3, mean of best 10) columns (:112-123).

    The mean over the photo cross product of (cos+1)/2 is (⟨centroid_q, centroid_g⟩ + 1)/2, so each modality is one fp32
    GEMM of card centroids per gallery chunk; `pfr_card_fuse_scores` applies the rule in place and the running top-k
    kernel keeps the best k.  Ties → lower gallery index (Python's stable `sorted(reverse=True)` keeps dict order).
    → dict(scores [Q,k] fp32, idx [Q,k] int32 (−1 past the end of a short list), count [Q], top1, mean3, mean10 [Q] fp64).
    A query with count 0 gets no row in the reference; with fewer than 3 / 10 entries the reference raises IndexError,
    here the available ones are averaged."""
    dev = q_head.device if q_head.numel() else q_body.device
    if dev.type != "cuda":
        raise PfrError("calc_scores runs on the gfx950 device only (tests use oracle/match_ref.py as the CPU checker)")
    if k > 512:
        raise PfrError(f"calc_scores: k={k} exceeds the 512-entry running list of the gfx950 top-K kernels")
    qf = _card_flags(q_head_seg, q_body_seg, q_type, dev)
    gf = _card_flags(g_head_seg, g_body_seg, g_type, dev)
    Q, G = qf.numel(), gf.numel()
    if Q > 65535:
        raise PfrError("calc_scores: split the query cards into blocks of <= 65535")

    def cents(emb, seg, n):
        if emb.numel() == 0:
            return None
        return card_centroids(emb.to(dev), seg)

    qc = (cents(q_head, q_head_seg, Q), cents(q_body, q_body_seg, Q))
    gc = (cents(g_head, g_head_seg, G), cents(g_body, g_body_seg, G))
    chunk = min(chunk, G)
    ld = (chunk + 3) // 4 * 4
    sb = [torch.zeros((Q, 1, 1, ld), dtype=torch.float32, device=dev) for _ in range(2)]
    state = torch.empty(lib.pfr_topk_state_bytes(Q, k), dtype=torch.uint8, device=dev)
    lib.pfr_topk_reset(state.data_ptr(), Q, k, _stream())
    thr = (ctypes.c_float * len(thresholds))(*thresholds)
    for c0 in range(0, G, chunk):
        n = min(chunk, G - c0)
        for m in range(2):
            if qc[m] is not None and gc[m] is not None:   # a modality nobody has stays 0 and is masked by the flags
                D = qc[m].shape[1]
                ops.conv2d_fwd(qc[m].view(Q, 1, 1, D), gc[m][c0:c0 + n].view(n, 1, 1, D), out=sb[m])
        lib.pfr_card_fuse_scores(sb[0].data_ptr(), sb[1].data_ptr(), Q, ld, n, c0, qf.data_ptr(), gf.data_ptr(),
                                 ctypes.addressof(thr), len(thresholds), _stream())
        lib.pfr_topk_update(sb[0].data_ptr(), Q, ld, n, c0, k, state.data_ptr(), 0, _stream())
    sc = torch.empty((Q, k), dtype=torch.float32, device=dev)
    idx = torch.empty((Q, k), dtype=torch.int32, device=dev)
    lib.pfr_topk_finish(state.data_ptr(), Q, k, sc.data_ptr(), idx.data_ptr(), _stream())
    skipped = torch.isinf(sc) & (sc < 0)          # pairs the reference `continue`s sort last; drop them
    idx = torch.where(skipped, torch.full_like(idx, -1), idx)
    count = (~skipped).sum(1)
    s64 = torch.where(skipped, torch.zeros_like(sc), sc).double()
    cs = s64.cumsum(1)

    def mean_first(m):
        m_eff = count.clamp(min=1, max=min(m, k))
        return cs.gather(1, (m_eff - 1)[:, None])[:, 0] / m_eff
    return {"scores": sc, "idx": idx, "count": count, "top1": s64[:, 0], "mean3": mean_first(3), "mean10": mean_first(10)}


def calc_scores_db(init_db, extra_db, device="cuda", k=100, thresholds=FUSION_THRESHOLDS):
    """`calc_scores(init_db, extra_db)` with the reference's own argument and return types (generate_tsv.py:91-125):
    both dbs map a card path to {'head_vectors': [tensor], 'body_vectors': [tensor], 'type': int}; → list of
    (query name, top-1 score, mean of best 3, mean of best 10, ','-joined names of the best `k` gallery cards)."""
    from pathlib import Path

    def pack(db):
        names, types, head, body, hs, bs = [], [], [], [], [0], [0]
        for f, card in db.items():
            names.append(Path(f).name)
            types.append(int(card['type']))
            head.extend(card['head_vectors'])
            body.extend(card['body_vectors'])
            hs.append(len(head))
            bs.append(len(body))
        cat = lambda v: torch.stack([t.reshape(-1).float() for t in v]).to(device) if v else torch.empty((0, 0), device=device)
        return names, cat(head), hs, cat(body), bs, types

    qn, qh, qhs, qb, qbs, qt = pack(init_db)
    gn, gh, ghs, gb, gbs, gt = pack(extra_db)
    if not qn or not gn:
        return []
    r = calc_scores(qh, qhs, qb, qbs, qt, gh, ghs, gb, gbs, gt, k=k, thresholds=thresholds)
    idx, cnt = r["idx"].cpu(), r["count"].cpu()
    top1, m3, m10 = r["top1"].cpu(), r["mean3"].cpu(), r["mean10"].cpu()
    rows = []
    for i, name in enumerate(qn):
        if int(cnt[i]) == 0:
            continue
        rows.append((str(name), float(top1[i]), float(m3[i]), float(m10[i]),
                     ','.join(gn[j] for j in idx[i, :int(cnt[i])].tolist())))
    return rows


TSV_COLUMNS = ('query', 'matched_1', 'matched_3', 'matched_10', 'answer')     # generate_tsv.py:129-135


def create_table(db, device="cuda", k=100):
    """`create_table(db)` of the reference (generate_tsv.py:128-142): db maps a folder to (initial_base_dict, extra_base_dict);
    every folder's cards are ranked on the device (calc_scores_db) and the rows are collected into one DataFrame with the
    reference's columns.  `create_table(db).to_csv(path, index=False, sep='\t')` is its TSV (generate_tsv.py:262-264)."""
    import pandas as pd
    rows = []
    for big_folder in db:
        init_db, extra_db = db[big_folder]
        rows.extend(calc_scores_db(init_db, extra_db, device=device, k=k))
    return pd.DataFrame(data=rows, columns=TSV_COLUMNS)


def cosine_topk_sharded(q, g_local, k, g_offset, group=None, **kw):
    """Gallery sharded by rows across the ranks of `group` (each rank holds rows [g_offset, g_offset + len(g_local))),
    queries replicated: local GEMM + top-k per rank, ONE all-gather of the (score, index) lists (k·8 bytes per query
    and rank), merge to the global top-k on every rank.  SURVEY.md §8e."""
    import torch.distributed as dist
    sc, idx = cosine_topk(q, g_local, k, **kw)
    kk = sc.shape[1]
    if kk < k:   # pad short lists so that every rank gathers equal shapes
        pad = k - kk
        sc = torch.cat([sc, torch.full((sc.shape[0], pad), -float("inf"), device=sc.device)], 1)
        idx = torch.cat([idx, torch.full((idx.shape[0], pad), -1, dtype=idx.dtype, device=idx.device)], 1)
    if sc.dtype != torch.float32:
        raise PfrError(f"cosine_topk_sharded: scores must be fp32 for the packed all-gather, got {sc.dtype}")
    if int(g_offset) < 0 or int(g_offset) + int(g_local.shape[0]) >= 2 ** 31:
        raise PfrError(f"cosine_topk_sharded: global gallery rows up to {int(g_offset) + int(g_local.shape[0])} do not fit the int32 "
                       f"index half of the packed (score, index) all-gather")
    gidx = torch.where(idx >= 0, idx + int(g_offset), idx).int()
    world = dist.get_world_size(group)
    # ONE collective: (fp32 score bits, int32 global index) pairs of every rank, [world][Q][k][2] int32
    mine = torch.stack([sc.contiguous().view(torch.int32), gidx], dim=2).contiguous()
    everyone = torch.empty((world,) + tuple(mine.shape), dtype=torch.int32, device=mine.device)
    dist.all_gather(list(everyone.unbind(0)), mine, group=group)     # (views of one buffer: gloo has no all_gather_into_tensor)
    S = everyone[..., 0].contiguous().view(torch.float32).permute(1, 0, 2).reshape(sc.shape[0], world * k)
    I = everyone[..., 1].permute(1, 0, 2).reshape(sc.shape[0], world * k).long()
    # merge: score descending, ties → lower global index
    key_i = torch.where(I >= 0, I, torch.full_like(I, 2 ** 62))
    order = torch.argsort(key_i, dim=1, stable=True)
    S, I = torch.gather(S, 1, order), torch.gather(I, 1, order)
    order = torch.argsort(S, dim=1, descending=True, stable=True)[:, :k]
    return torch.gather(S, 1, order), torch.gather(I, 1, order)
