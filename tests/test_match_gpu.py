"""Embedding match / candR@K on the HIP path vs the CPU oracle and the fixtures produced by the reference's own
Controller.test_epoch_end (tests/golden/recall.npz, see oracle/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


@pytest.mark.parametrize("name", ["n256", "n400", "ties"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_recall_at_k_identical_to_reference(name, dtype):
    from pets_face_recognition_amd.match import recall_at_k
    G = np.load(os.path.join(GOLD, "recall.npz"))
    emb = torch.tensor(G[f"{name}_emb"]).to(DEV)
    cls = torch.tensor(G[f"{name}_classes"]).to(DEV)
    got = recall_at_k(emb, cls, (10, 100), compute_dtype=dtype)
    for k in (10, 100):
        assert got[k] == G[f"{name}_recall{k}_counts"].tolist(), (k, got[k])
        if name != "ties":   # the reference's own ratio (its unstable argsort makes it undefined under ties)
            assert abs(got[k][0] / got[k][1] - float(G[f"{name}_recall{k}_ref"])) < 1e-12


def test_pair_similarity_vs_reference():
    from pets_face_recognition_amd.match import pair_similarity
    G = np.load(os.path.join(GOLD, "recall.npz"))
    emb = torch.tensor(G["n256_emb"]).to(DEV)
    pairs = G["n256_pairs"]
    sc = pair_similarity(emb, pairs[:, 0], pairs[:, 1])
    torch.cuda.synchronize()
    assert torch.allclose(sc.cpu(), torch.tensor(G["n256_pair_scores"]), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_query_gallery_topk_chunked(dtype):
    """multi-chunk running top-K (radix-select first chunk, threshold-filter later chunks) == oracle full sort"""
    from oracle import match_ref
    from pets_face_recognition_amd.match import cosine_topk
    g = torch.Generator().manual_seed(5)
    Q, Gn, D, K = 300, 20000, 512, 100
    centers = torch.randn(500, D, generator=g)
    gcls = torch.randint(0, 500, (Gn,), generator=g)
    gal = centers[gcls] + 2.5 * torch.randn(Gn, D, generator=g)
    qcls = torch.randint(0, 500, (Q,), generator=g)
    qry = centers[qcls] + 2.5 * torch.randn(Q, D, generator=g)
    gal[777] = gal[123]           # exact duplicate rows: tie → lower index first
    rs, ri = match_ref.topk_query_gallery(qry, gal, K)
    sc, idx = cosine_topk(qry.to(DEV), gal.to(DEV), K, compute_dtype=dtype, chunk=4096)
    torch.cuda.synchronize()
    idx = idx.cpu().long()
    # identical candidate sets; order may differ only where fp32 scores differ by rounding of the summation order
    same_set = [(set(idx[i].tolist()) == set(ri[i].tolist())) for i in range(Q)]
    assert sum(same_set) >= Q - 2, f"{Q - sum(same_set)} queries with a different top-{K} set"
    assert torch.allclose(sc.cpu(), rs, rtol=1e-4, atol=1e-5)
    a = match_ref.cand_recall_query_gallery(idx, qcls, gcls, (10, 100))
    b = match_ref.cand_recall_query_gallery(ri, qcls, gcls, (10, 100))
    assert a == b
    # small-gallery edge: fewer rows than K
    sc2, idx2 = cosine_topk(qry[:5].to(DEV), gal[:40].to(DEV), K, compute_dtype=dtype)
    torch.cuda.synchronize()
    assert (idx2[:, 40:] == -1).all() and (idx2[:, :40] >= 0).all()


def test_card_match_mean_strategy_vs_reference_formula():
    """generate_tsv.py:71-78: card score = clamp(mean over the photo cross product of (cos+1)/2, 0) — literal double loop
    on CPU vs the centroid-GEMM path on the GPU."""
    from pets_face_recognition_amd.match import card_match
    g = torch.Generator().manual_seed(8)
    D, nq, ng = 512, 17, 230
    qn = torch.randint(1, 5, (nq,), generator=g); gn = torch.randint(1, 6, (ng,), generator=g)
    qseg = torch.cat([torch.zeros(1, dtype=torch.long), qn.cumsum(0)]); gseg = torch.cat([torch.zeros(1, dtype=torch.long), gn.cumsum(0)])
    centers = torch.randn(40, D, generator=g)
    gcard_cls = torch.randint(0, 40, (ng,), generator=g); qcard_cls = torch.randint(0, 40, (nq,), generator=g)
    gemb = torch.cat([centers[gcard_cls[i]] + 1.5 * torch.randn(int(gn[i]), D, generator=g) for i in range(ng)])
    qemb = torch.cat([centers[qcard_cls[i]] + 1.5 * torch.randn(int(qn[i]), D, generator=g) for i in range(nq)])
    from oracle.arcface_ref import similarity_f
    ref = torch.zeros(nq, ng)
    for i in range(nq):
        a = qemb[qseg[i]:qseg[i + 1]]
        for j in range(ng):
            b = gemb[gseg[j]:gseg[j + 1]]
            pa = a.repeat_interleave(b.shape[0], 0); pb = b.repeat(a.shape[0], 1)
            ref[i, j] = similarity_f(pa, pb).mean().clamp(min=0.0)
    rs, ri = torch.sort(ref, dim=1, descending=True, stable=True)
    for dt in (torch.float32, torch.bfloat16):
        sc, idx = card_match(qemb.to(DEV), qseg, gemb.to(DEV), gseg, k=100, compute_dtype=dt)
        torch.cuda.synchronize()
        assert torch.allclose(sc.cpu(), rs[:, :100], rtol=1e-5, atol=1e-6)
        assert (idx.cpu().long() == ri[:, :100]).float().mean() > 0.999


def test_match_edge_cases():
    """single query, every class a singleton (the reference's ratio would be 0/0: counts must be [0, 0]), N = 2."""
    from oracle import match_ref
    from pets_face_recognition_amd.match import cosine_topk, recall_at_k
    g = torch.Generator().manual_seed(9)
    emb = torch.randn(37, 512, generator=g)
    # singletons only
    got = recall_at_k(emb.to(DEV), torch.arange(37).to(DEV), (10, 100), compute_dtype=torch.float32)
    assert got[10] == [0, 0] and got[100] == [0, 0]
    assert match_ref.recall_at_k_loop(emb, torch.arange(37), (10, 100)) == got
    # two items of one class: each finds the other at rank 1
    got = recall_at_k(emb[:2].to(DEV), torch.zeros(2, dtype=torch.long).to(DEV), (10, 100), compute_dtype=torch.float32)
    assert got[10] == [2, 2] and got[100] == [2, 2]
    # one query against a ragged gallery (not a multiple of any tile), k = 1
    rs, ri = match_ref.topk_query_gallery(emb[:1], emb[1:], 1)
    sc, idx = cosine_topk(emb[:1].to(DEV), emb[1:].to(DEV), 1, compute_dtype=torch.float32)
    torch.cuda.synchronize()
    assert idx.cpu().long().tolist() == ri.tolist() and torch.allclose(sc.cpu(), rs, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_filter_exclude_self_and_overflow_fallback(dtype):
    """The top-K filter fused into the match GEMM (chunks after the first): (1) all-vs-all with the diagonal excluded gives
    the unfused result; (2) a gallery ordered by ASCENDING score overflows the candidate buffer → the match is redone on the
    unfused path and is still exact."""
    from oracle import match_ref
    from pets_face_recognition_amd.match import cosine_topk
    g = torch.Generator().manual_seed(21)
    emb = torch.randn(700, 512, generator=g)
    a = cosine_topk(emb.to(DEV), emb.to(DEV), 50, compute_dtype=dtype, chunk=128, exclude_self=True, fused_filter=True)
    b = cosine_topk(emb.to(DEV), emb.to(DEV), 50, compute_dtype=dtype, chunk=128, exclude_self=True, fused_filter=False)
    torch.cuda.synchronize()
    assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0])
    assert not (a[1].cpu().long() == torch.arange(700)[:, None]).any()
    # ascending gallery: row i = cos(theta_i) q0 + sin(theta_i) r with theta decreasing → score vs q0 strictly increasing
    q0 = torch.nn.functional.normalize(torch.randn(1, 512, generator=g), dim=1)
    r = torch.randn(1, 512, generator=g)
    r = torch.nn.functional.normalize(r - (r * q0).sum() * q0, dim=1)
    th = torch.linspace(1.5, 0.05, 9000)[:, None]
    gal = torch.cos(th) * q0 + torch.sin(th) * r
    qs = q0.repeat(3, 1)
    sc, idx = cosine_topk(qs.to(DEV), gal.to(DEV), 10, compute_dtype=dtype, chunk=4096, fused_filter=True)
    torch.cuda.synchronize()
    rs, ri = match_ref.topk_query_gallery(qs, gal, 10)
    assert idx.cpu().long().tolist() == ri.tolist()
    assert torch.allclose(sc.cpu(), rs, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["n256", "n400", "ties"])
def test_controller_eval_on_gpu_equals_reference_metrics(name):
    """Controller.test_epoch_end on CUDA embeddings (fused pair-score kernel + GPU candR@K): 'ROC AUC', 'Accuracy' and
    Recall@K equal to what the REFERENCE's Controller.test_epoch_end printed for the same embeddings (recall.npz); and a
    custom config.similarity_f (not the flagged default cosine) is honoured on CUDA (reference controller.py:62)."""
    from pets_face_recognition_amd.engine.controller import Controller
    G = np.load(os.path.join(GOLD, "recall.npz"))
    emb = torch.tensor(G[f"{name}_emb"])
    classes = torch.tensor(G[f"{name}_classes"])
    pairs = [tuple(p) for p in G[f"{name}_pairs"].tolist()]
    plabels = G[f"{name}_plabels"].tolist()

    class PG:
        corrected_indices = pairs
        labels = plabels

    calls = []

    def default_sim(ps):
        t1 = torch.stack([p[0] for p in ps]); t2 = torch.stack([p[1] for p in ps])
        return (torch.nn.functional.cosine_similarity(t1, t2) + 1) / 2
    default_sim._is_default_cosine = True

    def custom_sim(ps):
        calls.append(len(ps))
        t1 = torch.stack([p[0] for p in ps]); t2 = torch.stack([p[1] for p in ps])
        return -(t1 - t2).norm(dim=1)

    class Cfg(dict):
        def pair_generator(self, i):
            return "Val", PG
        match_dtype = torch.float32

    N = emb.shape[0]
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(1))
    batches = [{"emb": emb[perm][i:i + 20].to(DEV), "label": classes[perm][i:i + 20].to(DEV), "index": perm[i:i + 20].to(DEV)}
               for i in range(0, N, 20)]
    c = Controller.__new__(Controller)
    torch.nn.Module.__init__(c)
    c.logger, c.current_epoch, c.last_metrics = None, 0, {}
    cfg = Cfg(); cfg.similarity_f = default_sim
    c.config = cfg
    m = c.test_epoch_end([batches])["Val"]
    assert abs(m["ROC AUC"] - float(G[f"{name}_auc_ref"])) < 1e-6
    assert abs(m["Accuracy"] - float(G[f"{name}_acc_ref"])) < 1e-9
    if name != "ties":
        for k in (10, 100):
            assert abs(m[f"Recall@K={k}"] - float(G[f"{name}_recall{k}_ref"])) < 1e-12
    cfg.similarity_f = custom_sim
    m2 = c.test_epoch_end([batches])["Val"]
    assert calls == [len(pairs)]
    from pets_face_recognition_amd.engine import metrics as M
    ref_scores = custom_sim([(emb[a], emb[b]) for a, b in pairs])
    assert abs(m2["ROC AUC"] - M.auroc(ref_scores, torch.tensor(plabels))) < 1e-6


def _calc_scores_golden():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "calc_scores.npz"))
    q = [z[k] for k in ("q_head", "q_head_seg", "q_body", "q_body_seg", "q_type")]
    g = [z[k] for k in ("g_head", "g_head_seg", "g_body", "g_body_seg", "g_type")]
    return z, q, g


def _check_calc_scores(res_idx, res_sc, cnt, top1, m3, m10, rows):
    """rows: reference-order (query, top1, mean3, mean10, answer list).  Order must agree except between entries whose
    reference scores are closer than fp32 accumulation noise."""
    have = {r[0]: r for r in rows}
    for qi in range(res_idx.shape[0]):
        if qi not in have:
            assert cnt[qi] == 0 and (res_idx[qi] == -1).all()
            continue
        _, t1, rm3, rm10, ans = have[qi]
        n = len(ans)
        assert cnt[qi] == n and (res_idx[qi, n:] == -1).all()
        assert abs(top1[qi] - t1) < 2e-6 and abs(m3[qi] - rm3) < 2e-6 and abs(m10[qi] - rm10) < 2e-6
        got = res_idx[qi, :n].tolist()
        assert sorted(got) == sorted(ans) or n == 100          # same candidate set when nothing is cut off
        sc = res_sc[qi, :n]
        for p, (a, b) in enumerate(zip(got, ans)):
            if a != b:                                          # a swap is only admissible inside a near-tie
                assert b in got and abs(sc[p] - sc[got.index(b)]) < 2e-6, (qi, p, a, b)


@pytest.mark.gpu
def test_calc_scores_fusion_rule_equals_reference_rows():
    """generate_tsv.py:91-125 on the device (centroid GEMMs + pfr_card_fuse_scores + running top-100) vs the rows the
    reference's own calc_scores wrote into tests/golden/calc_scores.npz, and vs the oracle at a larger ragged case"""
    from pets_face_recognition_amd.match import calc_scores, calc_scores_db
    from oracle.match_ref import calc_scores_reference
    z, q, g = _calc_scores_golden()
    dev = lambda t: [torch.from_numpy(t[0]).to(DEV), t[1], torch.from_numpy(t[2]).to(DEV), t[3], t[4]]
    for chunk in (32768, 64):                                   # one chunk / several gallery chunks
        r = calc_scores(*dev(q), *dev(g), k=100, chunk=chunk)
        rows = [(int(qi), t1, a, b, ans[ans >= 0].tolist())
                for qi, t1, a, b, ans in zip(z["rows_query"], z["top1"], z["mean3"], z["mean10"], z["answer"])]
        _check_calc_scores(r["idx"].cpu().numpy(), r["scores"].cpu().numpy(), r["count"].cpu().numpy(), r["top1"].cpu().numpy(),
                           r["mean3"].cpu().numpy(), r["mean10"].cpu().numpy(), rows)

    # the reference's own argument / return types
    from pathlib import Path
    def db(prefix, t):
        h, hs, b, bs, ty = t
        return {Path(f"/cards/{prefix}{c:04d}"): {"head_vectors": [torch.from_numpy(v) for v in h[hs[c]:hs[c + 1]]],
                                                  "body_vectors": [torch.from_numpy(v) for v in b[bs[c]:bs[c + 1]]],
                                                  "type": int(ty[c])} for c in range(len(ty))}
    out = calc_scores_db(db("q", q), db("g", g), device=DEV)
    assert [o[0] for o in out] == [f"q{int(i):04d}" for i in z["rows_query"]]
    for o, t1, ans in zip(out, z["top1"], z["answer"]):
        assert abs(o[1] - t1) < 2e-6
        names = o[4].split(",")
        assert len(names) == int((ans >= 0).sum()) and names[0] == f"g{int(ans[0]):04d}"

    from pets_face_recognition_amd.match import create_table, TSV_COLUMNS
    df = create_table({Path("/cards"): (db("q", q), db("g", g))}, device=DEV)
    assert tuple(df.columns) == TSV_COLUMNS == ('query', 'matched_1', 'matched_3', 'matched_10', 'answer') and len(df) == len(out)
    assert df["query"].tolist() == [o[0] for o in out] and df["answer"].iloc[0] == out[0][4]

    # a second ragged case (other seed, D=64) against the oracle restatement
    from oracle.match_ref import calc_scores_case
    qq, gg = calc_scores_case(seed=5, Q=16, G=700, D=64, n_id=90)
    rows = calc_scores_reference(qq, gg)
    r = calc_scores(*dev(list(qq)), *dev(list(gg)), k=100, chunk=256)
    _check_calc_scores(r["idx"].cpu().numpy(), r["scores"].cpu().numpy(), r["count"].cpu().numpy(), r["top1"].cpu().numpy(),
                       r["mean3"].cpu().numpy(), r["mean10"].cpu().numpy(), rows)


@pytest.mark.gpu
def test_pair_metrics_device_sort_scan_equals_host_path():
    """engine/metrics on CUDA scores (pfr_pair_curve: bitonic sort + scan in one workgroup) vs the sklearn-validated torch-CPU path:
    identical ROC points, AUROC, AP, best-threshold accuracy — heavy ties, non-power-of-two and degenerate sizes"""
    from pets_face_recognition_amd.engine import metrics as M
    g = torch.Generator().manual_seed(8)
    cases = []
    for P in (20000, 4097, 1000, 2, 1):
        s = torch.rand(P, generator=g)
        cases.append((s, (torch.rand(P, generator=g) < 0.5).long()))
        cases.append(((s * 20).round() / 20, (torch.rand(P, generator=g) < 0.3).long()))      # 21 distinct values: long tie runs
    cases.append((torch.full((513,), 0.25), (torch.arange(513) % 3 == 0).long()))             # one run
    cases.append((torch.tensor([0.9, -0.5, float("inf"), 0.0, -0.0, 1e-30]), torch.tensor([1, 0, 1, 0, 1, 0])))
    for s, y in cases:
        fc, tc, thc = M.roc_curve(s, y)
        fd, td, thd = M.roc_curve(s.to(DEV), y.to(DEV))
        assert torch.equal(fc, fd) and torch.equal(tc, td) and torch.equal(thc.float(), thd.float()), s.numel()
        assert M.auroc(s, y) == M.auroc(s.to(DEV), y)
        assert M.average_precision(s, y) == M.average_precision(s.to(DEV), y.to(DEV))
        assert M.best_threshold_accuracy(s, y, thc, fc, 1 - tc) == M.best_threshold_accuracy(s.to(DEV), y.to(DEV), thd, fd, 1 - td)


def test_pair_similarity_rejects_out_of_range_pairs():
    """the reference's `emb[idx]` raises IndexError; the device gather must not read outside the embedding matrix instead"""
    from pets_face_recognition_amd.match import pair_similarity
    emb = torch.randn(64, 512, device=DEV)
    with pytest.raises(IndexError):
        pair_similarity(emb, [0, 3, 700], [1, 2, 5])
    with pytest.raises(IndexError):
        pair_similarity(emb, [0, 3], [1, -2])
    assert pair_similarity(emb, [0, 63], [63, 0]).shape == (2,)


@pytest.mark.parametrize("dist", ["normal", "uniform", "constant", "bimodal", "heavy_tail", "ascending", "ties_at_top", "two_values"])
@pytest.mark.parametrize("K", [10, 100, 256])
def test_first_chunk_topk_pivot_paths_are_exact(dist, K):
    """pfr_topk_update on a first chunk of 65 536 scores per row picks its candidates with a pivot (normal approximation, then a
    sorted sample, then the radix select): whichever path decides, the result must be the exact top-K in (score descending,
    lower index first) order — checked on score distributions that defeat the first one or two guesses."""
    from pets_face_recognition_amd._hip import lib
    rows, n = 6, 65536
    g = torch.Generator().manual_seed(len(dist) * 1000 + K)
    if dist == "normal":
        s = torch.randn(rows, n, generator=g) * 0.05
    elif dist == "uniform":
        s = torch.rand(rows, n, generator=g) * 2 - 1
    elif dist == "constant":
        s = torch.full((rows, n), 0.25)
    elif dist == "bimodal":
        s = torch.where(torch.rand(rows, n, generator=g) < 0.5, torch.randn(rows, n, generator=g) * 0.01 - 0.8, torch.randn(rows, n, generator=g) * 0.01 + 0.7)
    elif dist == "heavy_tail":
        s = torch.randn(rows, n, generator=g).abs().pow(6) * 1e-3
    elif dist == "ascending":
        s = torch.linspace(-1, 1, n).repeat(rows, 1) + torch.arange(rows).unsqueeze(1) * 1e-3
    elif dist == "ties_at_top":
        s = torch.randn(rows, n, generator=g) * 0.05
        s[:, torch.randperm(n, generator=g)[:3 * K]] = 0.9
    else:
        s = torch.where(torch.rand(rows, n, generator=g) < 0.3, torch.tensor(0.5), torch.tensor(-0.5))
    s = s.to(DEV).contiguous()
    state = torch.empty(lib.pfr_topk_state_bytes(rows, K), dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    lib.pfr_topk_reset(state.data_ptr(), rows, K, st)
    lib.pfr_topk_update(s.data_ptr(), rows, n, n, 0, K, state.data_ptr(), 0, st)
    sc = torch.empty((rows, K), dtype=torch.float32, device=DEV)
    idx = torch.empty((rows, K), dtype=torch.int32, device=DEV)
    lib.pfr_topk_finish(state.data_ptr(), rows, K, sc.data_ptr(), idx.data_ptr(), st)
    flag = __import__("ctypes").c_int(0)
    lib.pfr_topk_flags(state.data_ptr(), rows, K, __import__("ctypes").addressof(flag), st)
    order = torch.argsort(s, dim=1, descending=True, stable=True)[:, :K]
    if flag.value & 1:      # more entries tied with the K-th value than the tie buffer holds: flagged, not silently wrong
        assert dist in ("constant", "two_values", "ties_at_top")
        return
    assert torch.equal(idx.long(), order), dist
    assert torch.equal(sc, torch.gather(s, 1, order))


@pytest.mark.parametrize("name", ["n256", "n400", "ties", "edge"])
def test_controller_evaluate_on_gpu_equals_reference_evaluate(name):
    """Controller._evaluate on CUDA embeddings (fused pair-score kernel, pfr_pair_curve sort + scan, GPU candR@K) → the metrics
    dict equals, key for key, what the REFERENCE's own Controller._evaluate printed for the same embeddings / pairs / thrs / k /
    far_thr / frr_thr (tests/golden/evaluate.npz, engine/controller.py:95-183).  "edge" has impostor pairs of identical rows and
    genuine pairs of opposite rows: whether torch-CPU's cosine returns exactly 1 / 0 there is rounding luck the reference's skip
    rule (`thr not in (0, 1)`) depends on, so that set feeds the reference's pair scores through config.similarity_f (a custom
    similarity is honoured on CUDA) and pins the device sort / scan / index rules on them."""
    from test_oracle_golden import _evaluate_case, check_evaluate_against_reference
    sim = None
    if name == "edge":
        sc = torch.tensor(np.load(os.path.join(GOLD, "evaluate.npz"))["edge_pair_scores"]).to(DEV)
        sim = lambda pairs: sc       # noqa: E731
    m, cm, gold, gold_cm = _evaluate_case(name, device=DEV, match_dtype=torch.float32, similarity_f=sim)
    check_evaluate_against_reference(m, cm, gold, gold_cm, name)
    # bf16 candidates + fp32 re-score: the same Recall@K
    m2, _, _, _ = _evaluate_case(name, device=DEV, match_dtype=torch.bfloat16, similarity_f=sim)
    for k in m:
        if k.startswith("Recall@K="):
            assert m2[k] == m[k], (name, k)


@pytest.mark.gpu
def test_prepared_gallery_handle_equals_the_one_shot_match():
    """match.prepare_gallery (round 5): reuse of the normalised gallery is an EXPLICIT handle — same top-k as the one-shot call
    (which re-scores from the raw rows x 1/|row|), for several query batches; a handle built for other settings is refused; refilling the
    caller's gallery buffer in place does not change what the handle returns (it owns its copies) but changes the one-shot result."""
    from pets_face_recognition_amd.match import cosine_topk, prepare_gallery
    from pets_face_recognition_amd._hip import PfrError
    g = torch.Generator().manual_seed(5)
    gal = torch.randn(70000, 128, generator=g).to(DEV)
    pg = prepare_gallery(gal)
    for s in range(2):
        q = torch.randn(300, 128, generator=g).to(DEV)
        s1, i1 = cosine_topk(q, gal, 20, chunk=16384)
        s2, i2 = cosine_topk(q, pg, 20, chunk=16384)
        assert torch.equal(i1, i2)
        assert torch.allclose(s1, s2, rtol=0, atol=2e-6)
    with pytest.raises(PfrError):
        cosine_topk(q, pg, 20, compute_dtype=torch.float32)
    with pytest.raises(PfrError):
        cosine_topk(q, pg, 20, rescore=False)
    keep = i2.clone()
    gal.copy_(torch.randn(70000, 128, generator=g))          # the caller refills its buffer
    s3, i3 = cosine_topk(q, pg, 20, chunk=16384)
    assert torch.equal(i3, keep)
    s4, i4 = cosine_topk(q, gal, 20, chunk=16384)
    assert not torch.equal(i4, keep)


def _sets_equal_up_to_fp64_near_ties(idx, ref_idx, q, g, k, tol=2e-6):
    """every member of one top-k set that is missing from the other scores within `tol` (fp64 cosine) of the k-th best"""
    qn = torch.nn.functional.normalize(q.double(), dim=1)
    gn = torch.nn.functional.normalize(g.double(), dim=1)
    bad = 0
    for r in range(idx.shape[0]):
        a, b = set(idx[r].tolist()), set(ref_idx[r].tolist())
        if a == b:
            continue
        kth = torch.sort(gn[ref_idx[r].long()] @ qn[r], descending=True).values[k - 1]
        diff = torch.tensor(sorted(a ^ b), dtype=torch.long)
        if ((gn[diff] @ qn[r]) - kth).abs().max() > tol:
            bad += 1
    return bad


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [65536, 16384])
def test_certified_match_recovers_what_a_short_candidate_list_loses(chunk):
    """Round 6: the bf16 candidate selection is checked, not trusted (match.cosine_topk `certify`).  Gallery with 6000 near-copies of one
    direction (fp32 scores 1e-6 apart, bf16 selection errors 1e-3): a 78-entry candidate list cannot contain the exact top-50 of the queries
    near that direction.  The certificate must flag exactly those queries, the wider list must fail too, and the fp32 re-match must return the
    reference's answer; queries elsewhere stay on the fast path.  Unfused (one chunk) and fused-filter (four chunks) schedules."""
    from pets_face_recognition_amd import match
    g = torch.Generator().manual_seed(11)
    D, K = 128, 50
    base = torch.randn(D, generator=g)
    dense = base[None, :] + 0.02 * torch.randn(6000, D, generator=g)
    gal = torch.cat([dense, torch.randn(54000, D, generator=g)])[torch.randperm(60000, generator=g)]
    q_dense = base[None, :] + 0.02 * torch.randn(6, D, generator=g)
    q_far = torch.randn(40, D, generator=g)
    qry = torch.cat([q_far[:20], q_dense, q_far[20:]])
    rs, ri = match.cosine_topk(qry.to(DEV), gal.to(DEV), K, compute_dtype=torch.float32, chunk=chunk)      # the reference's arithmetic
    sc, idx = match.cosine_topk(qry.to(DEV), gal.to(DEV), K, chunk=chunk)
    st = dict(match.last_match_stats)
    assert st["queries"] == 46 and st["candidates"] == 78
    assert st["widened"] == 6 and st["exact"] == 6, st
    assert 1e-4 < st["max_selection_error"] < 2e-2, st
    assert _sets_equal_up_to_fp64_near_ties(idx.cpu(), ri.cpu(), qry, gal, K) == 0
    far = torch.cat([torch.arange(20), torch.arange(26, 46)])
    assert _sets_equal_up_to_fp64_near_ties(idx.cpu()[far], ri.cpu()[far], qry[far], gal, K) == 0
    assert torch.allclose(sc, rs, rtol=0, atol=3e-6)
    # the uncertified call with the same short list does lose members of those six sets (what the check is for) ...
    _, i_unc = match.cosine_topk(qry.to(DEV), gal.to(DEV), K, chunk=chunk, slack=28, certify=False)
    assert _sets_equal_up_to_fp64_near_ties(i_unc.cpu()[20:26], ri.cpu()[20:26], qry[20:26], gal, K) > 0
    # ... and benign data is certified as it is: nothing re-matched
    gal2 = torch.randn(60000, D, generator=g)
    s2, i2 = match.cosine_topk(q_far.to(DEV), gal2.to(DEV), K, chunk=chunk)
    assert match.last_match_stats["widened"] == 0 and match.last_match_stats["exact"] == 0
    _, r2 = match.cosine_topk(q_far.to(DEV), gal2.to(DEV), K, compute_dtype=torch.float32, chunk=chunk)
    assert _sets_equal_up_to_fp64_near_ties(i2.cpu(), r2.cpu(), q_far, gal2, K) == 0


@pytest.mark.gpu
def test_certified_match_with_exclude_self_rematches_the_whole_set():
    """exclude_self identifies the query by row == column in the fused filter: a failed certificate re-matches every query, not a subset"""
    from pets_face_recognition_amd import match
    g = torch.Generator().manual_seed(12)
    D, K = 64, 20
    base = torch.randn(D, generator=g)
    emb = torch.cat([base[None, :] + 0.01 * torch.randn(3000, D, generator=g), torch.randn(30000, D, generator=g)]).to(DEV)
    rs, ri = match.cosine_topk(emb, emb, K, compute_dtype=torch.float32, exclude_self=True, chunk=8192)
    sc, idx = match.cosine_topk(emb, emb, K, exclude_self=True, chunk=8192)
    assert match.last_match_stats["exact"] == emb.shape[0]
    assert (idx != torch.arange(emb.shape[0], device=DEV, dtype=idx.dtype)[:, None]).all()
    assert _sets_equal_up_to_fp64_near_ties(idx.cpu()[:200], ri.cpu()[:200], emb.cpu()[:200], emb.cpu(), K) == 0
    assert torch.allclose(sc, rs, rtol=0, atol=3e-6)
