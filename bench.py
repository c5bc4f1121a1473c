#!/usr/bin/env python
"""bench.py — FE train images/sec on MI355X (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps 100 --warmup 20      (the defaults: SURVEY 8d's 20 warm-up + 100 timed steps)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = forward + backward + optimizer (+ gradient all-reduce when N > 1) of ResNet-50 → ArcFace(10 000 ids) on a
synthetic 224x224x3 batch of 256 images per GPU that is already resident in HBM (BASELINE.json configs[1] / [2]).
Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around every kernel launch of extra,
instrumented steps after the timed region; `cpu_baseline` times the oracle restatement of the reference path
(PyTorch-CPU fp32) on the host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # main + side + comm + RCCL streams must not share hardware queues (see package __init__)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work (SURVEY.md §8d / BASELINE.md §2): conv MACs per image, forward
CONV_GMAC = {"resnet50": 4.0871, "resnet18": 1.8136}
SWIN_GMAC = {"swin_t": 4.490}          # forward MACs per image; fwd+bwd = 3x
CONV1_GMAC = 0.1180
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


# every C-ABI entry point whose launches execute the FLOPs counted by conv_flops_per_img (conv / Linear forward, data and
# weight gradients incl. the fused-epilogue variants, windowed attention)
CONV_FAMILY = ("pfr_conv2d_fwd", "pfr_conv2d_wgrad", "pfr_conv2d_dgrad_join", "pfr_conv2d_dgrad_bn", "pfr_conv2d_dgrad_bn_sub",
               "pfr_conv2d_dgrad_bn_ex", "pfr_conv2d_dgrad_bn_sub_ex", "pfr_conv1x1_stats", "pfr_conv1x1_bn_tail", "pfr_conv1x1_dgrad2_bn",
               "pfr_gemm_act",
               "pfr_window_attn_fwd", "pfr_window_attn_bwd")


def conv_flops_per_img(arch):
    if arch in SWIN_GMAC:
        return 2.0 * 3.0 * SWIN_GMAC[arch] * 1e9
    # fwd + dgrad + wgrad of every conv; the stem has no data gradient
    return 2.0 * (3.0 * CONV_GMAC[arch] - CONV1_GMAC) * 1e9


def build(args, device):
    import pets_face_recognition_amd.models as M
    from pets_face_recognition_amd.losses import SoftmaxBasedMetricLearning
    from pets_face_recognition_amd.optim import FusedSGD

    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(123)
    if args.arch.startswith("swin"):
        backbone = getattr(M, args.arch)(num_classes=512, compute_dtype=dt)
    else:
        backbone = getattr(M, args.arch)(compute_dtype=dt)
        backbone.fc = torch.nn.Linear(backbone.fc.in_features, 512)
    ml = SoftmaxBasedMetricLearning(backbone, args.classes, 512, is_focal=True, arc_margin=True)
    ml.add_margin.compute_dtype = dt
    ml.return_logits = True
    ml = ml.to(device)
    ml.train()
    backbone.hip_engine(device)  # adopt parameters into the flat buffers before the optimizer captures them
    p1 = [p for n, p in ml.module.named_parameters() if "fc" not in n and p.requires_grad]
    p2 = [p for n, p in ml.module.named_parameters() if "fc" in n and p.requires_grad]
    # optimizer groups of the reference recipe (configs/dog_fe/fe_dogs_config.py:123-133)
    groups = [{"lr": 5e-3, "params": p1}] + ([{"lr": 1e-2, "params": p2}] if p2 else []) + \
             [{"lr": 1e-2, "params": ml.add_margin.parameters(), "weight_decay": 1e-4}]
    opt = FusedSGD(groups, 0.01, momentum=0.9)
    return ml, opt


def cpu_baseline(args):
    """The reference path restated on PyTorch-CPU (oracle/), timed on the host cores on a bounded sample."""
    from oracle import resnet_ref, arcface_ref
    if args.arch not in resnet_ref.ARCH:
        return None
    nthreads = min(64, os.cpu_count() or 1)   # more threads than this only adds contention on the 2x64-core host
    torch.set_num_threads(nthreads)
    B = args.cpu_batch
    sd = resnet_ref.init_state_dict(args.arch, 512, seed=0)
    names = resnet_ref.param_names(sd)
    ps = {k: (v.requires_grad_(True) if k in names else v) for k, v in sd.items()}
    g = torch.Generator().manual_seed(123)
    w = (torch.randn(args.classes, 512, generator=g) * 0.02).requires_grad_(True)
    x = torch.rand(B, 3, 224, 224, generator=g)
    y = torch.randint(0, args.classes, (B,), generator=g)
    params = [ps[k] for k in names] + [w]
    opt = torch.optim.SGD(params, 0.01, momentum=0.9)

    def step():
        opt.zero_grad()
        emb = resnet_ref.forward(ps, x, args.arch, train=True)
        loss = arcface_ref.focal_loss(arcface_ref.arc_margin_logits(emb, w, y, 64.0, 0.5), y)
        loss.backward()
        opt.step()

    step()
    t0 = time.perf_counter()
    n = 0
    while n < 2 or (time.perf_counter() - t0 < 12 and n < 10):
        step()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 2), "unit": "images/sec", "cores": nthreads, "kind": "port",
            "sample": f"{n} train steps of {args.arch}+ArcFace(C={args.classes}) at bs={B}, PyTorch-CPU fp32 oracle "
                      f"restatement (torchvision absent), {torch.get_num_threads()} threads"}


def cpu_thread_sweep(args):
    """BASELINE.md §3 protocol for the CPU leg: thread sweep at bs = cpu_batch (one warm-up + one timed step per thread count,
    then two more timed steps at the best), reported with the thread count that won."""
    from oracle import resnet_ref, arcface_ref
    if args.arch not in resnet_ref.ARCH:
        return None
    B = args.cpu_batch
    sd = resnet_ref.init_state_dict(args.arch, 512, seed=0)
    names = resnet_ref.param_names(sd)
    ps = {k: (v.requires_grad_(True) if k in names else v) for k, v in sd.items()}
    g = torch.Generator().manual_seed(123)
    w = (torch.randn(args.classes, 512, generator=g) * 0.02).requires_grad_(True)
    x = torch.rand(B, 3, 224, 224, generator=g)
    y = torch.randint(0, args.classes, (B,), generator=g)
    opt = torch.optim.SGD([ps[k] for k in names] + [w], 0.01, momentum=0.9)

    def step():
        opt.zero_grad()
        emb = resnet_ref.forward(ps, x, args.arch, train=True)
        arcface_ref.focal_loss(arcface_ref.arc_margin_logits(emb, w, y, 64.0, 0.5), y).backward()
        opt.step()

    ncpu = os.cpu_count() or 1
    sweep = {}
    for nt in [t for t in (8, 16, 32, 64, 128) if t <= ncpu] or [ncpu]:
        torch.set_num_threads(nt)
        step()
        t0 = time.perf_counter()
        step()
        sweep[nt] = B / (time.perf_counter() - t0)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    # BASELINE.md §3 protocol at the winning thread count: 2 warm-up + 5 timed steps
    for _ in range(2):
        step()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    v = B * 5 / (time.perf_counter() - t0)
    return {"value": round(v, 2), "unit": "images/sec", "cores": best, "kind": "port",
            "thread_sweep_img_s": {str(k): round(val, 2) for k, val in sweep.items()},
            "sample": f"{args.arch}+ArcFace(C={args.classes}) train steps at bs={B}, PyTorch-CPU fp32 oracle restatement (torchvision "
                      f"absent): thread sweep with 1 warm-up + 1 timed step per count, then 2 warm-up + 5 timed steps at the best "
                      f"({best} threads of {ncpu})"}


def cpu_extra_legs(args):
    """BASELINE.md §3's remaining CPU legs, bounded: (2) ResNet-50 + ArcFace at bs = 256 (1 warm-up + 2 timed steps) and
    (4) Swin-T + ArcFace at bs = 16 (2 warm-up + 5 timed steps), PyTorch-CPU fp32 on the host cores."""
    from oracle import resnet_ref, arcface_ref
    out = {}
    nt = min(64, os.cpu_count() or 1)
    torch.set_num_threads(nt)
    g = torch.Generator().manual_seed(123)

    def leg(fwd, params, B, warm, timed, what):
        w = (torch.randn(args.classes, 512, generator=g) * 0.02).requires_grad_(True)
        x = torch.rand(B, 3, 224, 224, generator=g)
        y = torch.randint(0, args.classes, (B,), generator=g)
        opt = torch.optim.SGD(list(params) + [w], 0.01, momentum=0.9)

        def step():
            opt.zero_grad()
            arcface_ref.focal_loss(arcface_ref.arc_margin_logits(fwd(x), w, y, 64.0, 0.5), y).backward()
            opt.step()
        for _ in range(warm):
            step()
        t0 = time.perf_counter()
        for _ in range(timed):
            step()
        dt = time.perf_counter() - t0
        return {"value": round(B * timed / dt, 2), "unit": "images/sec", "cores": nt, "kind": "port",
                "sample": f"{what} + ArcFace(C={args.classes}) train steps at bs={B}, PyTorch-CPU fp32, {warm} warm-up + {timed} timed, {nt} threads"}

    sd = resnet_ref.init_state_dict("resnet50", 512, seed=0)
    names = resnet_ref.param_names(sd)
    ps = {k: (v.requires_grad_(True) if k in names else v) for k, v in sd.items()}
    out["resnet50_bs256"] = leg(lambda x: resnet_ref.forward(ps, x, "resnet50", train=True), [ps[k] for k in names], 256, 1, 2,
                                "resnet50 (oracle restatement; torchvision absent)")
    del sd, ps
    # Swin-T: the package's models/swin.py on CPU tensors is the plain torch.nn restatement whose embeddings equal the reference's
    # (tests/golden/swin_t.npz); no HIP code runs on this leg
    import pets_face_recognition_amd.models as M
    torch.manual_seed(0)
    sw = M.swin_t(num_classes=512).train()
    out["swin_t_bs16"] = leg(sw, [p for p in sw.parameters() if p.requires_grad], 16, 2, 5, "swin_t (models/swin.py on CPU tensors = torch.nn restatement pinned to the reference's embeddings)")
    return out


def extras(args, device):
    """The other BASELINE configs in the driver-run record: Swin-T bs 128 (config 4), the 10k x 1M match
    (config 5) with its CPU baselines, the eval-mode embedder, and the fp32 (reference arithmetic) train step."""
    import types
    from pets_face_recognition_amd.match import cosine_topk
    out = {}

    def train_rate(arch, batch, dtype, steps, warm):
        a = types.SimpleNamespace(arch=arch, dtype=dtype, classes=args.classes, batch=batch)
        ml, opt = build(a, device)
        g = torch.Generator(device="cpu").manual_seed(7)
        x = torch.rand(batch, 3, 224, 224, generator=g).to(device)
        y = torch.randint(0, args.classes, (batch,), generator=g).to(device)

        def st():
            opt.zero_grad()
            o = ml(x, y)
            o["loss"].backward()
            opt.step()
            return o["loss"]
        for _ in range(warm):
            st()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            l = st()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        fl = conv_flops_per_img(arch) * batch
        res = {"value": round(batch / dt, 1), "unit": "images/sec", "ms_per_step": round(dt * 1e3, 3), "dtype": dtype,
               "loss_finite": bool(torch.isfinite(l).item()),
               "roofline": {"bound": "mfma", "achieved": round(fl / dt / 1e12, 1), "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
                            "frac": round(fl / dt / 1e12 / PEAK_TFLOPS[dtype], 4), "note": "whole step (all kernels), conv/linear FLOPs only"}}
        del ml, opt
        torch.cuda.empty_cache()
        return res

    # the same headline step through the data-parallel code path (FlatDDP bucket reducer + RCCL communicator of ONE rank): the
    # number the N = 1 point of a scaling run should be compared with, and what the comm / side / main stream joins cost
    try:
        import subprocess
        env = dict(os.environ, PFR_FORCE_DDP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "15", "--no-extras",
                            "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, text=True, timeout=600)
        j = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")][-1]
        out["ddp_path_w1"] = {"value": j["value"], "unit": "images/sec", "ms_per_step": j["ms_per_step"], "rccl_ranks": j["config"]["rccl_ranks"]}
    except Exception as e:   # noqa: BLE001
        out["ddp_path_w1"] = {"error": repr(e)[:200]}
    # the same workload END TO END through the drop-in CLI: main.py --config (dataloader workers -> pinned uint8 batches -> copy stream
    # two batches ahead -> device augmentation -> step), reported next to the resident-input headline (VERDICT r2 missing #5)
    try:
        cfg = os.path.join(ROOT, "pets-face-recognition_amd", "configs", "synthetic", "fe_r50_mi355x_pipeline.py")
        ncpu = os.cpu_count() or 8
        env = dict(os.environ, PFR_LIMIT_TRAIN_BATCHES="65", PFR_VAL_IDS="32", PFR_WORKERS=str(max(4, min(16, ncpu - 2))))
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--config", cfg], env=env, cwd=td, capture_output=True,
                               text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("THROUGHPUT ")][-1]
        j = json.loads(line[len("THROUGHPUT "):])
        out["main_py_pipeline"] = {"value": j["train_img_s"], "unit": "images/sec", "steps": 60, "loader_workers": int(env["PFR_WORKERS"]),
                                   "host_cores": ncpu, "prefetch_batches": j["prefetch_batches"],
                                   "input": "uint8 frames from DataLoader workers, pinned, copy stream, device augmentation"}
    except Exception as e:   # noqa: BLE001
        out["main_py_pipeline"] = {"error": repr(e)[:300]}
    out["swin_t_bs128"] = train_rate("swin_t", 128, "bf16", 10, 3)
    out["f32_resnet50_bs256"] = train_rate("resnet50", 256, "f32", 4, 2)
    # eval-mode embedder (Controller.validation_step path)
    a = types.SimpleNamespace(arch="resnet50", dtype="bf16", classes=args.classes, batch=256)
    ml, _ = build(a, device)
    ml.eval()
    x = torch.rand(256, 3, 224, 224).to(device)
    with torch.no_grad():
        for _ in range(3):
            ml(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            e = ml(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    out["eval_r50_bs256"] = {"value": round(256 / dt, 1), "unit": "images/sec", "ms_per_batch": round(dt * 1e3, 3),
                             "roofline": {"bound": "mfma", "achieved": round(2 * CONV_GMAC["resnet50"] * 1e9 * 256 / dt / 1e12, 1),
                                          "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                                          "frac": round(2 * CONV_GMAC["resnet50"] * 1e9 * 256 / dt / 1e12 / PEAK_TFLOPS["bf16"], 4)}}
    del ml
    torch.cuda.empty_cache()
    # config 5: 10 000 x 1 000 000 x 512 cosine match, top-100, candR@10/100 (synthetic per SURVEY 8d)
    Q, G, D, K = 10000, 1000000, 512, 100
    g = torch.Generator(device=device).manual_seed(123)
    ncls = G // 10
    centers = torch.randn(ncls, D, device=device, generator=g)
    gcls = torch.arange(ncls, device=device).repeat_interleave(10)[torch.randperm(G, device=device, generator=g)]
    gal = centers[gcls] + 3.2 * torch.randn(G, D, device=device, generator=g)
    qcls = torch.randint(0, ncls, (Q,), device=device, generator=g)
    qry = centers[qcls] + 3.2 * torch.randn(Q, D, device=device, generator=g)
    from pets_face_recognition_amd.match import prepare_gallery
    cosine_topk(qry, gal, K)                    # warm-up (allocations, first launches)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):                          # (median of three single matches: one shot right after the set-up work reads 0.5-1 ms high)
        t0 = time.perf_counter()
        sc, idx = cosine_topk(qry, gal, K)      # the headline: ONE match of raw embeddings, the gallery's normalisation inside the call
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[1]
    from pets_face_recognition_amd import match as _match
    cert = dict(_match.last_match_stats)        # what the candidate-list certificate did inside the timed call
    pg = prepare_gallery(gal)                   # the served-gallery form: normalised once, many query batches
    cosine_topk(qry, pg, K)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        cosine_topk(qry, pg, K)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt_prep = sorted(ts)[1]
    del pg
    hit = gcls[idx.long().clamp_min(0)] == qcls[:, None]
    m = {"seconds": round(dt, 4), "seconds_prepared_gallery": round(dt_prep, 4),
         "note": "`seconds`: one match of raw fp32 embeddings (query + gallery normalisation inside the timed call), median of three; "
                 "`seconds_prepared_gallery`: the gallery passed as a match.prepare_gallery() handle (one gallery, many query batches)",
         "tflops": round(2.0 * Q * G * D / dt / 1e12, 1), "dtype": "bf16 candidates + fp32 re-score", "certificate": cert,
         "candR10": round(hit[:, :10].any(1).float().mean().item(), 4), "candR100": round(hit.any(1).float().mean().item(), 4),
         "roofline": {"bound": "mfma", "achieved": round(2.0 * Q * G * D / dt / 1e12, 1), "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                      "frac": round(2.0 * Q * G * D / dt / 1e12 / PEAK_TFLOPS["bf16"], 4)}}
    # CPU baselines: (i) the reference's per-query python loop (cost-faithful restatement) at N = 2000, 250 of the queries;
    # (ii) vectorised mm + topk at the full size, queries in blocks of 500
    from oracle import match_ref
    ncpu = min(64, os.cpu_count() or 1)
    torch.set_num_threads(ncpu)
    ecpu = qry[:2000].float().cpu()
    ccpu = qcls[:2000].cpu()
    t0 = time.perf_counter()
    match_ref.recall_loop_reference_cost(ecpu, ccpu, range(0, 2000, 8))
    t_loop = time.perf_counter() - t0
    us_pair = t_loop / (250 * 1999) * 1e6
    galc = gal.float().cpu()
    galc = galc / galc.norm(dim=1, keepdim=True)
    qc = qry.float().cpu()
    qc = qc / qc.norm(dim=1, keepdim=True)
    t0 = time.perf_counter()
    cidx = []
    for i in range(0, Q, 500):
        cidx.append(torch.topk(qc[i:i + 500] @ galc.t(), K, dim=1).indices)
    t_mm = time.perf_counter() - t0
    cidx = torch.cat(cidx)
    agree = float((cidx[:, 0] == idx[:, 0].long().cpu()).float().mean())
    m["cpu_baseline"] = {"loop_N2000_us_per_pair": round(us_pair, 2), "loop_extrapolated_hours_10kx1M": round(us_pair * Q * G / 3.6e9, 1),
                         "mm_topk_seconds": round(t_mm, 2), "cores": ncpu, "kind": "port", "top1_agreement_with_gpu": round(agree, 5),
                         "sample": "reference per-query python loop restated (controller.py:77-90) on 250 of 2000 queries x 1999 others; "
                                   "torch.mm + topk fp32 at 10k x 1M in query blocks of 500"}
    out["match_10kx1M"] = m
    out["augment_bs256"] = augment_extra(device)
    return out


def augment_extra(device):
    """the train Compose pipeline of fe_dogs_config.py:17-26 on a 256 x 224 x 224 x 3 uint8 batch: device rate,
    HBM roofline on the algorithmic bytes (uint8 frame in, float32 NCHW out), and the PIL pipeline on one host core"""
    from pets_face_recognition_amd.data_loading import train_augmentation
    N = 256
    x = torch.randint(0, 256, (N, 224, 224, 3), dtype=torch.uint8, device=device)
    aug = train_augmentation(torch.Generator().manual_seed(3))
    flags, angles = aug.draw(N, 224, 224)
    for _ in range(3):
        aug.apply(x, flags, angles)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        y = aug.apply(x, flags, angles)
    torch.cuda.synchronize()
    dt_e2e = (time.perf_counter() - t0) / 20
    # device-only time of the two kernels: records resident, HIP events on the launch stream
    from pets_face_recognition_amd._hip import lib
    rec = torch.zeros((N, 12), dtype=torch.int32)
    lib.pfr_augment_params(flags.numpy().ctypes.data, angles.numpy().ctypes.data, N, 224, 224, rec.data_ptr())
    rec = rec.to(device)
    ws = torch.empty(lib.pfr_augment_ws_bytes(N, 224, 224), dtype=torch.uint8, device=device)
    st = torch.cuda.current_stream().cuda_stream
    e0.record()
    for _ in range(20):
        lib.pfr_augment_train(x.data_ptr(), N, 224, 224, 220, 220, 224, 224, rec.data_ptr(), y.data_ptr(), ws.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    dt_k = e0.elapsed_time(e1) / 20 / 1e3
    nbytes = N * 224 * 224 * 3 * (1 + 4)
    res = {"value": round(N / dt_e2e, 1), "unit": "images/sec", "ms_per_batch": round(dt_e2e * 1e3, 3), "dtype": "u8",
           "kernels_ms_per_batch": round(dt_k * 1e3, 4),
           "roofline": {"bound": "hbm", "achieved": round(nbytes / dt_k / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(nbytes / dt_k / 8e12, 4), "note": "uint8 frame read + float32 NCHW written, per batch"}}
    try:
        from PIL import Image, ImageEnhance, ImageOps
        xs = x[:64].cpu().numpy()
        fl, an = flags.numpy(), angles.numpy()
        t0 = time.perf_counter()
        for i in range(64):
            im = Image.fromarray(xs[i])
            if fl[i, 0]:
                im = ImageEnhance.Sharpness(im).enhance(0)
            if fl[i, 1]:
                im = ImageOps.autocontrast(im)
            im = im.crop((int(fl[i, 3]), int(fl[i, 2]), int(fl[i, 3]) + 220, int(fl[i, 2]) + 220)).resize((224, 224), Image.BILINEAR)
            im = im.rotate(float(an[i]), Image.NEAREST, fillcolor=(0, 0, 0))
            t = torch.from_numpy(np.asarray(im)).permute(2, 0, 1).float().div(255)
        dt_c = (time.perf_counter() - t0) / 64
        res["cpu_baseline"] = {"value": round(1 / dt_c, 1), "unit": "images/sec", "cores": 1, "kind": "reference",
                               "sample": "the same Pillow calls torchvision's PIL transforms make, 64 images, one dataloader worker"}
    except ImportError:
        res["cpu_baseline"] = None
    return res


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: re-execute this command line as N ranks (one per GPU) under
    torch.distributed.run on 127.0.0.1 with a free port; rank 0's JSON line is this process's only stdout line."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes needs it on this driver)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        sys.stderr.write(r.stdout[-4000:])
        return r.returncode or 1
    print(lines[-1])
    return 0


def launcher_selftest(args):
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "rank_sum": t.item(), "steps": args.steps, "warmup": args.warmup}))
    dist.destroy_process_group()


def match_sharded_leg(device, rank, world, dist):
    """SURVEY 8(e) row 2 on hardware: the 10 k x 1 M x 512 match with the gallery sharded by rows over the ranks (1 M / world rows each,
    125 k at 8 GPUs), queries replicated, ONE all-gather of the per-rank top-100 lists + merge (match.cosine_topk_sharded)."""
    from pets_face_recognition_amd.match import cosine_topk_sharded
    Q, G, D, K = 10000, 1000000, 512, 100
    rows = G // world
    g = torch.Generator(device=device).manual_seed(123)
    qry = torch.randn(Q, D, device=device, generator=g)                  # the same queries on every rank (same seed)
    g.manual_seed(1000 + rank)
    gal = qry[torch.randint(0, Q, (rows,), device=device, generator=g)] + 3.2 * torch.randn(rows, D, device=device, generator=g)
    cosine_topk_sharded(qry, gal, K, rank * rows)
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    sc, idx = cosine_topk_sharded(qry, gal, K, rank * rows)
    torch.cuda.synchronize(); dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    chk = idx.sum().reshape(1).double()
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return {"seconds": round(t.item(), 4), "ranks": world, "gallery_rows_per_rank": rows, "queries": Q, "k": K,
            "tflops": round(2.0 * Q * rows * world * D / t.item() / 1e12, 1), "every_rank_same_result": bool(lo.item() == hi.item()),
            "note": "max over ranks; raw fp32 embeddings in, normalisation + local match + one all-gather + merge inside the timed call"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--classes", type=int, default=10000)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the Swin-T / match / eval / fp32 measurements after the headline")
    ap.add_argument("--detail", default=None, help="write per-launch timings grouped by geometry to this JSON file")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="no GPU work: every rank joins a gloo group, sums the rank ids and rank 0 prints one JSON line (CPU test of the --gpus N launcher path)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if args.launcher_selftest:
        return launcher_selftest(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    use_ddp = world > 1 or os.environ.get("PFR_FORCE_DDP") == "1"   # the switch exercises the RCCL path on one GPU
    if use_ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    rccl_ranks = dist.get_world_size() if dist is not None else 0   # ranks the RCCL communicator actually has (0: no communicator)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    ml, opt = build(args, device)
    ddp = None
    if use_ddp:
        from pets_face_recognition_amd.engine import FlatDDP
        ddp = FlatDDP(ml, bucket_mb=int(os.environ.get("PFR_BUCKET_MB", "25")))

    g = torch.Generator(device="cpu").manual_seed(123 + rank)
    x = torch.rand(args.batch, 3, 224, 224, generator=g).to(device)
    y = torch.randint(0, args.classes, (args.batch,), generator=g).to(device)

    def step():
        opt.zero_grad()
        out = ml(x, y)
        out["loss"].backward()
        if ddp is not None:
            ddp.finish_backward()
        opt.step()
        return out["loss"]

    for _ in range(args.warmup):
        loss = step()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_ms = (time.perf_counter() - t0) / args.steps * 1e3   # host time to ENQUEUE a step (GPU-bound if << ms_per_step)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = float(loss.item())
    # host time to enqueue ONE step into an EMPTY queue (no back-pressure from the GPU): what the host side really costs
    host_free = []
    for _ in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        host_free.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    host_free_ms = min(host_free)
    ms = elapsed / args.steps * 1e3
    value = args.batch * world * args.steps / elapsed

    roof = None
    tr = None
    nprof = 3
    if not args.no_roofline:
        # EVERY rank runs the instrumented steps (they contain the gradient all-reduces); only rank 0 reports them
        from pets_face_recognition_amd._hip import set_tracer, EventTracer
        tr = EventTracer()
        fence()
        set_tracer(tr)
        for _ in range(nprof):
            step()
        set_tracer(None)
        fence()
    if rank == 0 and tr is not None:
        summ = tr.summary()
        # per launch geometry: FLOPs, algorithmic HBM bytes (input + output activations + weights, each once) and the
        # bound time max(FLOPs / MFMA peak, bytes / HBM peak) -- most ResNet-50 layers at bs 256 are HBM-bound, so the
        # whole-family MFMA fraction above cannot approach 1; `layer_bound` is the sum of the per-launch bounds over the
        # measured time of the same launches
        esz = 2 if args.dtype == "bf16" else 4
        det = {}
        for name, a, e0, e1 in tr.records:
            if name == "pfr_conv2d_fwd":
                key = "fwd N%d H%d W%d C%d Co%d R%d s%d dil%d OH%d pro%d" % (a[5], a[6], a[7], a[8], a[9], a[10], a[12], a[14], a[15], 1 if a[22] else 0)
                fl = 2.0 * a[5] * a[15] * a[16] * a[9] * a[10] * a[11] * a[8] / (4 ** a[14])
                by = esz * (a[5] * a[6] * a[7] * a[8] + a[5] * a[15] * a[16] * a[9] + a[9] * a[10] * a[11] * a[8])
            elif name == "pfr_conv2d_wgrad":
                key = "wgrad N%d H%d W%d C%d Co%d R%d s%d OH%d pro%d" % (a[5], a[6], a[7], a[8], a[9], a[10], a[12], a[14], 1 if a[17] else 0)
                fl = 2.0 * a[5] * a[14] * a[15] * a[9] * a[10] * a[11] * a[8]
                by = esz * (a[5] * a[6] * a[7] * a[8] + a[5] * a[14] * a[15] * a[9]) + 4 * a[9] * a[10] * a[11] * a[8]
            elif name == "pfr_conv2d_dgrad_join":
                # (dy, wt, dx, dtype, N, H, W, C, Cout, R, S, pad, idil, OH, OW, res, mask): reads dy + res, writes dx
                key = "dgrad_join N%d H%d W%d C%d Co%d R%d dil%d OH%d" % (a[4], a[5], a[6], a[7], a[8], a[9], a[12], a[13])
                fl = 2.0 * a[4] * a[13] * a[14] * a[8] * a[9] * a[10] * a[7] / (4 ** a[12])
                by = esz * (a[4] * a[5] * a[6] * a[7] + 2 * a[4] * a[13] * a[14] * a[8] + a[8] * a[9] * a[10] * a[7])
            elif name in ("pfr_conv1x1_stats", "pfr_conv1x1_bn_tail"):
                # recompute form of a block's last 1x1 conv: (x, w, dtype, N, H, W, C, Cout, part) / (x, w, y, mask, dtype, N, H, W, C, Cout, ...).
                # The algorithmic FLOPs of the convolution are counted ONCE (on the tail launch); the statistics pass is extra work.
                o = 2 if name == "pfr_conv1x1_stats" else 4
                Nn, Hh, Ww, Cc, Co = a[o + 1], a[o + 2], a[o + 3], a[o + 4], a[o + 5]
                key = "%s N%d H%d W%d C%d Co%d" % (name[4:], Nn, Hh, Ww, Cc, Co)
                fl = 0.0 if name == "pfr_conv1x1_stats" else 2.0 * Nn * Hh * Ww * Cc * Co
                by = esz * Nn * Hh * Ww * Cc + (0 if name == "pfr_conv1x1_stats" else esz * 2 * Nn * Hh * Ww * Co + Nn * Hh * Ww * Co // 8)
            elif name == "pfr_conv1x1_dgrad2_bn":
                # (g, z, wcat, bias, dx, dtype, N, H, W, C1, C2, Cout, bn_x, ...): conv3's BN-input-free data gradient over [g | z]; the
                # algorithmic FLOPs are those of the plain data gradient (C1 -> Cout); the z·S part is extra work
                key = "dgrad2_bn N%d H%d W%d C%d+%d Co%d" % (a[6], a[7], a[8], a[9], a[10], a[11])
                fl = 2.0 * a[6] * a[7] * a[8] * a[9] * a[11]
                by = esz * a[6] * a[7] * a[8] * (a[9] + a[10] + 2 * a[11])
            elif name in ("pfr_conv2d_dgrad_bn", "pfr_conv2d_dgrad_bn_ex"):
                # (dy, wt, dx, dtype, N, H, W, C, Cout, R, S, pad, idil, OH, OW, res, mask, acc, bn_x, ...): the data gradient (+ join)
                # that also reads the BN input for the BatchNorm-backward sums (what pfr_bn_bwd_reduce read in a separate pass)
                key = "dgrad_bn%s N%d H%d W%d C%d Co%d R%d dil%d OH%d" % ("_join" if a[15] else "", a[4], a[5], a[6], a[7], a[8], a[9], a[12], a[13])
                fl = 2.0 * a[4] * a[13] * a[14] * a[8] * a[9] * a[10] * a[7] / (4 ** a[12])
                by = esz * (a[4] * a[5] * a[6] * a[7] + (3 if a[15] else 2) * a[4] * a[13] * a[14] * a[8] + a[8] * a[9] * a[10] * a[7])
            elif name in ("pfr_conv2d_dgrad_bn_sub", "pfr_conv2d_dgrad_bn_sub_ex"):
                # (dy, wt, dx, dtype, N, H, W, C, Cout, OH, OW, res_compact, bn_x, ...): 1x1 data gradient + compact shortcut gradient + BN sums
                key = "dgrad_bn_sub N%d H%d W%d C%d Co%d" % (a[4], a[5], a[6], a[7], a[8])
                fl = 2.0 * a[4] * a[9] * a[10] * a[8] * a[7]
                by = esz * (a[4] * a[5] * a[6] * a[7] + 2.25 * a[4] * a[9] * a[10] * a[8] + a[8] * a[7])
            else:
                key, fl, by = name, 0.0, 0.0
            d = det.setdefault(key, [0, 0.0, fl, by])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
        peak_f, peak_b = PEAK_TFLOPS[args.dtype] * 1e12, 8.0e12
        lb_ms = lb_meas = lb_hbm = 0.0
        for k, v in det.items():
            if v[2]:
                n = v[0] / nprof
                lb_ms += n * max(v[2] / peak_f, v[3] / peak_b) * 1e3
                lb_hbm += n * (v[3] / peak_b) * 1e3
                lb_meas += v[1] / nprof
        if args.detail:
            rows = sorted(((k, v[0] // nprof, v[1] / nprof, v[2], v[3]) for k, v in det.items()), key=lambda r: -r[2])
            with open(args.detail, "w") as f:
                json.dump([{"op": k, "launches_per_step": n, "ms_per_step": round(ms_, 4),
                            "tflops": round(fl * n / (ms_ * 1e-3) / 1e12, 1) if fl else None,
                            "mfma_bound_ms": round(n * fl / peak_f * 1e3, 4) if fl else None,
                            "hbm_bound_ms": round(n * by / peak_b * 1e3, 4) if fl else None,
                            "bound_over_measured": round(n * max(fl / peak_f, by / peak_b) * 1e3 / ms_, 3) if fl else None}
                           for k, n, ms_, fl, by in rows], f, indent=1)
        conv_ms_all = sum(v[1] for k, v in summ.items() if k in CONV_FAMILY) / nprof
        total_ms = sum(v[1] for v in summ.values()) / nprof
        flops = conv_flops_per_img(args.arch) * args.batch
        # roofline.frac = ALL conv / linear FLOPs of the step / the time of ALL conv-family launches / peak.  The data-gradient
        # launches that also do the BatchNorm-backward reduction in their epilogue (pfr_conv2d_dgrad_bn[_sub]) are conv launches: their
        # FLOPs and their time both count.  The ratio without them is kept under its own key (`frac_excl_bn_sum_launches`).
        fused_ms = sum(summ[k][1] for k in ("pfr_conv2d_dgrad_bn", "pfr_conv2d_dgrad_bn_sub", "pfr_conv2d_dgrad_bn_ex", "pfr_conv2d_dgrad_bn_sub_ex", "pfr_conv1x1_dgrad2_bn") if k in summ) / nprof
        fused_flops = sum(v[0] / nprof * v[2] for k, v in det.items() if k.startswith("dgrad_bn") or k.startswith("dgrad2_bn"))
        conv_ms = conv_ms_all
        ach = flops / (conv_ms_all * 1e-3) / 1e12
        ach_excl = (flops - fused_flops) / ((conv_ms_all - fused_ms) * 1e-3) / 1e12 if conv_ms_all > fused_ms else None
        peak = PEAK_TFLOPS[args.dtype]
        traffic = None
        traffic_stale = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.arch == "resnet50" and args.batch == 256 and args.dtype == "bf16":
            # HBM bytes of the conv family per step from rocprofv3 PMC passes of THIS command (FETCH_SIZE x2-corrected
            # + WRITE_SIZE, see profiles/traffic.json); bench.py cannot run the profiler on itself.
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj.get("conv_family_bytes_per_step")
            import glob, hashlib
            h = hashlib.sha256()
            csrc = os.path.join(ROOT, "pets-face-recognition_amd", "csrc")
            for fn in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
                with open(fn, "rb") as f:
                    h.update(f.read())
            traffic_stale = tj.get("csrc_sha256") != h.hexdigest()   # counters collected with other kernel sources
        roof = {"bound": "mfma", "kernel": "all conv / linear launches of a step (forward, data gradient, weight gradient): igemm_kernel (tiles), sconv_kernel / sconv3_kernel (weight-stationary streaming 1x1 / halo-staged 3x3, incl. the data-gradient launches whose epilogue also does the BatchNorm-backward sums), wgrad3_kernel / swgrad_kernel",
                "definition": "conv_flops_per_step / conv_ms_per_step / peak (SURVEY 8d); conv_ms_per_step = HIP-event time of every launch of the CONV_FAMILY entry points",
                "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                "traffic_note": "HBM bytes per step of the same launches (PMC), algorithmic minimum = activations+weights once",
                "traffic_stale": traffic_stale,
                "conv_ms_per_step": round(conv_ms, 3), "all_kernels_ms_per_step": round(total_ms, 3),
                "conv_tflop_per_step": round(flops / 1e12, 4),
                "frac_excl_bn_sum_launches": round(ach_excl / peak, 4) if ach_excl else None,
                "bn_sum_launch_tflop_per_step": round(fused_flops / 1e12, 4),
                # BatchNorm bookkeeping that runs on MFMA but is no layer of the network (Gram matrix of conv3's input: bn3's statistics
                # and backward sums, pfr_gram_colsum) is counted with the BatchNorm launches, not in conv_ms_per_step
                "bn_gram_ms_per_step": round(summ["pfr_gram_colsum"][1] / nprof, 3) if "pfr_gram_colsum" in summ else 0.0,
                # the launches that ALSO do the BatchNorm-backward reduction in their epilogue (pfr_conv2d_dgrad_bn) and what is left
                # of the separate pfr_bn_bwd_reduce pass
                "fused_bn_sums_ms_per_step": round(fused_ms, 3),
                "bn_bwd_reduce_ms_per_step": round(summ["pfr_bn_bwd_reduce"][1] / nprof, 3) if "pfr_bn_bwd_reduce" in summ else 0.0,
                "whole_step_frac": round(flops / (ms * 1e-3) / 1e12 / peak, 4),
                "layer_bound": {"bound_ms": round(lb_ms, 3), "hbm_only_ms": round(lb_hbm, 3), "measured_ms": round(lb_meas, 3),
                                "frac": round(lb_ms / lb_meas, 4) if lb_meas else None,
                                "note": "sum over conv launches of max(FLOPs/MFMA peak, (in+out activations+weights)/8 TB/s) "
                                        "over their measured time; per-geometry rows: --detail / profiles/*layer_roofline*"},
                "by_entry_point_ms": {k: round(v[1] / nprof, 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1][1])}}
    cpu = None
    extra = None
    ms_leg = None
    if dist is not None:      # N > 1 — or the DDP path forced at one rank (PFR_FORCE_DDP=1): the same code on the 1-GPU box
        dist.barrier()
        del ml, opt, ddp
        ml = opt = None
        torch.cuda.empty_cache()
        if not args.no_extras:
            ms_leg = match_sharded_leg(device, rank, world, dist)
            if rank == 0 and world > 1:
                extra = {"match_sharded": ms_leg}
    if rank == 0 and world == 1 and not args.no_extras and args.arch == "resnet50" and args.dtype == "bf16":
        del ml, opt
        torch.cuda.empty_cache()
        extra = extras(args, device)
        if ms_leg is not None:
            extra["match_sharded"] = ms_leg
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_thread_sweep(args)
        if cpu is not None and args.arch == "resnet50" and not args.no_extras:
            try:
                cpu["other_legs"] = cpu_extra_legs(args)   # bs = 256 ResNet-50 and Swin-T bs = 16 on the host cores
            except Exception as e:   # noqa: BLE001
                cpu["other_legs"] = {"error": repr(e)[:300]}

    if rank == 0:
        line = {"metric": f"FE train images/sec @224^2 bs={args.batch}/GPU", "value": round(value, 1), "unit": "images/sec",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "host_enqueue_ms_per_step": round(host_ms, 3),
                "host_enqueue_ms_empty_queue": round(host_free_ms, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": f"{args.arch} FE + ArcFace(s=64,m=0.5) + CE, {args.classes} ids, 224x224x3, "
                                       f"fwd+bwd+SGD(momentum 0.9, param groups of the reference recipe)",
                           "global_batch": args.batch * world,
                           "per_gpu_batch": args.batch, "parallelism": f"dp{world}", "loss": round(final_loss, 4),
                           "rccl_ranks": rccl_ranks},
                "roofline": roof, "cpu_baseline": cpu, "extra": extra}
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
