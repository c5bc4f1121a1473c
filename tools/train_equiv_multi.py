"""Training equivalence at the HEADLINE geometries with a recipe that learns (VERDICT r4 #8-ii): ResNet-50 + ArcFace through
`main.py --config` at 224 x 224 (bs 32: 100 k / 25 k rows in layer1 / layer2 — the streaming kernels and the BN-input-free form are active,
which 64 x 64 images never reach), 40 epochs x 50 steps = 2000 optimizer steps, base rate 0.005 with 5 warm-up epochs.
Variants: fp32 | bf16 (default: BN-input-free form + Gram statistics) | bf16 with PFR_BNFREE=0 (stored form), each with TWO data-order /
initialisation seeds: the seed-to-seed spread of one precision is the yardstick for the gap between precisions.
usage: python tools/train_equiv_multi.py [tag] [epochs] [image_size] [n_val_ids] [seeds, e.g. 3,4] [variants, e.g. f32,bf16] [workers] [arch]
       -> gpurun_out/<tag>_train_equiv_multi.json (rewritten after every run: a run is 12-17 minutes at 224 x 224, loader-bound)"""
import json, os, re, subprocess, sys, tempfile, textwrap, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 40
size = int(sys.argv[3]) if len(sys.argv) > 3 else 224
n_val = int(sys.argv[4]) if len(sys.argv) > 4 else 128
seeds = tuple(int(v) for v in sys.argv[5].split(",")) if len(sys.argv) > 5 else (3, 4)
only = sys.argv[6].split(",") if len(sys.argv) > 6 and sys.argv[6] != "all" else None
workers = int(sys.argv[7]) if len(sys.argv) > 7 else 0      # loader workers (0 = the seed-3 record's setting; the data order does not depend on it)
arch = sys.argv[8] if len(sys.argv) > 8 else "resnet50"     # "swin_t": BASELINE config 4 (the bf16_stored variant does not apply)
common = os.path.join(ROOT, "pets-face-recognition_amd", "configs", "synthetic")
VARIANTS = [("f32", "torch.float32", {}), ("bf16", "torch.bfloat16", {}), ("bf16_stored", "torch.bfloat16", {"PFR_BNFREE": "0"})]
runs = {}
if only:
    VARIANTS = [v for v in VARIANTS if v[0] in only]
for seed in seeds:
    for name, dt, env in VARIANTS:
        with tempfile.TemporaryDirectory() as td:
            cfg = os.path.join(td, f"equiv_{name}.py")
            open(cfg, "w").write(textwrap.dedent(f"""
                import sys, torch
                sys.path.insert(0, {common!r})
                from _common import make as _make
                _make(globals(), arch={arch!r}, n_train_ids=200, n_val_ids={n_val}, photos=8, image_size={size}, train_bs=32,
                      test_bs=64, device='cuda:0', n_epochs={epochs}, n_pairs=400, compute_dtype={dt}, seed={seed}, noise=1.0, workers={workers})
                init_lr = 0.005
                trainer_kwargs = dict(trainer_kwargs, check_val_every_n_epoch={max(1, epochs // 8)})
                _opt0 = optimizer
                def optimizer(model_):
                    opts, scheds = _opt0(model_)       # the three parameter groups of the reference recipe at base rate init_lr
                    ms = ({int(epochs * 0.7)}, {int(epochs * 0.9)})
                    f = lambda e: min(1.0, (e + 1) / 5) * (0.1 ** sum(e >= m for m in ms))
                    return opts, [torch.optim.lr_scheduler.LambdaLR(opts[0], f)]
            """))
            t0 = time.time()
            r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--config", cfg], cwd=td, capture_output=True, text=True,
                               timeout=3000, env=dict(os.environ, **env))
            if r.returncode != 0:
                print(r.stdout[-2000:], r.stderr[-3000:])
                raise SystemExit(1)
            losses = [float(m.group(1)) for m in re.finditer(r"^epoch \d+ step \d+ loss ([\-0-9.eE]+)", r.stdout, flags=re.M)]
            series = {}
            for m in re.finditer(r"^(Val) (ROC AUC|Recall@K=10|Recall@K=5|Accuracy)\t([\-0-9.eE]+)$", r.stdout, flags=re.M):
                series.setdefault(m.group(2), []).append(float(m.group(3)))
            runs[f"{name}_s{seed}"] = {"logged_losses": losses, "per_validation": series, "seconds": round(time.time() - t0, 1)}
            print(f"{name}_s{seed}: {time.time() - t0:.0f} s, final " + ", ".join(f"{k} {v[-1]:.4f}" for k, v in series.items()), flush=True)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            json.dump({"runs": runs}, open(os.path.join(ROOT, "gpurun_out", f"{tag}_train_equiv_multi.json"), "w"), indent=1)
final = {k: {m: v[-1] for m, v in r["per_validation"].items()} for k, r in runs.items()}
metrics = sorted(next(iter(final.values())))
summary = {}
for m in metrics:
    per = {v: [final[f"{v}_s{s}"][m] for s in seeds] for v, _, _ in VARIANTS}
    mean = {v: sum(x) / len(x) for v, x in per.items()}
    summary[m] = {"per_variant_and_seed": per, "seeds": list(seeds), "mean": {v: round(x, 4) for v, x in mean.items()},
                  "seed_spread": {v: round(max(x) - min(x), 4) for v, x in per.items()}}
    if "bf16" in mean and "f32" in mean:
        summary[m]["bf16_minus_f32"] = round(mean["bf16"] - mean["f32"], 4)
    if "bf16" in mean and "bf16_stored" in mean:
        summary[m]["bf16_minus_bf16_stored"] = round(mean["bf16"] - mean["bf16_stored"], 4)
out = {"workload": f"{arch} + ArcFace(200 ids), synthetic {size}x{size} (pattern + N(0,1) noise), bs 32, {epochs} epochs x 50 steps, FusedSGD base rate 0.005, "
                   f"5 warm-up epochs, decay at 70 % / 90 %, main.py --config; validation on {n_val * 8} images of {n_val} held-out ids",
       "steps": epochs * 50, "final_validation": summary, "runs": runs}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_train_equiv_multi.json"), "w"), indent=1)
print(json.dumps(out["final_validation"], indent=1))
