"""Stored form vs BN-input-free form of conv3 -> bn3 in TRAINING: ResNet-50 + ArcFace at 224x224 (the streaming geometries the
form applies to need >= 16 k rows per layer: bs 32 at 224^2 gives 100 k / 25 k rows in layer1 / layer2) through `main.py --config`, bf16,
identical seeds / data order, PFR_BNFREE=0 vs 1; compares the logged losses and the validation metrics of every epoch.
usage: python tools/train_equiv_bnfree.py [tag] [epochs]      -> gpurun_out/<tag>_train_equiv_bnfree.json"""
import json, os, re, subprocess, sys, tempfile, textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
common = os.path.join(ROOT, "pets-face-recognition_amd", "configs", "synthetic")
runs = {}
for name, flag in (("stored", "0"), ("bnfree", "1")):
    with tempfile.TemporaryDirectory() as td:
        cfg = os.path.join(td, f"equiv_{name}.py")
        open(cfg, "w").write(textwrap.dedent(f"""
            import sys, torch
            sys.path.insert(0, {common!r})
            from _common import make as _make
            _make(globals(), arch='resnet50', n_train_ids=100, n_val_ids=32, photos=8, image_size=224, train_bs=32, test_bs=64,
                  device='cuda:0', n_epochs={epochs}, n_pairs=400, compute_dtype=torch.bfloat16, seed=3, noise=1.0)
        """))
        env = dict(os.environ, PFR_BNFREE=flag)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--config", cfg], cwd=td, capture_output=True, text=True,
                           timeout=3000, env=env)
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-3000:])
            raise SystemExit(1)
        losses = [float(m.group(1)) for m in re.finditer(r"^epoch \d+ step \d+ loss ([\-0-9.eE]+)", r.stdout, flags=re.M)]
        series = {}
        for m in re.finditer(r"^(Val) (ROC AUC|Recall@K=10|Recall@K=5|Accuracy)\t([\-0-9.eE]+)$", r.stdout, flags=re.M):
            series.setdefault(m.group(2), []).append(float(m.group(3)))
        runs[name] = {"logged_losses": losses, "per_epoch": series}
a, b = runs["stored"], runs["bnfree"]
n = min(len(a["logged_losses"]), len(b["logged_losses"]))
rel = [abs(x - y) / max(abs(x), 1e-3) for x, y in zip(a["logged_losses"][:n], b["logged_losses"][:n])]
per_epoch = {k: {"stored": a["per_epoch"][k], "bnfree": b["per_epoch"][k]} for k in sorted(set(a["per_epoch"]) & set(b["per_epoch"]))}
out = {"workload": f"resnet50 + ArcFace(100 ids), synthetic 224x224, bs 32, {epochs} epochs x 25 steps, bf16, FusedSGD, seed 3, main.py --config; "
                   "PFR_BNFREE=0 (conv3 output stored, bn3 backward through it) vs 1 (BN-input-free form, statistics from the Gram matrix)",
       "loss_stored": a["logged_losses"][:n], "loss_bnfree": b["logged_losses"][:n], "max_rel_loss_diff": round(max(rel), 4),
       "validation_per_epoch": per_epoch,
       "within_tolerance": bool(max(rel[:max(1, n // 2)]) <= 0.05 and max(rel) <= 0.20)}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_train_equiv_bnfree.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
