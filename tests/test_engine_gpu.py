"""main.py --config on the HIP device: a short training run + validation epoch through Controller / Trainer."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_main_on_hip_device(tmp_path):
    cfg = tmp_path / "fe_small_hip.py"
    common = os.path.join(ROOT, "pets-face-recognition_amd", "configs", "synthetic")
    cfg.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {common!r})
        from _common import make as _make
        _make(globals(), arch='resnet18', n_train_ids=24, n_val_ids=10, photos=4, image_size=64, train_bs=16, test_bs=8,
              device='cuda:0', n_epochs=2, n_pairs=30)
    """))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--config", str(cfg)], cwd=tmp_path,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "Completed!" in r.stdout and "Val Recall@K=10" in r.stdout
    losses = [float(l.split("loss")[1]) for l in r.stdout.splitlines() if l.startswith("epoch") and "loss" in l]
    assert len(losses) >= 2 and all(l == l for l in losses)          # finite
    import torch
    runs = list((tmp_path / "results").iterdir())
    sd = torch.load(runs[0] / "checkpoints" / "epoch=1.ckpt", map_location="cpu")
    assert sd["model_loss.module.conv1.weight"].shape == (64, 3, 7, 7)
    assert int(sd["model_loss.module.bn1.num_batches_tracked"]) == 12   # 2 epochs x 6 train batches
