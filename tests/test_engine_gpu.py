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


_DDP_SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, {root!r})
import torch.distributed as dist
import pets_face_recognition_amd.models as M
from pets_face_recognition_amd.losses import SoftmaxBasedMetricLearning
from pets_face_recognition_amd.engine import FlatDDP

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)


def build(arch):
    torch.manual_seed(7)
    if arch == "swin_t":
        bb = getattr(M, arch)(num_classes=512, compute_dtype=torch.bfloat16)
    else:
        bb = getattr(M, arch)(compute_dtype=torch.bfloat16)
        bb.fc = torch.nn.Linear(bb.fc.in_features, 512)
    ml = SoftmaxBasedMetricLearning(bb, 100, 512, is_focal=True, arc_margin=True).to(dev).train()
    bb.hip_engine(dev)
    return ml


def grads(ml, ddp, x, y):
    for p in ml.parameters():
        p.grad = None
    ml(x, y)["loss"].backward()
    if ddp is not None:
        ddp.finish_backward()
    torch.cuda.synchronize()
    return torch.cat([p.grad.float().flatten() for p in ml.parameters() if p.grad is not None]).clone()


dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for arch, hw in (("resnet18", 64), ("swin_t", 224)):
    g = torch.Generator().manual_seed(3)
    x = torch.rand(4, 3, hw, hw, generator=g).to(dev)
    y = torch.randint(0, 100, (4,), generator=g).to(dev)
    a = grads(build(arch), None, x, y)
    ml = build(arch)
    b = grads(ml, FlatDDP(ml, bucket_mb=1), x, y)
    assert torch.isfinite(a).all() and a.abs().sum() > 0
    assert torch.equal(a, b), (arch, (a - b).abs().max().item())
    print("OK", arch, a.numel())
# gallery-sharded match over RCCL (one all-gather of the per-rank top-K lists) == unsharded match
from pets_face_recognition_amd.match import cosine_topk, cosine_topk_sharded
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(64, 512, device=dev, generator=g); gal = torch.randn(5000, 512, device=dev, generator=g)
s0, i0 = cosine_topk(q, gal, 20)
s1, i1 = cosine_topk_sharded(q, gal, 20, 0)
assert torch.equal(i0.long(), i1.long()) and torch.equal(s0, s1)
print("OK sharded match")
dist.destroy_process_group()
"""


def test_flat_ddp_world1_rccl_path_equals_single_gpu(tmp_path):
    """The RCCL path (bucketed in-place all-reduce on the communication stream, head-gradient hook, side stream joined at
    the bucket marks) with world size 1 must give bit-identical gradients to the plain single-GPU step."""
    script = tmp_path / "ddp_w1.py"
    script.write_text(_DDP_SCRIPT.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "OK resnet18" in r.stdout and "OK swin_t" in r.stdout and "OK sharded match" in r.stdout
