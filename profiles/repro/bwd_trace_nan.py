"""Runs the backward launch list of a bs-256 ResNet-50 step op by op on one stream and reports the first launch whose OUTPUT
tensor holds a non-finite value (all buffers of the plan are poisoned... no: checked after every conv / BN launch)."""
import sys, os
os.environ["PFR_C_PLAN"] = "0"
os.environ["PFR_SIDE_STREAM"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pets_face_recognition_amd.models as M
from pets_face_recognition_amd._hip import lib, dtype_id
dev = "cuda:0"
torch.manual_seed(0)
m = M.resnet50(compute_dtype=torch.bfloat16)
m.fc = torch.nn.Linear(2048, 512)
m = m.to(dev).train()
g = torch.Generator().manual_seed(1)
x = torch.rand(256, 3, 224, 224, generator=g).to(dev)
dy = torch.randn(256, 512, generator=g).to(dev)
eng = m.hip_engine(dev)
out = m(x)
plan = eng._last_plan
bufs = {}
for v in plan.bufs.values():
    for t in (v if isinstance(v, tuple) else (v,)):
        bufs[t.data_ptr()] = t
stream = torch.cuda.current_stream().cuda_stream
lib.pfr_cast(dy.data_ptr(), dtype_id(dy.dtype), plan.meta["demb"].data_ptr(), eng.did, dy.numel(), stream)
names = {id(getattr(lib, n)): n for n in ("pfr_conv2d_fwd", "pfr_conv2d_dgrad_join", "pfr_bn_bwd_apply", "pfr_bn_bwd_reduce", "pfr_maxpool_bwd", "pfr_avgpool_bwd")}
n = 0
for fn, args in plan.meta["bwd0"]:
    if fn is None:
        continue
    if fn.__class__ is int:
        if fn == 1:
            args[0](*args[1], stream)
        continue
    fn(*args, stream)
    nm = names.get(id(fn))
    if nm in ("pfr_conv2d_fwd", "pfr_conv2d_dgrad_join"):
        torch.cuda.synchronize()
        o = bufs.get(args[2])
        n += 1
        if o is not None and not torch.isfinite(o.float()).all():
            bad = (~torch.isfinite(o.float())).reshape(-1, o.shape[-1])
            rows = bad.any(1).nonzero().flatten()
            print("first non-finite output: launch", n, nm, "args", args[3:18], "shape", tuple(o.shape), "bad rows", rows.numel(), rows[:10].tolist(), "bad cols of first", bad[rows[0]].nonzero().flatten()[:10].tolist())
            xin = bufs.get(args[0])
            print(" input finite:", None if xin is None else bool(torch.isfinite(xin.float()).all()))
            break
else:
    print("no non-finite conv output in", n, "conv launches")
