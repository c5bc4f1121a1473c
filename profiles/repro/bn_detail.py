"""per-geometry achieved bandwidth of the BN / elementwise launches of a ResNet-50 bs256 train step (serialised, HIP events)"""
import sys, os, types, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pets_face_recognition_amd._hip import set_tracer, EventTracer
dev = torch.device("cuda:0")
a = types.SimpleNamespace(arch="resnet50", dtype="bf16", classes=10000, batch=256)
ml, opt = bench.build(a, dev)
x = torch.rand(256, 3, 224, 224, device=dev); y = torch.randint(0, 10000, (256,), device=dev)
def step():
    opt.zero_grad(); o = ml(x, y); o["loss"].backward(); opt.step()
for _ in range(3): step()
tr = EventTracer(); torch.cuda.synchronize(); set_tracer(tr)
for _ in range(3): step()
set_tracer(None); torch.cuda.synchronize()
det = collections.defaultdict(lambda: [0, 0.0, 0.0])
for name, a_, e0, e1 in tr.records:
    if name == "pfr_bn_bwd_reduce":
        mode, rows, C = a_[7], a_[9], a_[10]; by = 2 * rows * C * 2 + (rows * C // 8 if mode == 3 else 0)
    elif name == "pfr_bn_bwd_apply":
        mode, rows, C = a_[6], a_[10], a_[11]; by = (3 + (1 if a_[8] else 0)) * rows * C * 2 + (rows * C // 8 if mode == 3 else 0)
        mode = f"{mode}{'+gres' if a_[8] else ''}"
    elif name in ("pfr_bn_act", "pfr_bn_act_mask"):
        # (x1,a1,b1,x2,a2,b2,y,[mask],dtype,rows,C,relu)
        off = 1 if name == "pfr_bn_act_mask" else 0
        rows, C = a_[8 + off], a_[9 + off]; n_in = 1 + (1 if a_[3] else 0)
        mode = f"in{n_in}"; by = (n_in + 1) * rows * C * 2 + (rows * C // 8 if off else 0)
    else:
        continue
    d = det[(name, mode, rows, C)]; d[0] += 1; d[1] += e0.elapsed_time(e1); d[2] = by
tot = collections.defaultdict(float)
for k, (n, ms, by) in sorted(det.items(), key=lambda kv: -kv[1][1]):
    per = ms / n
    print(f"{k[0]:20s} mode {str(k[1]):7s} rows {k[2]:7d} C {k[3]:5d}  x{n//3:2d}/step  {per*1e3:7.1f} us  {by/per/1e9:6.2f} TB/s")
    tot[k[0]] += ms / 3
print(dict(tot))
