"""Host-side cost of one C-ABI launch from Python: ctypes marshalling vs the HIP launch itself."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import lib, ops
x = torch.randn(8, 64, device='cuda'); out = torch.empty(1, device='cuda')
st = torch.cuda.current_stream().cuda_stream
lib.pfr_mean(x.data_ptr(), out.data_ptr(), x.numel(), st)
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    lib.pfr_mean(x.data_ptr(), out.data_ptr(), x.numel(), st)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"pfr_mean (4 args): enqueue {1e6*(t1-t0)/n:.1f} us/launch, drained after {1e6*(t2-t0)/n:.1f} us/launch")
args = (x.data_ptr(), out.data_ptr(), x.numel())
fn = lib.pfr_mean
t0 = time.perf_counter()
for _ in range(n):
    fn(*args, st)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"pre-bound args: {1e6*(t1-t0)/n:.1f} us/launch")
# a 27-argument launch (conv) on a tiny problem
xx = torch.randn(1, 8, 8, 64, device='cuda').bfloat16(); w = torch.randn(64, 1, 1, 64, device='cuda').bfloat16()
y, _ = ops.conv2d_fwd(xx, w, stride=1, pad=0, stats=False) if isinstance(ops.conv2d_fwd(xx, w, stride=1, pad=0, stats=False), tuple) else (ops.conv2d_fwd(xx, w, stride=1, pad=0, stats=False), None)
a = (xx.data_ptr(), w.data_ptr(), y.data_ptr(), 1, 1, 1, 8, 8, 64, 64, 1, 1, 1, 0, 0, 8, 8, 64, 0, 0, 0, 0, 0, 0, 0, 0)
fn = lib.pfr_conv2d_fwd
fn(*a, st); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    fn(*a, st)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"pfr_conv2d_fwd (27 args): enqueue {1e6*(t1-t0)/n:.1f} us/launch, drained {1e6*(t2-t0)/n:.1f}")
e = torch.cuda.Event()
t0 = time.perf_counter()
for _ in range(n):
    e.record()
t1 = time.perf_counter()
print(f"event record: {1e6*(t1-t0)/n:.1f} us")
t0 = time.perf_counter()
for _ in range(n):
    torch.add(x, 1.0, out=x)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"torch.add: {1e6*(t1-t0)/n:.1f} us/launch")
