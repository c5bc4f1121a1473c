"""World-1 all-reduce of a 25 MB fp32 bucket through torch.distributed (nccl) and through the C-ABI communicator: microseconds per call."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from pets_face_recognition_amd._hip import comm as C
c = C.Communicator(0, 1, C.unique_id(), device=dev)
for n in (6553600, 1 << 20, 1 << 14):
    t = torch.randn(n, device=dev)
    for name, fn in (("torch", lambda: dist.all_reduce(t, op=dist.ReduceOp.AVG)), ("pfr avg", lambda: c.allreduce_(t, average=True)), ("pfr sum", lambda: c.allreduce_(t, average=False))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fn()
        b.record(); torch.cuda.synchronize()
        print(f"{n * 4 / 1e6:7.2f} MB {name:8s} {a.elapsed_time(b) / 20 * 1e3:9.1f} us per call", flush=True)
# does the call block the HOST until the stream reaches it?  20 ms of spinning kernel in front, host time of the call
import time
t = torch.randn(6553600, device=dev)
for name, fn in (("torch", lambda: dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True)), ("pfr avg", lambda: c.allreduce_(t, average=True))):
    for _ in range(2):
        torch.cuda.synchronize()
        torch.cuda._sleep(40_000_000)
        h0 = time.perf_counter(); fn(); h1 = time.perf_counter()
        torch.cuda.synchronize(); h2 = time.perf_counter()
        print(f"{name:8s} host time of the call behind a busy stream: {(h1 - h0) * 1e3:7.3f} ms (stream drained after {(h2 - h0) * 1e3:7.3f} ms)", flush=True)
