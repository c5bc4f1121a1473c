"""Data-parallel gradient exchange for the FE train step: one process per GPU, RCCL all-reduce over xGMI.

Reference behaviour (/root/reference/utils/__init__.py:114-119 → PL DDPPlugin(find_unused_parameters=False,
gradient_as_bucket_view=True); engine/trainer.py:105 sync_batchnorm=False): the whole model + ArcFace head is
replicated, every rank keeps private BN statistics, gradients are averaged across ranks in ~25 MB buckets
overlapped with backward.

MI355X design: gradients already live in ONE flat fp32 buffer in forward-parameter order (FEEngine), and backward
finishes it from the end towards the front.  `BucketReducer.ready(off)` is called by the engine each time the suffix
[off, end) became final; full buckets of that suffix are all-reduced (AVG) in place on a dedicated communication
stream while the remaining dgrad/wgrad kernels run.  No gradient copies, no per-parameter hooks, 5 large
collectives per step for ResNet-50 (xGMI is point-to-point: few, large messages).

ONE collective implementation at a time: by default the RCCL communicator torch.distributed ("nccl") owns; with
`collective="pfr"` (or PFR_DDP_COLLECTIVE=pfr) the all-reduces go through the C-ABI communicator of csrc/pfr_comm.hip
(`pfr_comm_allreduce`, RCCL resolved by dlopen) — what a non-Python host of libpfr_hip.so drives; torch.distributed is then only
the rendezvous that hands the 128-byte unique id around.  Both are exercised by the same tests.
"""
import os

import torch
import torch.distributed as dist


class BucketReducer:
    def __init__(self, flat, bucket_elems=6 * 1024 * 1024, group=None, average=True, comm=None):
        self.comm = comm     # a _hip.comm.Communicator: all-reduces through the C-ABI (pfr_comm_allreduce) instead of torch.distributed
        self.flat = flat
        self.n = flat.numel()
        self.bucket = int(bucket_elems)
        self.group = group
        self.average = average
        self.hi = self.n
        # once at most `tail` elements remain below a mark, everything final goes out at that mark: the LAST collective of a step (launched at
        # mark 0, behind the stem's weight gradient — nothing is left to overlap it with) then carries only the stem's parameters instead of up to
        # a whole bucket (ResNet-50: 0.6 M elements = 2.4 MB instead of 4.9 M = 19.6 MB).  Only when at least `tail` elements are waiting:
        # the closely spaced marks of the first blocks do not each become a collective (one more collective per step)
        self.tail = self.bucket // 8
        self.handles = []
        self.launched = []   # (lo, hi) ranges, for tests / introspection
        self.cuda = flat.is_cuda
        self.comm_stream = torch.cuda.Stream(device=flat.device) if self.cuda else None
        self.side_stream_fn = None   # callable -> the engine's side (weight-gradient) stream, or None

    def reset(self):
        self.hi = self.n
        self.handles = []
        self.launched = []

    def reduce_tensor(self, view, lo=0, hi=0):
        """asynchronous mean all-reduce of `view` in place (RCCL: AVG on the communication stream, ordered after everything
        the compute stream has enqueued so far; gloo/CPU: SUM now, the division in finish())"""
        op = dist.ReduceOp.AVG if (self.average and self.cuda) else dist.ReduceOp.SUM
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            side = self.side_stream_fn() if self.side_stream_fn is not None else None
            if side is not None:
                # the weight gradients of this range were enqueued on the engine's side stream: the COMMUNICATION stream waits for
                # them, the main stream (the dgrad -> BN-backward chain) does not stall at the bucket boundary
                self.comm_stream.wait_stream(side)
            if self.comm is not None:
                # stream-ordered on the communication stream: no handle to wait for, finish() joins the stream
                self.comm.allreduce_(view, average=self.average, stream=self.comm_stream)
                h = None
            else:
                with torch.cuda.stream(self.comm_stream):
                    h = dist.all_reduce(view, op=op, group=self.group, async_op=True)
        else:
            h = dist.all_reduce(view, op=op, group=self.group, async_op=True)
        self.handles.append((h, view if (lo == 0 and hi == 0) else None, lo, hi))

    def _reduce(self, lo, hi):
        self.reduce_tensor(self.flat[lo:hi], lo, hi)
        self.launched.append((lo, hi))

    def ready(self, off):
        """flat[off:] is final.  Launch every full bucket of the not-yet-reduced part (everything when off == 0)."""
        while self.hi - off >= self.bucket:
            self._reduce(self.hi - self.bucket, self.hi)
            self.hi -= self.bucket
        if 0 < off <= self.tail and self.hi - off >= self.tail:
            self._reduce(off, self.hi)
            self.hi = off
        if off == 0 and self.hi > 0:
            self._reduce(0, self.hi)
            self.hi = 0

    def finish(self):
        """Make the compute stream wait for all outstanding reductions."""
        if self.hi > 0:
            self.ready(0)
        for h, t, lo, hi in self.handles:
            if h is None:
                continue
            if self.cuda:
                with torch.cuda.stream(self.comm_stream):
                    h.wait()
            else:
                h.wait()
                if self.average:
                    (t if t is not None else self.flat[lo:hi]).div_(dist.get_world_size(self.group))
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.reset()


class FlatDDP:
    """Wraps a `SoftmaxBasedMetricLearning` (HIP backbone + margin head) for data-parallel training."""

    def __init__(self, model_loss, bucket_mb=25, group=None, collective=None):
        self.model_loss = model_loss
        self.group = group
        eng = model_loss.module.hip_engine()
        self.eng = eng
        self.extra = [p for n, p in model_loss.named_parameters() if not n.startswith("module.")]
        collective = collective or os.environ.get("PFR_DDP_COLLECTIVE", "torch")
        if collective not in ("torch", "pfr"):
            raise ValueError(f"collective must be 'torch' or 'pfr', got {collective!r}")
        self.comm = None
        if collective == "pfr":
            from .._hip import comm as C
            rank, world = dist.get_rank(group), dist.get_world_size(group)
            uid = torch.zeros(C.UNIQUE_ID_BYTES, dtype=torch.uint8, device=eng.grad.device)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(C.unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, 0, group=group)          # rendezvous only
            self.comm = C.Communicator(rank, world, bytes(uid.cpu().tolist()), device=eng.grad.device)
        # replicate rank 0's parameters / BN buffers
        self.broadcast_parameters()
        self.reducer = BucketReducer(eng.grad, bucket_elems=bucket_mb * 1024 * 1024 // 4, group=group, comm=self.comm)
        eng.grad_ready_hook = self.reducer.ready
        if self.reducer.cuda:
            self.reducer.side_stream_fn = lambda: getattr(eng, "side", None)
            eng.hook_syncs_side = True
        # The head's gradient (ArcFace weight, 20 MB at 10 000 ids) is complete BEFORE the backbone's backward starts:
        # its all-reduce is launched from a post-accumulate hook and overlaps the whole backbone backward.
        self._extra_done = set()
        self._hooks = [p.register_post_accumulate_grad_hook(self._reduce_param) for p in self.extra if p.requires_grad]

    def broadcast_parameters(self):
        """rank 0's parameters / BN statistics / head weights → every rank (construction, and after a checkpoint was loaded)"""
        ts = [self.eng.master] + ([self.eng.stats] if hasattr(self.eng, "stats") else []) + [p.data for p in self.extra]
        self._bcast(ts)

    def _bcast(self, tensors):
        if self.comm is None:
            for t in tensors:
                dist.broadcast(t, 0, group=self.group)
            return
        # through the C-ABI communicator: the other ranks contribute zeros to a SUM all-reduce (x + 0 + ... + 0 = x exactly).  The
        # exchange runs on a TEMPORARY buffer and is copied back only after it completed: a failed or timed-out collective
        # must not leave a rank with zeroed parameters.  The two communicators (this one and torch.distributed's) are never in flight
        # together: the device is drained before the first and after the last pfr collective, so a torch collective issued next —
        # e.g. sync_buffers' broadcast of num_batches_tracked — is ordered after them on every rank.
        dev = self.eng.grad.device
        torch.cuda.synchronize(dev)
        staged = []
        for t in tensors:
            if t.dtype not in (torch.float32, torch.bfloat16):
                dist.broadcast(t, 0, group=self.group)
                torch.cuda.synchronize(dev)
                continue
            tmp = t.detach().clone().contiguous() if self.comm.rank == 0 else torch.zeros(t.shape, dtype=t.dtype, device=t.device)
            self.comm.allreduce_(tmp, average=False)
            staged.append((t, tmp))
        torch.cuda.synchronize(dev)          # raises here if a collective failed: nothing has been overwritten yet
        for t, tmp in staged:
            t.copy_(tmp)
        torch.cuda.synchronize(dev)

    def detach(self):
        """unhook from the model (the trainer builds a new FlatDDP when the model got a new engine)"""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self.eng.grad_ready_hook == self.reducer.ready:
            self.eng.grad_ready_hook = None
            self.eng.hook_syncs_side = False

    def check(self):
        """the reducer must be bound to the engine that is actually executing the model"""
        cur = self.model_loss.module.hip_engine()
        if cur is not self.eng or cur.grad_ready_hook != self.reducer.ready or cur.grad.data_ptr() != self.reducer.flat.data_ptr():
            raise RuntimeError("FlatDDP is bound to a stale engine (the model was moved / re-adopted): gradients of this rank "
                               "would not be all-reduced — rebuild FlatDDP (Trainer._setup does)")

    def _reduce_param(self, p):
        self.reducer.reduce_tensor(p.grad)
        self._extra_done.add(id(p))

    def reduce_extra(self):
        """all-reduce gradients outside the engine's flat buffer whose hook did not fire (e.g. gradient accumulation paths)"""
        for p in self.extra:
            if p.grad is not None and id(p) not in self._extra_done:
                self._reduce_param(p)
        self._extra_done.clear()

    def finish_backward(self):
        self.check()
        self.reduce_extra()
        self.reducer.finish()

    def sync_buffers(self):
        """Rank 0's BatchNorm running statistics → every rank.  torch's DistributedDataParallel (the reference's wrapper,
        utils/__init__.py:114-119) broadcasts module buffers from rank 0 (`broadcast_buffers=True`), so every rank evaluates
        with rank 0's statistics; here the statistics of a step never feed the next train step (train-mode BN uses batch
        statistics), so one broadcast before an evaluation pass gives the same evaluation-time state."""
        if hasattr(self.eng, "stats"):
            self._bcast([self.eng.stats])
            dist.broadcast(self.eng.nbt, 0, group=self.group)


class GenericDDP:
    """Gradient averaging for the torch (CPU) execution path: the same replicate-all / mean-reduce semantics on
    ordinary parameter tensors (used by the CPU plumbing configs and the gloo tests)."""

    def __init__(self, module, group=None):
        self.module = module      # identity of what this reducer is bound to (Trainer._setup rebinds when the model changes)
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.group = group
        self.buffers = [b for b in module.buffers() if b.dtype.is_floating_point]
        self.broadcast_parameters()

    def broadcast_parameters(self):
        for p in self.params:
            dist.broadcast(p.data, 0, group=self.group)
        for b in self.buffers:
            dist.broadcast(b.data, 0, group=self.group)

    def finish_backward(self):
        world = dist.get_world_size(self.group)
        grads = [p.grad for p in self.params if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, group=self.group)
        flat.div_(world)
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n

    def sync_buffers(self):
        for b in self.buffers:
            dist.broadcast(b.data, 0, group=self.group)
