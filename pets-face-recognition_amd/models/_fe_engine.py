"""FEEngine — executes a ResNet feature extractor (forward, backward) on the gfx950 kernels of libpfr_hip.so.

This is the MI355X counterpart of what, in the reference, is `self.module(img)` + autograd through torchvision's
resnet (/root/reference/losses/__init__.py:38-41 called from engine/controller.py:27-29).  Design:

  * parameters live in ONE flat fp32 master buffer (conv weights stored [Cout][R][S][Cin] = NHWC-friendly; the
    nn.Parameters the reference API exposes are strided views of it, so state_dict()/optimizers/DDP see ordinary
    tensors with torchvision names) and gradients in ONE flat fp32 buffer of the same layout → one fused optimizer
    launch per param group, one RCCL all-reduce per bucket range;
  * activations are NHWC in the compute dtype (bf16, or f32 for the parity path); every buffer of a step is
    allocated once per input shape and the whole step is a pre-built list of C-ABI calls (a "plan") with fixed
    device pointers — no allocator traffic, no autograd graph, trivially capturable in a hipGraph;
  * batch statistics come out of the producing conv's epilogue; z = relu(BN(c)) is materialised by ONE elementwise pass
    (pfr_bn_act; the fused consumer-prologue form exists but is off — it costs ~2x kernel time); the bottleneck
    tail (BN + projection-BN + residual add + ReLU) is one pass that also writes the ReLU sign as a bit mask, which the two
    backward kernels of the block's last BN and the residual join in conv1's data-gradient epilogue read; inner BNs recompute
    their ReLU mask from scale*x+shift;
  * a step's launch lists are replayed from C (csrc/pfr_plan.hip) unless PFR_C_PLAN=0 or a launch tracer is installed;
    two plan slots per input shape allow list inputs (two forwards before one backward), each with its own BN coefficients.
"""
import os
import weakref

import torch
import torch.nn as nn

from .._hip import lib, dtype_id, PfrError
from .._hip.lib import _TRACER
from .._hip.ops import conv_out_hw
from .._hip.cplan import CPlan

_ALIGN = 64  # elements; keeps every parameter 16-byte aligned in both fp32 and bf16 shadows


def default_compute_dtype():
    v = os.environ.get("PFR_COMPUTE_DTYPE", "bf16").lower()
    return torch.float32 if v in ("f32", "fp32", "float32") else torch.bfloat16


def flat_layout(model):
    """name -> offset (in elements) of every parameter inside the flat master / gradient buffers, and their length: forward
    (registration) order, each parameter padded to a multiple of _ALIGN elements"""
    offs, total = {}, 0
    for name, p in model.named_parameters():
        offs[name] = total
        total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
    return offs, total


def grad_ready_marks(model):
    """The offsets `off` at which the backward pass reports "flat gradient [off, end) is final" (FEEngine._mark), in the order
    it reports them: after fc, after every residual block (last block first; a block's first parameter is conv1.weight),
    and 0 at the end.  Pure host logic — what engine/ddp.py's bucket reducer is driven by."""
    offs, _ = flat_layout(model)
    marks = [offs["fc.weight"]]
    for lname in ("layer4", "layer3", "layer2", "layer1"):
        layer = getattr(model, lname)
        for bi in reversed(range(len(layer))):
            marks.append(offs[f"{lname}.{bi}.conv1.weight"])
    marks.append(0)
    return marks


class _Conv:
    pass


class _BN:
    pass


# op codes of the backward list besides plain (ctypes function, args) launches on the main stream
_SIDE, _FORK, _SREC, _WAIT, _MWAIT = 1, 2, 3, 4, 5   # _MWAIT: a wait that only a grad-ready hook (DDP bucket) needs


def _side_with_ddp():
    """main + side + communication stream + RCCL's internal stream need more than HIP's default 4 hardware queues: with
    4, two of them share a queue and serialise (measured 9.05 k vs 9.70 k img/s); the package asks for 8 at import."""
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= 8
    except ValueError:
        return False


class PlanTicket:
    """Held by the autograd node of one forward pass: while it is alive (and its backward has not run) the plan that
    produced the saved activations is not handed to another forward pass."""
    __slots__ = ("__weakref__",)


class _Plan:
    __slots__ = ("ops", "bufs", "meta")

    def __init__(self):
        self.ops = []
        self.bufs = {}
        self.meta = {}

    def run(self, stream, hook=None):
        for fn, args in self.ops:
            if fn is None:
                if hook is not None:
                    hook(*args)
            else:
                fn(*args, stream)


class FEEngine:
    def __init__(self, model, device, compute_dtype=None):
        if not str(device).startswith("cuda"):
            raise PfrError("FEEngine runs on the HIP device only (no CPU fallback)")
        lib.pfr_version()  # fail loudly if the shared library is missing
        self.device = torch.device(device)
        self.dtype = compute_dtype or default_compute_dtype()
        self.did = dtype_id(self.dtype)
        self.kp = 8 if self.dtype == torch.bfloat16 else 4
        self.model_id = id(model)
        self.fc_id = id(model.fc)
        self.plans = {}
        self.side = None          # side stream of the weight-gradient launches (see build_plan)
        self.side_events = []
        self.fold_eval = True     # inference: BN folded into the convs (36 k -> 56 k img/s; the unfolded eval plan stays for tests)
        self.fold_w = None
        self.fold_cache = True    # inference: re-fold only when the parameters changed
        self.fold_state = None
        self.graph_eval = os.environ.get("PFR_GRAPH_EVAL", "0") == "1"   # opt-in: inference plans replayed as hipGraphs
        self.wt_fork = self.wt_ready = None
        self.wt_pending = False
        self.side_stream_enabled = os.environ.get("PFR_SIDE_STREAM", "1") != "0"
        # (BN-apply + ReLU fused into the CONSUMER conv's operand prologue cost ~2x the kernel time it saved — the transform is repeated
        #  per tap and per Cout tile — and was retired in round 4: profiles/HISTORY.md; z = relu(BN(c)) is materialised once by pfr_bn_act)
        # replay the step's launch lists from C (csrc/pfr_plan.hip) instead of a Python loop: ~16 ms -> ~2 ms of host time per
        # ResNet-50 step (PFR_C_PLAN=0 keeps the interpreter loop; a launch tracer always uses it)
        self.c_plan = os.environ.get("PFR_C_PLAN", "1") != "0"
        # Opt-in (PFR_FUSE_BNB=1): BatchNorm-backward sums (Σ g·mask, Σ g·mask·x̂) out of the epilogue of the data-gradient launch
        # that produces g (pfr_conv2d_dgrad_bn) instead of a separate pass over g and x (pfr_bn_bwd_reduce).  It removes one
        # read of g per BN layer (-5.7 GB/step) and 45 launches, but MEASURED SLOWER (25.2 vs 23.2 ms/step): the data-gradient
        # kernels are bound by their vector-memory path into LDS, not by HBM, so the extra x reads and the registers of the sums
        # lengthen them (3.5 -> 7.4 ms) by more than the HBM-speed reduce kernels (2.1 ms at 4.5 TB/s) cost.  DESIGN.md §6.
        # PFR_FUSE_BNB=2 (round 3): the same fusion through the STREAMING kernels only (pfr_sconv.hip EP 4 / 5: their epilogue has
        # the slack the tile kernel's lacks) — 1x1 data gradients with one consuming BN; everything else keeps the separate pass.
        # Measured (profiles/r03_bnb_streaming.txt): 19.64 -> 19.35 ms/step; default.  0 = separate pass everywhere.
        self.fuse_bnb = int(os.environ.get("PFR_FUSE_BNB", "2") or 0)
        lib.pfr_set_tuning(b"bnb", self.fuse_bnb)
        self._tuning_epoch = lib.pfr_tuning_epoch()
        self._bnb_min_rows = 0
        # (retired in round 4 after losing their A/B, evidence under profiles/: reduce + finalize in one launch — r03_finalize_fusion.txt;
        #  the stem's max-pool gradient gathered inside the BN-backward passes — neutral; a CU-masked side stream — r03_cumask_sweep.txt;
        #  weight gradients handed to the side stream in groups — r03_wgrad_batch.txt)
        # BN-input-free backward of a bottleneck's conv3 + bn3 (csrc/pfr_bnfree.hip; round 4): the gradient that reaches the block
        # output is stored THROUGH the block's ReLU mask by its producer, bn3's backward sums come out of the weight-gradient GEMM, and
        # conv3's data gradient is G·(A∘W) + z2·S + bias — bn3's input and its gradient are never read / written in the backward pass,
        # so the forward pass does not store conv3's output either (statistics pass + recompute with the block tail in the epilogue).
        # bf16 streaming geometries only (layer1-2 of ResNet-50 at bs 256); PFR_BNFREE=0 keeps the materialised form everywhere.
        self.bnfree = os.environ.get("PFR_BNFREE", "1") != "0" and self.dtype == torch.bfloat16 and self.fuse_bnb == 2
        # bn3's batch statistics of those blocks from the Gram matrix of conv3's input (pfr_bn_stats_from_gram) instead of a statistics
        # pass over conv3: the statistics of the EXACT convolution (~1e-6 from those of its bf16-rounded values), -0.3 ms/step;
        # PFR_BNFREE_GRAMSTATS=0 keeps the statistics pass, whose forward is bit-identical to the stored form
        self.gram_stats = os.environ.get("PFR_BNFREE_GRAMSTATS", "1") != "0"
        # ... unless a channel's variance comes out of a cancellation the fp32 Gram matrix cannot resolve (|mean| >> std): the kernel bounds
        # its rounding error per channel and reports through this host-pinned word; the engine then goes back to the statistics pass
        # (checked without a synchronisation at the start of every forward pass, so the fallback takes effect a step or two later)
        self.gram_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.ws_main = None             # split-K workspace of weight-gradient launches on the MAIN stream (self.ws belongs to the side stream)
        self.grad_ready_hook = None     # callable(lo, hi): flat-grad range [lo, hi) is final (DDP bucket hook)
        self.hook_syncs_side = False    # True: the hook makes ITS stream wait for self.side (the main stream then never waits at a mark)
        self.bucket_elems = 6 * 1024 * 1024
        self._adopt(model)

    # ------------------------------------------------------------------------------------------ parameters
    def _adopt(self, model):
        dev = self.device
        named = list(model.named_parameters())
        offs, total = flat_layout(model)
        self.n_flat = total
        self.master = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.shadow = self.master if self.dtype == torch.float32 else torch.zeros(total, dtype=self.dtype, device=dev)
        self.offs = offs
        self.param_list = []
        self._views = {}
        for name, p in named:
            o, n = offs[name], p.numel()
            if p.dim() == 4:
                O, I, R, S = p.shape
                mv = self.master[o:o + n].view(O, R, S, I)
                mv.copy_(p.data.detach().to(dev).permute(0, 2, 3, 1))
                pv = mv.permute(0, 3, 1, 2)
                gv = self.grad[o:o + n].view(O, R, S, I).permute(0, 3, 1, 2)
            else:
                mv = self.master[o:o + n].view(p.shape)
                mv.copy_(p.data.detach().to(dev))
                pv = mv
                gv = self.grad[o:o + n].view(p.shape)
            p.data = pv
            p.grad = None
            self._views[name] = (p, gv)
            self.param_list.append(p)
        self.first_param = named[0][1]
        # BN buffers → flat
        bns = [(n, m) for n, m in model.named_modules() if isinstance(m, nn.BatchNorm2d)]
        nstat = sum(m.num_features for _, m in bns)
        self.stats = torch.zeros(2 * nstat, dtype=torch.float32, device=dev)
        self.nbt = torch.zeros(len(bns), dtype=torch.int64, device=dev)
        so = 0
        self._bn_of = {}
        for i, (n, m) in enumerate(bns):
            C = m.num_features
            rm = self.stats[so:so + C]
            rv = self.stats[nstat + so:nstat + so + C]
            rm.copy_(m.running_mean.detach().to(dev))
            rv.copy_(m.running_var.detach().to(dev))
            self.nbt[i] = int(m.num_batches_tracked.item())
            m.running_mean = rm
            m.running_var = rv
            m.num_batches_tracked = self.nbt[i]
            so += C
            b = _BN()
            b.C, b.eps, b.momentum = C, m.eps, (m.momentum if m.momentum is not None else 0.1)
            b.gamma = self.master[offs[n + ".weight"]:offs[n + ".weight"] + C]
            b.beta = self.master[offs[n + ".bias"]:offs[n + ".bias"] + C]
            b.dgamma = self.grad[offs[n + ".weight"]:offs[n + ".weight"] + C]
            b.dbeta = self.grad[offs[n + ".bias"]:offs[n + ".bias"] + C]
            b.g_off = offs[n + ".weight"]
            b.rm, b.rv = rm, rv
            b.coef = torch.zeros((4, C), dtype=torch.float32, device=dev)   # mean, invstd, scale, shift
            b.bcoef = torch.zeros((3, C), dtype=torch.float32, device=dev)  # backward coefficients
            self._bn_of[id(m)] = b
        self.bn_list = [self._bn_of[id(m)] for _, m in bns]
        maxc = max(m.num_features for _, m in bns)
        self.bn_ws = torch.empty(max(1, lib.pfr_bn_finalize_ws_floats(1 << 20, maxc)), dtype=torch.float32, device=dev)

        # conv / fc records
        def conv_rec(name, m):
            c = _Conv()
            c.name = name
            c.Cout, c.Cin, c.R, c.S = m.weight.shape
            c.stride, c.pad = m.stride[0], m.padding[0]
            c.off = offs[name + ".weight"]
            n = m.weight.numel()
            c.w = self.shadow[c.off:c.off + n].view(c.Cout, c.R, c.S, c.Cin)
            c.g = self.grad[c.off:c.off + n].view(c.Cout, c.R, c.S, c.Cin)
            c.wt = torch.zeros((c.Cin, c.R, c.S, c.Cout), dtype=self.dtype, device=dev)
            c.need_wt = True
            return c

        self._conv_of = {}
        for n, m in model.named_modules():
            if isinstance(m, nn.Conv2d):
                if m.bias is not None or m.groups != 1 or m.dilation[0] != 1:
                    raise PfrError(f"{n}: only bias-free, dense, undilated convolutions are supported")
                self._conv_of[id(m)] = conv_rec(n, m)
        # stem: channel-padded weight copy (3 → kp channels so that a channel chunk is 16 bytes)
        st = self._conv_of[id(model.conv1)]
        st.need_wt = False
        self.cp = (st.Cin + self.kp - 1) // self.kp * self.kp
        st.w_pad = torch.zeros((st.Cout, st.R, st.S, self.cp), dtype=self.dtype, device=dev)
        st.g_pad = torch.zeros((st.Cout, st.R, st.S, self.cp), dtype=torch.float32, device=dev)
        # space-to-depth form of a 7x7 / stride-2 / pad-3 stem (pfr_s2d_*): 4x4 stride-1 pad-2 conv over [H/2][W/2][cs2d]
        self.s2d = None
        # Used on the bf16 (throughput) path.  The fp32 (parity) path keeps the plain 7x7 form: the space-to-depth sums are
        # just as exact (tests/test_kernels_gpu.py::test_stem_space_to_depth_exact), but their different rounding order moves
        # a handful of ReLU / max-pool near-ties, which the end-to-end fp32 gradient and loss-trace tests are sensitive to.
        if (st.R, st.S, st.stride, st.pad) == (7, 7, 2, 3) and self.dtype == torch.bfloat16:
            q = _Conv()
            q.name = st.name + "(s2d)"
            q.Cout, q.Cin, q.R, q.S, q.stride, q.pad = st.Cout, (4 * st.Cin + self.kp - 1) // self.kp * self.kp, 4, 4, 1, 2
            q.w = torch.zeros((q.Cout, 4, 4, q.Cin), dtype=self.dtype, device=dev)
            q.g = torch.zeros((q.Cout, 4, 4, q.Cin), dtype=torch.float32, device=dev)
            q.off, q.need_wt = st.off, False
            self.s2d = q
        self.stem = (st, self._bn_of[id(model.bn1)])
        # blocks
        self.blocks = []
        for layer in (model.layer1, model.layer2, model.layer3, model.layer4):
            for blk in layer:
                convs = []
                i = 1
                while hasattr(blk, f"conv{i}"):
                    convs.append((self._conv_of[id(getattr(blk, f"conv{i}"))], self._bn_of[id(getattr(blk, f"bn{i}"))]))
                    i += 1
                down = None
                if blk.downsample is not None:
                    down = (self._conv_of[id(blk.downsample[0])], self._bn_of[id(blk.downsample[1])])
                self.blocks.append((convs, down))
        # fc
        fc = model.fc
        if not isinstance(fc, nn.Linear):
            raise PfrError("model.fc must be an nn.Linear")
        f = _Conv()
        f.name = "fc"
        f.Cout, f.Cin, f.R, f.S, f.stride, f.pad = fc.out_features, fc.in_features, 1, 1, 1, 0
        f.off = offs["fc.weight"]
        n = fc.weight.numel()
        f.w = self.shadow[f.off:f.off + n].view(f.Cout, 1, 1, f.Cin)
        f.g = self.grad[f.off:f.off + n].view(f.Cout, 1, 1, f.Cin)
        f.wt = torch.zeros((f.Cin, 1, 1, f.Cout), dtype=self.dtype, device=dev)
        f.need_wt = True
        if fc.bias is not None:
            bo = offs["fc.bias"]
            f.bias = self.master[bo:bo + f.Cout]
            f.dbias = self.grad[bo:bo + f.Cout]
        else:
            f.bias = f.dbias = None
        self.fc = f
        self.emb_dim = f.Cout
        self.all_convs = [c for c in self._conv_of.values()] + [f]
        # wgrad workspace (max over layers is found at plan-build time)
        self.ws = None
        torch.cuda.synchronize(dev)

    def matches(self, model):
        return (id(model) == self.model_id and id(model.fc) == self.fc_id
                and self.first_param.data.data_ptr() == self.master.data_ptr())

    def attach_grads(self):
        """Point every parameter's .grad at its slice of the flat gradient buffer."""
        for p, gv in self._views.values():
            p.grad = gv

    def flat_ranges(self, params):
        """Contiguous [lo, hi) ranges of the flat buffers covered by `params` (for fused optimizers / buckets)."""
        base = self.master.data_ptr()
        spans = []
        for p in params:
            lo = (p.data.data_ptr() - base) // 4
            if lo < 0 or lo >= self.n_flat:
                raise PfrError("parameter does not belong to this engine")
            hi = lo + (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            spans.append((lo, hi))
        spans.sort()
        out = []
        for lo, hi in spans:
            if out and out[-1][1] == lo:
                out[-1] = (out[-1][0], hi)
            else:
                out.append((lo, hi))
        return out

    def _side_ok(self):
        """Side stream off: PFR_SIDE_STREAM=0; a launch tracer is active (it brackets launches with events on ONE stream);
        or gradients are being all-reduced (DDP) while fewer than 8 hardware queues are available (see _side_with_ddp)."""
        return self.side_stream_enabled and _TRACER[0] is None and (self.grad_ready_hook is None or _side_with_ddp())

    # ------------------------------------------------------------------------------------------ weights
    def refresh_weights(self, stream, for_backward=True, cast=True):
        """master fp32 → compute-dtype shadow, channel-padded stem weights, data-gradient weight layouts."""
        if cast and self.dtype != torch.float32:
            lib.pfr_cast(self.master.data_ptr(), 0, self.shadow.data_ptr(), self.did, self.n_flat, stream)
        st = self.stem[0]
        lib.pfr_nchw_to_nhwc(self.master.data_ptr() + 4 * st.off, st.w_pad.data_ptr(), self.did, st.Cout * st.R * st.S,
                             st.Cin, 1, 1, self.cp, stream)
        if self.s2d is not None:
            lib.pfr_s2d_weight(self.master.data_ptr() + 4 * st.off, self.s2d.w.data_ptr(), self.did, st.Cout, st.Cin, self.s2d.Cin,
                               stream)
        if for_backward:
            # the flipped / transposed copies are first needed by the backward pass: build them on the side stream,
            # concurrent with the forward pass (backward() waits for wt_ready)
            sptr = stream
            use_side = self._side_ok()
            if use_side:
                if self.side is None:
                    self.side = torch.cuda.Stream(device=self.device)
                if self.wt_fork is None:
                    self.wt_fork, self.wt_ready = torch.cuda.Event(), torch.cuda.Event()
                self.wt_fork.record(torch.cuda.current_stream())
                self.side.wait_event(self.wt_fork)
                sptr = self.side.cuda_stream
            # one launch for every conv's data-gradient weights (descriptor table built once: the pointers are fixed)
            tab = getattr(self, "_wt_table", None)
            if tab is None:
                import struct
                convs = [c for c in self.all_convs if c.need_wt]
                raw = b"".join(struct.pack("<QQiiii", c.w.data_ptr(), c.wt.data_ptr(), c.Cout, c.R, c.S, c.Cin) for c in convs)
                tab = self._wt_table = (torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device), len(convs))
            lib.pfr_weight_dgrad_layout_batch(tab[0].data_ptr(), tab[1], self.did, sptr)
            self.wt_pending = use_side
            if use_side:
                self.wt_ready.record(self.side)

    def _use_s2d(self, H, W):
        return self.s2d is not None and H % 2 == 0 and W % 2 == 0

    def _input_layout(self, plan, x, stream):
        """fp32 NCHW batch of the reference's dataloaders → the stem's operand layout (compute dtype)"""
        N, C, H, W = x.shape
        if plan.meta.get("s2d"):
            lib.pfr_s2d_input(x.data_ptr(), plan.meta["x_nhwc"].data_ptr(), self.did, N, C, H, W, self.s2d.Cin, stream)
        else:
            lib.pfr_nchw_to_nhwc(x.data_ptr(), plan.meta["x_nhwc"].data_ptr(), self.did, N, C, H, W, self.cp, stream)

    # ------------------------------------------------------------------------------------------ inference (BN folded)
    def _fold_setup(self):
        """Folded-weight buffers + the descriptor table of pfr_fold_bn (one record per conv/BN pair)."""
        import struct
        pairs = [self.stem] + [cb for convs, down in self.blocks for cb in convs] + [down for _, down in self.blocks if down]
        if self.s2d is not None:
            pairs.append((self.s2d, self.stem[1]))
        nw = sum((c.w_pad.numel() if c is self.stem[0] else c.w.numel()) for c, _ in pairs)
        nb = sum(c.Cout for c, _ in pairs)
        self.fold_w = torch.empty(nw + _ALIGN * len(pairs), dtype=self.dtype, device=self.device)
        self.fold_b = torch.empty(nb, dtype=torch.float32, device=self.device)
        rec = b""
        wo = bo = 0
        es = self.fold_w.element_size()
        for c, bn in pairs:
            if c is self.stem[0]:
                src, f32, K = c.w_pad.data_ptr(), 0, c.R * c.S * self.cp   # padded compute-dtype copy (refresh_weights)
            elif c is self.s2d:
                src, f32, K = c.w.data_ptr(), 0, c.R * c.S * c.Cin         # space-to-depth copy (refresh_weights)
            else:
                src, f32, K = self.master.data_ptr() + 4 * c.off, 1, c.R * c.S * c.Cin
            c.wf = self.fold_w[wo:wo + c.Cout * K].view(c.Cout, c.R, c.S, K // (c.R * c.S))
            c.bf = self.fold_b[bo:bo + c.Cout]
            rec += struct.pack("<7q2qfi", src, bn.gamma.data_ptr(), bn.beta.data_ptr(), bn.rm.data_ptr(), bn.rv.data_ptr(),
                               c.wf.data_ptr(), c.bf.data_ptr(), c.Cout, K, float(bn.eps), f32)
            wo += (c.Cout * K + _ALIGN - 1) // _ALIGN * _ALIGN
            bo += c.Cout
        assert len(rec) == 80 * len(pairs)
        self.fold_desc = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(self.device)
        self.fold_n = len(pairs)
        self.ident = torch.cat([torch.ones(self.stem[0].Cout), torch.zeros(self.stem[0].Cout)]).to(self.device)

    def build_eval_plan(self, N, H, W):
        """Embedding extraction (no gradient): every eval-mode BN is folded into its conv (pfr_fold_bn, one launch per
        forward), so a bottleneck is three conv launches: bias + ReLU, and for the last one + shortcut, in the epilogue."""
        if getattr(self, "fold_w", None) is None:
            self._fold_setup()
        plan = _Plan()
        ops = plan.ops
        st, _ = self.stem

        def conv(x, xshape, c, relu, residual=None, out_hw=None):
            Nn, Hh, Ww, C = xshape
            OH, OW = out_hw or conv_out_hw(Hh, Ww, c.R, c.S, c.stride, c.pad)
            y = self._A(plan, (Nn, OH, OW, c.Cout))
            ops.append((lib.pfr_conv2d_fwd, (x.data_ptr(), c.wf.data_ptr(), y.data_ptr(), self.did, self.did, Nn, Hh, Ww, C, c.Cout,
                                             c.R, c.S, c.stride, c.pad, 0, OH, OW, c.Cout, c.bf.data_ptr(),
                                             0 if residual is None else residual.data_ptr(), 0, int(relu), 0, 0, 0, 0)))
            return y, (Nn, OH, OW, c.Cout)

        if self._use_s2d(H, W):
            x_nhwc = self._A(plan, (N, H // 2, W // 2, self.s2d.Cin))
            plan.meta["s2d"] = True
            c1, s1 = conv(x_nhwc, (N, H // 2, W // 2, self.s2d.Cin), self.s2d, True, out_hw=(H // 2, W // 2))
        else:
            x_nhwc = self._A(plan, (N, H, W, self.cp))
            c1, s1 = conv(x_nhwc, (N, H, W, self.cp), st, True)
        plan.meta["x_nhwc"] = x_nhwc
        _, H1, W1, _ = s1
        PH, PW = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
        cur = self._A(plan, (N, PH, PW, st.Cout))
        ops.append((lib.pfr_bn_relu_maxpool_fwd, (c1.data_ptr(), self.ident.data_ptr(), self.ident.data_ptr() + 4 * st.Cout,
                                                  cur.data_ptr(), 0, self.did, N, H1, W1, st.Cout, 0)))
        cshape = (N, PH, PW, st.Cout)
        for convs, down in self.blocks:
            xin, xshape = cur, cshape
            short = xin
            if down is not None:
                short, _ = conv(xin, xshape, down[0], False)
            z, zs = xin, xshape
            for ci, (c, _) in enumerate(convs):
                last = ci + 1 == len(convs)
                z, zs = conv(z, zs, c, True, residual=short if last else None)
            cur, cshape = z, zs
        Nn, Hh, Ww, Cf = cshape
        gap = self._A(plan, (N, Cf))
        ops.append((lib.pfr_avgpool_fwd, (cur.data_ptr(), gap.data_ptr(), self.did, N, Hh * Ww, Cf)))
        emb = self._A(plan, (N, self.emb_dim), torch.float32)
        f = self.fc
        self._conv_fwd(ops, gap, (N, 1, 1, Cf), f.w, emb, f, 1, 0, 1, 1, bias=f.bias)
        plan.meta["emb"] = emb
        plan.meta["fwd"] = ops
        plan.meta["folded"] = True
        return plan

    # ------------------------------------------------------------------------------------------ plan building
    def _A(self, plan, shape, dtype=None):
        t = torch.empty(shape, dtype=dtype or self.dtype, device=self.device)
        plan.bufs[len(plan.bufs)] = t
        return t

    def _conv_fwd(self, ops, x, xshape, w, y, c, stride, pad, OH, OW, pro=None, part=None, bias=None, idil=0,
                  accumulate=0, Cout=None, R=None, S=None):
        N, H, W, C = xshape
        Cout = Cout or c.Cout
        R = R or c.R
        S = S or c.S
        ps = psh = 0
        prelu = 0
        if pro is not None:
            ps, psh, prelu = pro[0].data_ptr(), pro[1].data_ptr(), 1
        ops.append((lib.pfr_conv2d_fwd, (x.data_ptr(), w.data_ptr(), y.data_ptr(), self.did, dtype_id(y.dtype), N, H, W, C,
                                         Cout, R, S, stride, pad, idil, OH, OW, y.shape[-1],
                                         0 if bias is None else bias.data_ptr(), 0, accumulate, 0, ps, psh, prelu,
                                         0 if part is None else part.data_ptr())))

    def _stats_buf(self, plan, xshape, c, OH, OW, pro=False):
        N, H, W, C = xshape
        M = N * OH * OW
        mt = lib.pfr_conv2d_mtile(N, H, W, C, c.Cout, c.R, c.S, c.stride, c.pad, OH, OW, self.did, self.did, int(pro))
        nt = (M + mt - 1) // mt
        return self._A(plan, (nt, 2, c.Cout), torch.float32), nt, mt

    def _bn_fwd(self, ops, bn, part, nparts, count, train, mt=0):
        if train:
            nws = lib.pfr_bn_finalize_ws_floats(nparts, bn.C)
            assert nws <= self.bn_ws.numel()
            ops.append((lib.pfr_bn_finalize, (part.data_ptr(), nparts, mt, bn.C,
                                              float(count), bn.gamma.data_ptr(), bn.beta.data_ptr(), float(bn.eps),
                                              float(bn.momentum), bn.rm.data_ptr(), bn.rv.data_ptr(), bn.coef[0].data_ptr(),
                                              bn.coef[1].data_ptr(), bn.coef[2].data_ptr(), bn.coef[3].data_ptr(),
                                              self.bn_ws.data_ptr() if nws else 0)))
        else:
            ops.append((lib.pfr_bn_eval_coeff, (bn.C, bn.gamma.data_ptr(), bn.beta.data_ptr(), bn.rm.data_ptr(),
                                                bn.rv.data_ptr(), float(bn.eps), bn.coef[2].data_ptr(),
                                                bn.coef[3].data_ptr())))

    def _conv_bn(self, plan, ops, x, xshape, c, bn, train, pro=None, w=None, out_hw=None):
        N, H, W, C = xshape
        OH, OW = out_hw or conv_out_hw(H, W, c.R, c.S, c.stride, c.pad)
        y = self._A(plan, (N, OH, OW, c.Cout))
        part, nt, mt = (None, 0, 0)
        if train:
            part, nt, mt = self._stats_buf(plan, xshape, c, OH, OW, pro is not None)
        self._conv_fwd(ops, x, xshape, w if w is not None else c.w, y, c, c.stride, c.pad, OH, OW, pro=pro, part=part)
        self._bn_fwd(ops, bn, part, nt, N * OH * OW, train, mt)
        return y, (N, OH, OW, c.Cout)

    def _dgrad_parts(self, dyshape, c, dxshape, two_bns=False, accumulates=False):
        """partial rows per BN of a data-gradient launch that also leaves BatchNorm-backward sums (0: separate reduce pass)"""
        if not self.fuse_bnb or (self.fuse_bnb == 2 and (two_bns or accumulates)):
            return 0
        if self.fuse_bnb == 2 and dxshape[0] * dxshape[1] * dxshape[2] < self._bnb_min_rows:
            return 0
        return lib.pfr_conv2d_dgrad_bn_parts(self.did, dyshape[0], dyshape[1], dyshape[2], dyshape[3], c.Cin, c.R, c.S,
                                             {1: 0, 2: 1}[c.stride], dxshape[1], dxshape[2])

    def _bnfree_set(self, cshape):
        """Blocks whose conv3 + bn3 take the BN-input-free form (pfr_bnfree.hip) for a network input that reaches layer1 with shape
        `cshape`: {block index: partial rows of conv3's streaming data gradient}.  Geometry only — the forward pass (which then does
        not store conv3's output) and the backward pass (which then never asks for it) both decide with this one function:
          * conv3 is 1x1 / stride 1 with 64 | 128 | 256 inputs and its data gradient is a streaming-kernel geometry;
          * the gradient at the block's output is produced by a streaming join of the NEXT block (conv1's data gradient of an
            identity block, or the compact-shortcut form of a projection block): those store it through this block's ReLU mask."""
        out = {}
        if not self.bnfree:
            return out
        shapes = []
        cur = cshape
        for convs, down in self.blocks:
            xs = cur
            zs = []
            for c, _ in convs:
                oh, ow = conv_out_hw(cur[1], cur[2], c.R, c.S, c.stride, c.pad)
                cur = (cur[0], oh, ow, c.Cout)
                zs.append(cur)
            shapes.append((xs, zs, cur))
        for k in range(len(self.blocks) - 1):
            convs, down = self.blocks[k]
            xs, zs, oshape = shapes[k]
            if len(convs) != 3:
                continue
            c3 = convs[2][0]
            if c3.R != 1 or c3.stride != 1 or c3.Cin not in (64, 128, 256) or c3.Cout % 64:
                continue
            npart = self._dgrad_parts(oshape, c3, zs[1])
            if npart <= 0 or lib.pfr_conv1x1_tail_mtile(self.did, zs[1][0], zs[1][1], zs[1][2], c3.Cin, c3.Cout) <= 0:
                continue
            # producer of this block's output gradient = the next block's first data gradient
            nconvs, ndown = self.blocks[k + 1]
            nxs, nzs, _ = shapes[k + 1]
            c0 = nconvs[0][0]
            if c0.stride != 1 or self._dgrad_parts(nzs[0], c0, nxs) <= 0:
                continue
            if ndown is not None:
                dc = ndown[0]
                if (down is not None or dc.R != 1 or dc.stride != 2 or nxs[1] % 2 or nxs[2] % 2):
                    continue
            out[k] = npart
        return out

    def build_plan(self, N, H, W, train, with_backward):
        plan = _Plan()
        ops = plan.ops
        T = self.dtype
        st, stbn = self.stem
        use_s2d = self._use_s2d(H, W)
        if use_s2d:
            xin_shape = (N, H // 2, W // 2, self.s2d.Cin)
            plan.meta["s2d"] = True
        else:
            xin_shape = (N, H, W, self.cp)
        x_nhwc = self._A(plan, xin_shape)
        plan.meta["x_nhwc"] = x_nhwc
        # batch-statistics coefficients are part of what a forward pass SAVES for its backward: one set per plan (pointers are
        # baked into the op lists below), so that a second forward before the first backward cannot overwrite them
        for b in self.bn_list:
            b.coef = torch.zeros((4, b.C), dtype=torch.float32, device=self.device)    # mean, invstd, scale, shift
            b.bcoef = torch.zeros((3, b.C), dtype=torch.float32, device=self.device)   # backward coefficients
            plan.bufs[len(plan.bufs)] = (b.coef, b.bcoef)
        saved = {}
        # ---- stem: conv 7x7/2 (space-to-depth form when H, W are even) → (BN+ReLU+MaxPool fused)
        if use_s2d:
            c1, s1 = self._conv_bn(plan, ops, x_nhwc, xin_shape, self.s2d, stbn, train, w=self.s2d.w, out_hw=(H // 2, W // 2))
        else:
            c1, s1 = self._conv_bn(plan, ops, x_nhwc, xin_shape, st, stbn, train, w=st.w_pad)
        _, H1, W1, _ = s1
        PH, PW = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
        pooled = self._A(plan, (N, PH, PW, st.Cout))
        idx = self._A(plan, (N, PH, PW, st.Cout), torch.uint8) if with_backward else None
        ops.append((lib.pfr_bn_relu_maxpool_fwd, (c1.data_ptr(), stbn.coef[2].data_ptr(), stbn.coef[3].data_ptr(),
                                                  pooled.data_ptr(), 0 if idx is None else idx.data_ptr(), self.did, N, H1,
                                                  W1, st.Cout, 1)))
        saved["stem"] = (c1, s1, idx, (N, PH, PW, st.Cout))
        cur, cshape = pooled, (N, PH, PW, st.Cout)
        # ---- residual blocks
        bsaved = []
        free_set = self._bnfree_set(cshape) if train else {}
        plan.meta["free_set"] = free_set
        for bk, (convs, down) in enumerate(self.blocks):
            xin, xshape = cur, cshape
            raws = []
            acts = []      # materialised relu(BN(c)) of the inner convs (None when fused into the consumer's prologue)
            pro = None
            src, sshape = xin, xshape
            for ci, (c, bn) in enumerate(convs):
                if bk in free_set and ci == 2:
                    # recompute form: this pass leaves only bn3's statistics; the tail launch below computes conv3 again
                    OH3, OW3 = conv_out_hw(sshape[1], sshape[2], c.R, c.S, c.stride, c.pad)
                    gram = plan.meta.get("gram", {}).get(bk)
                    if gram is not None and self.gram_stats:
                        # bn3's batch statistics from conv3's INPUT (Gram matrix + column sums of z2): no pass over conv3 at all
                        rows3 = sshape[0] * OH3 * OW3
                        if train:    # statistics + finalize in one launch (pfr_bn_stats_from_gram + pfr_bn_finalize give the same)
                            ops.append((lib.pfr_bn_finalize_from_gram, (gram.data_ptr(), c.w.data_ptr(), self.did, c.Cout, c.Cin, float(rows3),
                                                                        bn.gamma.data_ptr(), bn.beta.data_ptr(), float(bn.eps), float(bn.momentum),
                                                                        bn.rm.data_ptr(), bn.rv.data_ptr(), bn.coef[0].data_ptr(), bn.coef[1].data_ptr(),
                                                                        bn.coef[2].data_ptr(), bn.coef[3].data_ptr(), self.gram_flag.data_ptr())))
                        else:
                            self._bn_fwd(ops, bn, None, 1, rows3, train, rows3)
                    else:
                        part3, nt3, mt3 = self._stats_buf(plan, sshape, c, OH3, OW3)
                        ops.append((lib.pfr_conv1x1_stats, (src.data_ptr(), c.w.data_ptr(), self.did, sshape[0], sshape[1], sshape[2], c.Cin,
                                                            c.Cout, part3.data_ptr())))
                        self._bn_fwd(ops, bn, part3, nt3, sshape[0] * OH3 * OW3, train, mt3)
                    z2_in = (src, sshape)
                    raws.append((None, (sshape[0], OH3, OW3, c.Cout)))
                    acts.append(None)
                    sshape = raws[-1][1]
                    continue
                y, yshape = self._conv_bn(plan, ops, src, sshape, c, bn, train, pro=pro)
                raws.append((y, yshape))
                if ci + 1 < len(convs):
                    z = self._A(plan, yshape)
                    ops.append((lib.pfr_bn_act, (y.data_ptr(), bn.coef[2].data_ptr(), bn.coef[3].data_ptr(), 0, 0, 0,
                                                 z.data_ptr(), self.did, yshape[0] * yshape[1] * yshape[2], yshape[3], 1)))
                    acts.append(z)
                    pro = None
                    src, sshape = z, yshape
                    if bk in free_set and ci == 1:
                        # z2ᵀz2 and the column sums of z2 for the BN-input-free backward, while z2 is fresh in the Infinity Cache
                        rows2 = yshape[0] * yshape[1] * yshape[2]
                        nws = lib.pfr_gram_ws_floats(rows2, yshape[3])
                        if nws > 0:
                            gram = self._A(plan, (yshape[3] * yshape[3] + yshape[3],), torch.float32)
                            gws = plan.meta.get("gram_ws")
                            if gws is None or gws.numel() < nws:
                                gws = plan.meta["gram_ws"] = self._A(plan, (nws,), torch.float32)
                            ops.append((lib.pfr_gram_colsum, (z.data_ptr(), self.did, rows2, yshape[3], gram.data_ptr(), gws.data_ptr())))
                            plan.meta.setdefault("gram", {})[bk] = gram
                else:
                    acts.append(None)
                    pro = (bn.coef[2], bn.coef[3])
                    src, sshape = y, yshape
            src = raws[-1][0]
            lastc, lastbn = convs[-1]
            out = self._A(plan, sshape)
            rows = sshape[0] * sshape[1] * sshape[2]
            cd = None
            # sign of the pre-ReLU block output as a bit mask (1 bit instead of a 16-bit re-read, twice, in backward)
            rmask = self._A(plan, (rows, sshape[3] // self.kp), torch.uint8) if with_backward else None
            mptr = 0 if rmask is None else rmask.data_ptr()
            if bk in free_set:
                if rmask is None:     # (a training forward without a backward pass: the kernel still writes the mask)
                    rmask = self._A(plan, (rows, sshape[3] // self.kp), torch.uint8)
                res, a2, b2 = xin, 0, 0
                if down is not None:
                    dc, dbn = down
                    cd, _ = self._conv_bn(plan, ops, xin, xshape, dc, dbn, train)
                    res, a2, b2 = cd, dbn.coef[2].data_ptr(), dbn.coef[3].data_ptr()
                zt, zts = z2_in
                ops.append((lib.pfr_conv1x1_bn_tail, (zt.data_ptr(), lastc.w.data_ptr(), out.data_ptr(), rmask.data_ptr(), self.did, zts[0],
                                                      zts[1], zts[2], lastc.Cin, lastc.Cout, lastbn.coef[2].data_ptr(),
                                                      lastbn.coef[3].data_ptr(), res.data_ptr(), a2, b2)))
            elif down is not None:
                dc, dbn = down
                cd, _ = self._conv_bn(plan, ops, xin, xshape, dc, dbn, train)
                ops.append((lib.pfr_bn_act_mask, (src.data_ptr(), lastbn.coef[2].data_ptr(), lastbn.coef[3].data_ptr(),
                                                  cd.data_ptr(), dbn.coef[2].data_ptr(), dbn.coef[3].data_ptr(), out.data_ptr(),
                                                  mptr, self.did, rows, sshape[3], 1)))
            else:
                ops.append((lib.pfr_bn_act_mask, (src.data_ptr(), lastbn.coef[2].data_ptr(), lastbn.coef[3].data_ptr(),
                                                  xin.data_ptr(), 0, 0, out.data_ptr(), mptr, self.did, rows, sshape[3], 1)))
            bsaved.append((xin, xshape, raws, cd, out if rmask is None else rmask, sshape, acts))
            cur, cshape = out, sshape
        # ---- global average pool + fc
        Nn, Hh, Ww, Cf = cshape
        gap = self._A(plan, (N, Cf))
        ops.append((lib.pfr_avgpool_fwd, (cur.data_ptr(), gap.data_ptr(), self.did, N, Hh * Ww, Cf)))
        emb = self._A(plan, (N, self.emb_dim), torch.float32)
        f = self.fc
        self._conv_fwd(ops, gap, (N, 1, 1, Cf), f.w, emb, f, 1, 0, 1, 1, bias=f.bias)
        plan.meta["emb"] = emb
        plan.meta["n_fwd"] = len(ops)
        if not with_backward:
            return plan

        # =================================================================== backward (appended after n_fwd)
        # Weight gradients feed nothing downstream until the optimizer, so they run on a SIDE stream, concurrent with the
        # dgrad -> BN-backward chain of the following layers (MFMA/LDS-bound wgrad next to HBM-bound streaming kernels).
        # Symbolic ops: ("fork", k): side waits for main's current point; ("srec", k): side records "wgrad k done";
        # ("wait", k): main waits for wgrad k (emitted before a pooled buffer it read is handed out again, before every
        # grad-ready mark and at the end).  The pool is FIFO so that a re-used buffer is the one released longest ago.
        pool = {}
        lag = 32
        pending = {}   # data_ptr -> index of the last side-stream wgrad that reads this buffer
        nside = [0]

        def G(shape, dtype=None):
            key = (tuple(shape), dtype or T)
            lst = pool.setdefault(key, [])
            # FIFO, but a buffer whose side-stream reader was issued fewer than `lag` weight gradients ago is left alone (a
            # fresh one is allocated instead: HBM is plentiful, 18 GB peak at bs 256) -- re-using it would make the main
            # stream wait for the side stream, which runs behind; the wait still emitted for an old reader has long been
            # satisfied.  Measured: lag 0 -> 32 = 22.66 -> 21.9 ms/step.
            for i, t in enumerate(lst):
                k = pending.get(t.data_ptr())
                if k is None or nside[0] - k >= lag:
                    lst.pop(i)
                    if k is not None:
                        del pending[t.data_ptr()]
                        ops.append(("wait", (k,)))
                    return t
            return self._A(plan, shape, dtype)

        def release(t):
            pool.setdefault((tuple(t.shape), t.dtype), []).append(t)

        ws_need = [0]

        def wgrad(x, xshape, dy, dyshape, c, pro=None, out=None, C=None, acc=0):
            Nq, Hq, Wq, Cq = xshape
            _, OH, OW, Co = dyshape
            KK = c.R * c.S * Cq
            splits = lib.pfr_conv2d_wgrad_splits(Nq * OH * OW, Co, KK)
            ws_need[0] = max(ws_need[0], splits * Co * KK)
            ps = psh = 0
            prelu = 0
            if pro is not None:
                ps, psh, prelu = pro[0].data_ptr(), pro[1].data_ptr(), 1
            dst = out if out is not None else c.g
            k = nside[0]
            nside[0] += 1
            ops.append(("fork", (k,)))
            ops.append(("wgrad_noacc" if out is not None else "wgrad", (x.data_ptr(), dy.data_ptr(), dst.data_ptr(), None, self.did, Nq, Hq, Wq, Cq, Co, c.R, c.S,
                                  c.stride, c.pad, OH, OW, Co, ps, psh, prelu, 1.0, acc)))
            ops.append(("srec", (k,)))
            pending[dy.data_ptr()] = k

        ws_main_need = [0]

        def wgrad_main(x, xshape, dy, dyshape, c, out):
            """a weight-gradient GEMM on the MAIN stream (its result feeds the next launches), into `out`, own split-K workspace"""
            Nq, Hq, Wq, Cq = xshape
            _, OH, OW, Co = dyshape
            KK = c.R * c.S * Cq
            splits = lib.pfr_conv2d_wgrad_splits(Nq * OH * OW, Co, KK)
            ws_main_need[0] = max(ws_main_need[0], splits * Co * KK)
            ops.append(("wgrad_main", (x.data_ptr(), dy.data_ptr(), out.data_ptr(), None, self.did, Nq, Hq, Wq, Cq, Co, c.R, c.S,
                                       c.stride, c.pad, OH, OW, Co, 0, 0, 0, 1.0, 0)))

        def side_op(fn, args):
            """any launch on the side stream (fork / record like a weight gradient) → its index for a later ("wait", k)"""
            k = nside[0]
            nside[0] += 1
            ops.append(("fork", (k,)))
            ops.append(("sideop", (fn, args)))
            ops.append(("srec", (k,)))
            return k

        def dgrad(dy, dyshape, c, dx, dxshape, accumulate=0):
            log2 = {1: 0, 2: 1}[c.stride]
            self._conv_fwd(ops, dy, dyshape, c.wt, dx, c, 1, c.R - 1 - c.pad, dxshape[1], dxshape[2], idil=log2,
                           accumulate=accumulate, Cout=c.Cin)

        def bn_bwd(dout, out_act, x, xshape, bn, mask_mode, dx, gres, acc, pre=None):
            rows = xshape[0] * xshape[1] * xshape[2]
            C = xshape[3]
            oa = 0 if out_act is None else out_act.data_ptr()
            if pre is not None:
                part, nb = pre          # the launch that produced `dout` already left the partial sums (pfr_conv2d_dgrad_bn)
            else:
                nb = lib.pfr_colreduce_blocks(C, self.did, rows)
                part = G((nb, 2, C), torch.float32)
                ops.append((lib.pfr_bn_bwd_reduce, (dout.data_ptr(), oa, x.data_ptr(), bn.coef[0].data_ptr(), bn.coef[1].data_ptr(),
                                                    bn.coef[2].data_ptr(), bn.coef[3].data_ptr(), mask_mode, self.did, rows, C,
                                                    part.data_ptr())))
            ops.append((lib.pfr_bn_bwd_finalize, (part.data_ptr(), nb, C, float(rows), bn.gamma.data_ptr(), bn.coef[0].data_ptr(),
                                                  bn.coef[1].data_ptr(), bn.dgamma.data_ptr(), bn.dbeta.data_ptr(),
                                                  bn.bcoef.data_ptr(), acc)))
            ops.append((lib.pfr_bn_bwd_apply, (dout.data_ptr(), oa, x.data_ptr(), bn.bcoef.data_ptr(), bn.coef[2].data_ptr(),
                                               bn.coef[3].data_ptr(), mask_mode, dx.data_ptr(),
                                               0 if gres is None else gres.data_ptr(), self.did, rows, C)))
            release(part)

        dgrad_parts = self._dgrad_parts

        def dgrad_bn(dy, dyshape, c, dx, dxshape, bn1, bn2=None, res=None, res_mask=None, accumulate=0, flags=0, wt=None):
            """data gradient + BN-backward partial sums; bn1 = (x, bn record, bit mask or None, part), bn2 = (x, bn record, part);
            flags: pfr_conv2d_dgrad_bn_ex (1 = store through the bit mask, 2 = no input for bn1)"""
            x1, b1, mk, p1 = bn1
            x2p = c2p = p2p = 0
            if bn2 is not None:
                x2p, c2p, p2p = bn2[0].data_ptr(), bn2[1].coef.data_ptr(), bn2[2].data_ptr()
            args = (dy.data_ptr(), (c.wt if wt is None else wt).data_ptr(), dx.data_ptr(), self.did, dyshape[0], dyshape[1],
                    dyshape[2], dyshape[3], c.Cin, c.R, c.S, c.R - 1 - c.pad, {1: 0, 2: 1}[c.stride],
                    dxshape[1], dxshape[2], 0 if res is None else res.data_ptr(),
                    0 if res_mask is None else res_mask.data_ptr(), accumulate, 0 if x1 is None else x1.data_ptr(),
                    b1.coef.data_ptr(), 0 if mk is None else mk.data_ptr(), p1.data_ptr(), x2p, c2p, p2p)
            if flags:
                ops.append((lib.pfr_conv2d_dgrad_bn_ex, args + (flags,)))
            else:
                ops.append((lib.pfr_conv2d_dgrad_bn, args))

        acc = 0  # placeholder: _finalize_plan emits an overwrite (0) and an accumulate (1) variant of every grad write
        # fc
        demb = self._A(plan, (N, self.emb_dim))
        plan.meta["demb"] = demb
        if f.dbias is not None:
            ops.append(("colsum", (demb.data_ptr(), self.did, N, self.emb_dim, f.dbias.data_ptr(), acc)))
        wgrad(gap, (N, 1, 1, Cf), demb, (N, 1, 1, self.emb_dim), f)
        dgap = G((N, Cf))
        dgrad(demb, (N, 1, 1, self.emb_dim), f, dgap, (N, 1, 1, Cf))
        self._mark(ops, f.off)
        dcur = G(cshape)
        ops.append((lib.pfr_avgpool_bwd, (dgap.data_ptr(), dcur.data_ptr(), self.did, N, Hh * Ww, Cf)))
        release(dgap)
        # blocks in reverse
        nblk = len(self.blocks)
        pre3 = {}     # block index -> partial sums of its last BN / projection BN left by the producer of its output gradient
        # BN-input-free backward (pfr_bnfree.hip): geometry test per block, and — on the side stream, ahead of their use — the two
        # quantities that depend on forward values only: G2 = z2ᵀz2 and the column sums of z2
        bnf = {}        # block index -> dict(G2, zsum, side index, npart of conv3's streaming data gradient)
        premasked = {}  # block index -> True: the gradient at its output was stored through the block's ReLU mask by its producer
        if free_set:
            for k, npart in free_set.items():
                convs, _ = self.blocks[k]
                _, _, raws, _, _, oshape, acts = bsaved[k]
                c3 = convs[2][0]
                zs = raws[1][1]
                rows = zs[0] * zs[1] * zs[2]
                gram = plan.meta.get("gram", {}).get(k)
                if gram is not None:      # computed by the forward pass (pfr_gram_colsum)
                    bnf[k] = dict(G2=gram[:c3.Cin * c3.Cin], zsum=gram[c3.Cin * c3.Cin:], side=None, npart=npart)
                    continue
                G2 = self._A(plan, (c3.Cin, c3.Cin), torch.float32)
                zsum = self._A(plan, (c3.Cin,), torch.float32)
                cws = self._A(plan, (max(1, lib.pfr_colsum_ws_floats(rows, c3.Cin)),), torch.float32)
                wgrad(acts[1], zs, acts[1], zs, c3, out=G2)
                kk = side_op(lib.pfr_colsum, (acts[1].data_ptr(), self.did, rows, c3.Cin, zsum.data_ptr(), 0, cws.data_ptr()))
                bnf[k] = dict(G2=G2, zsum=zsum, side=kk, npart=npart)
            if bnf:
                cmax = max(self.blocks[k][0][2][0].Cout for k in bnf)
                kmax = max(self.blocks[k][0][2][0].Cin for k in bnf)
                bnf_G1 = self._A(plan, (cmax * kmax,), torch.float32)
                bnf_coef = self._A(plan, (3 * cmax,), torch.float32)
                bnf_wat = self._A(plan, ((cmax + kmax) * kmax,))      # [K][C] (A∘W)ᵀ, or — two-source form — wcat [K][C + K] = [(A∘W)ᵀ | S]
                bnf_S = self._A(plan, (kmax * kmax,))
                bnf_bias = self._A(plan, (kmax,), torch.float32)
        for k in range(nblk - 1, -1, -1):
            convs, down = self.blocks[k]
            xin, xshape, raws, cd, out, oshape, acts = bsaved[k]
            lastc, lastbn = convs[-1]
            ylast, _ = raws[-1]
            # BN(last) + residual + ReLU backward.  `out` is the block's ReLU bit mask; dcur (the gradient that arrived at
            # the block output) is KEPT: the residual branch consumes it through the same mask (no masked copy is written)
            p3 = pre3.get(k)
            rmask = out
            free3 = k in bnf
            if free3 and not (premasked.get(k, False) and p3 is not None):
                raise PfrError(f"block {k}: the forward pass dropped conv3's output but its output gradient was not produced masked")
            if free3:
                # conv3 + bn3 without bn3's input (pfr_bnfree.hip).  dcur holds G = g∘mask already.
                c3, bn3 = convs[2]
                _, bn2 = convs[1]
                z2, (c2raw, zs) = acts[1], raws[1]
                f = bnf[k]
                C3, K3, rows3 = c3.Cout, c3.Cin, float(oshape[0] * oshape[1] * oshape[2])
                wgrad_main(z2, zs, dcur, oshape, c3, bnf_G1)                                   # G1 = Gᵀ z2
                if f["side"] is not None:
                    ops.append(("wait", (f["side"],)))                                          # G2, zsum (side stream, issued long ago)
                part3, np3 = p3[0]
                # W = the bf16 weights the forward convolution multiplied with: every term that reconstructs x = z2·Wᵀ (dγ, S, the B-term of
                # dW) must be built from the x that was actually normalised, not from the fp32 masters
                Wm = c3.w.data_ptr()
                ops.append((lib.pfr_bn3_bwd_coef, (part3.data_ptr(), np3, bnf_G1.data_ptr(), f["zsum"].data_ptr(), Wm, self.did, bn3.gamma.data_ptr(),
                                                   bn3.coef[1].data_ptr(), C3, K3, rows3, bn3.dgamma.data_ptr(), bn3.dbeta.data_ptr(),
                                                   bnf_coef.data_ptr(), acc)))
                np2 = lib.pfr_conv1x1_dgrad2_bn_parts(self.did, zs[0], zs[1], zs[2], C3, K3, K3)
                ops.append((lib.pfr_bn3_bwd_weights, (bnf_coef.data_ptr(), bnf_G1.data_ptr(), f["G2"].data_ptr(), f["zsum"].data_ptr(), Wm, self.did,
                                                      C3, K3, rows3, c3.g.data_ptr(), bnf_wat.data_ptr(), 0 if np2 > 0 else bnf_S.data_ptr(),
                                                      bnf_bias.data_ptr(), acc)))
                release(part3)
                dz2 = G(zs)
                if np2 > 0:
                    # ONE launch over both row sources [G | z2] against wcat = [(A∘W)ᵀ | S], bias inside the accumulators
                    part2 = G((np2, 2, K3), torch.float32)
                    ops.append((lib.pfr_conv1x1_dgrad2_bn, (dcur.data_ptr(), z2.data_ptr(), bnf_wat.data_ptr(), bnf_bias.data_ptr(), dz2.data_ptr(),
                                                            self.did, zs[0], zs[1], zs[2], C3, K3, K3, c2raw.data_ptr(), bn2.coef.data_ptr(),
                                                            part2.data_ptr())))
                else:
                    np2 = f["npart"]
                    y1 = G(zs)
                    ops.append((lib.pfr_conv2d_fwd, (z2.data_ptr(), bnf_S.data_ptr(), y1.data_ptr(), self.did, self.did, zs[0], zs[1], zs[2], K3, K3,
                                                     1, 1, 1, 0, 0, zs[1], zs[2], K3, bnf_bias.data_ptr(), 0, 0, 0, 0, 0, 0, 0)))
                    part2 = G((np2, 2, K3), torch.float32)
                    dgrad_bn(dcur, oshape, c3, dz2, zs, (c2raw, bn2, None, part2), res=y1, wt=bnf_wat)
                    release(y1)
                bn_bwd(dz2, None, c2raw, zs, bn2, 2, dz2, None, acc, pre=(part2, np2))
                dy, dyshape = dz2, zs
            else:
                dz3 = G(oshape)
                bn_bwd(dcur, out, ylast, oshape, lastbn, 3, dz3, None, acc, pre=None if p3 is None else p3[0])
                dy, dyshape = dz3, oshape
            for i in range(len(convs) - 1 - (1 if free3 else 0), 0, -1):
                c, bn = convs[i]
                pc, pbn = convs[i - 1]
                xraw, xrs = raws[i - 1]
                wgrad(acts[i - 1], xrs, dy, dyshape, c)
                dz = G(xrs)
                npart = dgrad_parts(dyshape, c, xrs)
                if npart > 0:
                    part = G((npart, 2, xrs[3]), torch.float32)
                    dgrad_bn(dy, dyshape, c, dz, xrs, (xraw, pbn, None, part))
                    release(dy)
                    bn_bwd(dz, None, xraw, xrs, pbn, 2, dz, None, acc, pre=(part, npart))
                else:
                    dgrad(dy, dyshape, c, dz, xrs)
                    release(dy)
                    bn_bwd(dz, None, xraw, xrs, pbn, 2, dz, None, acc)
                dy, dyshape = dz, xrs
            c0, bn0 = convs[0]
            wgrad(xin, xshape, dy, dyshape, c0)
            dxin = G(xshape)
            # the launch that finishes dxin also leaves the BN-backward sums of the PREVIOUS block's output BN(s), whose
            # output gradient dxin is (through that block's ReLU bit mask)
            nxt = None
            if k > 0:
                pconvs, pdown = self.blocks[k - 1]
                _, _, praws, pcd, pmask, poshape, _ = bsaved[k - 1]
                nxt = (praws[-1][0], pconvs[-1][1], pmask, pcd, None if pdown is None else pdown[1])
            if down is not None:
                dc, dbn = down
                dgd = G(oshape)
                bn_bwd(dcur, rmask, cd, oshape, dbn, 3, dgd, None, acc,
                       pre=None if (p3 is None or p3[1] is None) else p3[1])      # projection-shortcut BN: g = dcur ∘ mask
                release(dcur)
                wgrad(xin, xshape, dgd, oshape, dc)
                # (the in-place form — the shortcut writes all of dxin first, the main branch adds to it — measured slower, retired in round 4)
                npart3 = 0
                if (nxt is not None and self.fuse_bnb == 2 and nxt[3] is None and dc.R == 1 and dc.stride == 2
                        and xshape[1] % 2 == 0 and xshape[2] % 2 == 0):
                    npart3 = dgrad_parts(dyshape, c0, xshape)
                if npart3 > 0:
                    # the shortcut's gradient densely on its own grid (a plain GEMM), then the main branch's streaming join adds it at
                    # the even pixels and leaves the previous block's BN-backward sums: no scattered accumulate pass over dxin
                    comp = G((xshape[0], oshape[1], oshape[2], xshape[3]))
                    self._conv_fwd(ops, dgd, oshape, dc.wt, comp, dc, 1, 0, oshape[1], oshape[2], idil=0, Cout=dc.Cin)
                    part = G((npart3, 2, xshape[3]), torch.float32)
                    sub_args = (dy.data_ptr(), c0.wt.data_ptr(), dxin.data_ptr(), self.did, dyshape[0], dyshape[1], dyshape[2], dyshape[3],
                                c0.Cin, xshape[1], xshape[2], comp.data_ptr(), 0 if nxt[0] is None else nxt[0].data_ptr(), nxt[1].coef.data_ptr(),
                                nxt[2].data_ptr(), part.data_ptr())
                    if (k - 1) in bnf:      # the previous block takes the BN-input-free backward: its output gradient leaves masked,
                        ops.append((lib.pfr_conv2d_dgrad_bn_sub_ex, sub_args + (3,)))     # and only sum g*mask is asked of this launch
                        premasked[k - 1] = True
                    else:
                        ops.append((lib.pfr_conv2d_dgrad_bn_sub, sub_args))
                    pre3[k - 1] = ((part, npart3), None)
                    release(comp)
                    npart = -1
                else:
                    # main branch first (writes all of dxin), then the projection shortcut ACCUMULATES: for its 1x1 / stride-2
                    # conv only the (even, even) positions of dxin receive anything, and only those rows are touched
                    dgrad(dy, dyshape, c0, dxin, xshape)
                    npart = dgrad_parts(oshape, dc, xshape, accumulates=True) if nxt is not None else 0
                if npart < 0:
                    pass
                elif npart > 0:
                    part = G((npart, 2, xshape[3]), torch.float32)
                    part2 = G((npart, 2, xshape[3]), torch.float32) if nxt[3] is not None else None
                    dgrad_bn(dgd, oshape, dc, dxin, xshape, (nxt[0], nxt[1], nxt[2], part),
                             None if part2 is None else (nxt[3], nxt[4], part2), accumulate=1)
                    pre3[k - 1] = ((part, npart), None if part2 is None else (part2, npart))
                else:
                    dgrad(dgd, oshape, dc, dxin, xshape, accumulate=1)
                release(dgd)
            else:
                # identity shortcut: dxin = dgrad(conv1) + dcur ∘ mask in the data-gradient epilogue
                log2 = {1: 0, 2: 1}[c0.stride]
                npart = dgrad_parts(dyshape, c0, xshape) if nxt is not None else 0   # (the join form also takes the two-BN case)
                if npart > 0:
                    part = G((npart, 2, xshape[3]), torch.float32)
                    part2 = G((npart, 2, xshape[3]), torch.float32) if nxt[3] is not None else None
                    dgrad_bn(dy, dyshape, c0, dxin, xshape, (nxt[0], nxt[1], nxt[2], part),
                             None if part2 is None else (nxt[3], nxt[4], part2), res=dcur, res_mask=rmask,
                             flags=3 if (k - 1) in bnf else 0)
                    if (k - 1) in bnf:
                        premasked[k - 1] = True
                    pre3[k - 1] = ((part, npart), None if part2 is None else (part2, npart))
                else:
                    ops.append((lib.pfr_conv2d_dgrad_join, (dy.data_ptr(), c0.wt.data_ptr(), dxin.data_ptr(), self.did, dyshape[0],
                                                            dyshape[1], dyshape[2], dyshape[3], c0.Cin, c0.R, c0.S, c0.R - 1 - c0.pad,
                                                            log2, xshape[1], xshape[2], dcur.data_ptr(), rmask.data_ptr())))
                release(dcur)
            self._mark(ops, c0.off)  # conv1.weight is the block's first parameter: flat grads [c0.off, end) are final
            release(dy)
            dcur = dxin
        # stem
        c1, s1, idx, pshape = saved["stem"]
        dz = G(s1)
        ops.append((lib.pfr_maxpool_bwd, (dcur.data_ptr(), idx.data_ptr(), dz.data_ptr(), self.did, N, s1[1], s1[2], s1[3])))
        release(dcur)
        bn_bwd(dz, None, c1, s1, stbn, 2, dz, None, acc)
        if use_s2d:
            wgrad(x_nhwc, xin_shape, dz, s1, self.s2d, out=self.s2d.g)
            ops.append(("wait", (nside[0] - 1,)))   # the un-packing below reads what the stem wgrad wrote
            ops.append(("s2dunpack", (self.s2d.g.data_ptr(), st.g.data_ptr(), st.Cout, st.Cin, self.s2d.Cin, acc)))
        else:
            wgrad(x_nhwc, xin_shape, dz, s1, st, out=st.g_pad)
            ops.append(("wait", (nside[0] - 1,)))   # the un-padding copy below reads what the stem wgrad wrote
            ops.append(("copy2d", (st.g_pad.data_ptr(), self.cp, st.g.data_ptr(), st.Cin, st.Cout * st.R * st.S, st.Cin, 1.0, acc)))
        self._mark(ops, 0)
        plan.meta["n_side"] = nside[0]
        # workspace
        if self.ws is None or self.ws.numel() < ws_need[0]:
            self.ws = torch.empty(ws_need[0], dtype=torch.float32, device=self.device)
        plan.meta["ws_need"] = ws_need[0]
        if ws_main_need[0] and (self.ws_main is None or self.ws_main.numel() < ws_main_need[0]):
            self.ws_main = torch.empty(ws_main_need[0], dtype=torch.float32, device=self.device)
        plan.meta["bnfree_blocks"] = sorted(k for k in bnf if premasked.get(k))
        assert plan.meta["bnfree_blocks"] == sorted(free_set), (plan.meta["bnfree_blocks"], sorted(free_set))
        return plan

    def _mark(self, ops, off):
        # gradients of flat offsets >= off are final once main has also seen the side stream's latest wgrad
        last = max((a[0] for f, a in ops if f == "srec"), default=None)
        if last is not None:
            # off == 0 is the end of the backward pass: always joined.  Intermediate marks only matter to a bucket hook.
            ops.append(("wait" if off == 0 else "mwait", (last,)))
        ops.append((None, (off,)))

    # ------------------------------------------------------------------------------------------ execution
    def _check_tuning(self):
        """A plan bakes kernel choices in (statistics-partial granularity, partial-row counts of the fused BatchNorm sums, which
        blocks take the BN-input-free form).  When a pfr_set_tuning call changed a knob since the plans were built — another engine,
        a test, a host sweep — this engine's own mode is re-asserted and every plan no forward pass still owns is rebuilt."""
        ep = lib.pfr_tuning_epoch()
        stale = False
        if self.gram_stats and int(self.gram_flag[0]) != 0:
            import warnings
            warnings.warn("pfr: bn3 statistics from the Gram matrix lost precision to cancellation (a nearly constant conv3 channel); "
                          "falling back to the statistics pass over conv3's output for the rest of the run")
            self.gram_stats = False
            stale = True
        if ep == self._tuning_epoch and not stale:
            return
        lib.pfr_set_tuning(b"bnb", self.fuse_bnb)
        self._tuning_epoch = lib.pfr_tuning_epoch()
        for k, q in list(self.plans.items()):
            if not self._plan_busy(q):
                self.plans.pop(k)

    def get_plan(self, N, H, W, train, with_backward, slot=0):
        self._check_tuning()
        key = (N, H, W, train, with_backward) + ((slot,) if slot else ())
        p = self.plans.get(key)
        if p is None:
            if len(self.plans) >= 8:
                # evict the oldest plan that no forward pass in flight still owns
                for k, q in list(self.plans.items()):
                    if not self._plan_busy(q):
                        self.plans.pop(k)
                        break
            if not train and not with_backward and self.fold_eval:
                p = self.build_eval_plan(N, H, W)
            else:
                p = self.build_plan(N, H, W, train, with_backward)
                self._finalize_plan(p)
            self.plans[key] = p
        return p

    @staticmethod
    def _plan_busy(plan):
        own = plan.meta.get("owner")
        return own is not None and own() is not None

    def acquire_plan(self, N, H, W, train, with_backward, ticket):
        """A plan owns the activation buffers its backward pass reads.  Two training forwards before a backward (list input of
        SoftmaxBasedMetricLearning — reference losses/__init__.py:39 — or any two-view step) therefore get DIFFERENT plan
        instances ("slots"); a slot is free again when its backward ran or its autograd node died."""
        slot = 0
        while True:
            plan = self.get_plan(N, H, W, train, with_backward, slot)
            if ticket is None or not self._plan_busy(plan):
                break
            slot += 1
            if slot >= 8:
                raise PfrError("more than 8 forward passes of one shape are waiting for their backward pass")
        if ticket is not None:
            plan.meta["owner"] = weakref.ref(ticket)
        return plan

    def _finalize_plan(self, plan):
        """Resolve symbolic ops (workspace pointer, accumulate flag) into two concrete op lists."""
        fwd = plan.ops[:plan.meta["n_fwd"]]
        bwd = plan.ops[plan.meta["n_fwd"]:]
        plan.meta["fwd"] = fwd
        for acc in (0, 1):
            res = []
            for fn, args in bwd:
                if fn in ("wgrad", "wgrad_noacc"):
                    a = list(args)
                    a[3] = self.ws.data_ptr()
                    a[-1] = acc if fn == "wgrad" else 0
                    res.append((_SIDE, (lib.pfr_conv2d_wgrad, tuple(a))))
                elif fn == "wgrad_main":
                    a = list(args)
                    a[3] = self.ws_main.data_ptr() if self.ws_main is not None else 0
                    res.append((lib.pfr_conv2d_wgrad, tuple(a)))
                elif fn == "sideop":
                    res.append((_SIDE, (args[0], tuple(args[1]))))
                elif fn == "fork":
                    res.append((_FORK, args[0]))
                elif fn == "srec":
                    res.append((_SREC, args[0]))
                elif fn == "wait":
                    res.append((_WAIT, args[0]))
                elif fn == "mwait":
                    res.append((_MWAIT, args[0]))
                elif fn == "colsum":
                    res.append((lib.pfr_colsum, tuple(args[:-1]) + (acc, 0)))
                elif fn == "copy2d":
                    res.append((lib.pfr_copy2d_f32, tuple(args[:-1]) + (acc,)))
                elif fn == "s2dunpack":
                    res.append((lib.pfr_s2d_wgrad, tuple(args[:-1]) + (acc,)))
                elif fn is lib.pfr_bn_bwd_finalize or fn is lib.pfr_bn3_bwd_coef \
                        or fn is lib.pfr_bn3_bwd_weights:
                    res.append((fn, tuple(args[:-1]) + (acc,)))
                else:
                    res.append((fn, args))
            plan.meta["bwd%d" % acc] = res
            plan.meta.pop("c_bwd%d" % acc, None)
        plan.meta["ws_ptr"] = self._ws_key()

    def _ws_key(self):
        """the split-K workspaces a resolved plan has baked in (side stream, main stream)"""
        return (self.ws.data_ptr() if self.ws is not None else 0, self.ws_main.data_ptr() if self.ws_main is not None else 0)

    def forward(self, x, train, with_backward, ticket=None):
        if x.dim() != 4 or x.shape[1] != self.stem[0].Cin:
            raise PfrError(f"expected NCHW input with {self.stem[0].Cin} channels, got {tuple(x.shape)}")
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        N, _, H, W = x.shape
        plan = self.acquire_plan(N, H, W, train, with_backward, ticket if with_backward else None)
        plan.meta["epoch"] = self._tuning_epoch     # the knobs this forward pass (and the launches its backward replays) ran under
        if with_backward and plan.meta.get("ws_ptr", 0) != self._ws_key():
            self._finalize_plan(plan)
        if plan.meta.get("folded") and self.graph_eval and _TRACER[0] is None:
            return self._forward_graphed(plan, x)
        stream = torch.cuda.current_stream().cuda_stream
        if plan.meta.get("folded") and self.fold_cache:
            # inference: cast + fold only when a parameter or running statistic changed since the folded weights were made
            # (decided on the device by a checksum, pfr_fold_bn_cached); the stem's two small layout copies are always redone
            self.refresh_weights(stream, for_backward=False, cast=False)
            if self.fold_state is None:
                self.fold_state = torch.tensor([0, -1, 0, 0], dtype=torch.int64, device=self.device)
            lib.pfr_fold_bn_cached(self.fold_desc.data_ptr(), self.fold_n, self.did, self.master.data_ptr(), self.n_flat,
                                   self.shadow.data_ptr() if self.dtype != torch.float32 else 0, self.fold_state.data_ptr(), stream)
        else:
            self.refresh_weights(stream, for_backward=with_backward)
            if plan.meta.get("folded"):
                lib.pfr_fold_bn(self.fold_desc.data_ptr(), self.fold_n, self.did, stream)
        self._input_layout(plan, x, stream)
        if not self._run_list(plan, "fwd", stream):
            for fn, args in plan.meta["fwd"]:
                fn(*args, stream)
        if train:
            self.nbt.add_(1)
        self._last_plan = plan
        return plan.meta["emb"]

    def _run_list(self, plan, key, stream, side=0, hook=None, n_events=0):
        """replays plan.meta[key] (a launch list) — through the C executor when possible"""
        if self.c_plan and _TRACER[0] is None:
            ck = "c_" + key
            cp = plan.meta.get(ck, False)
            if cp is False:
                cp = plan.meta[ck] = CPlan.compile(plan.meta[key], n_events)
            if cp is not None:
                cp.run(stream, side, hook, self.hook_syncs_side)
                return True
        return False

    def _eval_launches(self, plan, xin):
        """weight refresh + BN fold + input layout + every launch of the inference plan, on the current stream"""
        N, _, H, W = xin.shape
        stream = torch.cuda.current_stream().cuda_stream
        self.refresh_weights(stream, for_backward=False)
        lib.pfr_fold_bn(self.fold_desc.data_ptr(), self.fold_n, self.did, stream)
        self._input_layout(plan, xin, stream)
        if not self._run_list(plan, "fwd", stream):
            for fn, args in plan.meta["fwd"]:
                fn(*args, stream)

    def _forward_graphed(self, plan, x):
        """Opt-in (PFR_GRAPH_EVAL=1): the inference plan replayed as ONE hipGraph, captured on the second call (the first
        runs eagerly and warms every kernel up); pointers are fixed (the input is copied into a static buffer) and the
        weights are re-read from the master buffer inside the graph, so optimizer steps / checkpoint loads between calls
        are honoured.  Measured: no gain — 1.38 ms per batch of 20 and 0.96 ms per single image either way, because the
        ~60 dependent launches are bound by their own k-loop latency, not by the 4 µs host cost of a launch."""
        m = plan.meta
        if "x_in" not in m:
            m["x_in"] = torch.empty_like(x)
        xin = m["x_in"]
        xin.copy_(x)
        g = m.get("graph")
        if g is None:
            if m.get("warm", 0) < 1:
                m["warm"] = 1
                self._eval_launches(plan, xin)
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._eval_launches(plan, xin)
                m["graph"] = g
                g.replay()
        else:
            g.replay()
        self._last_plan = plan
        return m["emb"]

    def _run_bwd_ops(self, plan, acc, hook):
        main = torch.cuda.current_stream()
        stream = main.cuda_stream
        use_side = self._side_ok()
        if use_side and self.side is None:
            self.side = torch.cuda.Stream(device=self.device)
        side_handle = self.side.cuda_stream if use_side else 0
        if self._run_list(plan, "bwd%d" % acc, stream, side_handle,
                          (lambda off: hook(off)) if hook is not None else None, 2 * plan.meta.get("n_side", 0)):
            return
        if use_side:
            side, sptr = self.side, self.side.cuda_stream
            ev = self.side_events
            while len(ev) < 2 * plan.meta.get("n_side", 0):
                ev.append(torch.cuda.Event())
        for fn, args in plan.meta["bwd%d" % acc]:
            if fn is None:
                if hook is not None:
                    hook(args[0])
            elif fn.__class__ is int:
                if not use_side:
                    if fn == _SIDE:
                        args[0](*args[1], stream)
                elif fn == _SIDE:
                    args[0](*args[1], sptr)
                elif fn == _FORK:
                    e = ev[2 * args]
                    e.record(main)
                    side.wait_event(e)
                elif fn == _SREC:
                    ev[2 * args + 1].record(side)
                elif fn == _WAIT or (hook is not None and not self.hook_syncs_side):
                    main.wait_event(ev[2 * args + 1])
            else:
                fn(*args, stream)

    def backward(self, demb, plan=None):
        plan = plan if plan is not None else self._last_plan
        if lib.pfr_tuning_epoch() != plan.meta.get("epoch", self._tuning_epoch):
            # the backward list was built for the kernels the knobs selected at forward time (partial-row counts of the fused BatchNorm
            # sums, which blocks dropped conv3's output): replaying it under other knobs would compute wrong sums silently
            raise PfrError("a pfr_set_tuning call changed a kernel-selection knob between this forward pass and its backward pass "
                           "(another engine with a different PFR_FUSE_BNB mode, or a host sweep): run forward again")
        stream = torch.cuda.current_stream().cuda_stream
        demb = demb.contiguous()
        if demb.numel() != plan.meta["demb"].numel():
            raise PfrError(f"backward: gradient of {tuple(demb.shape)} does not match the plan's embedding buffer "
                           f"{tuple(plan.meta['demb'].shape)}")
        if plan.meta.get("ws_ptr", 0) != self._ws_key():
            self._finalize_plan(plan)   # the shared weight-gradient workspace grew after this plan was resolved
        lib.pfr_cast(demb.data_ptr(), dtype_id(demb.dtype), plan.meta["demb"].data_ptr(), self.did, demb.numel(), stream)
        acc = 1 if self.first_param.grad is not None else 0
        plan.meta["owner"] = None
        # gradients are final (DDP bucket hook) only when no other forward pass still waits for its backward
        hook = self.grad_ready_hook if not any(self._plan_busy(q) for q in self.plans.values()) else None
        if hook is None and self.grad_ready_hook is not None and not getattr(self, "_warned_hook", False):
            import warnings
            self._warned_hook = True
            warnings.warn("FEEngine.backward: another forward pass of this model still holds its activations (a retained autograd "
                          "graph?), so gradient buckets are reduced after the backward pass instead of overlapped with it")
        main = torch.cuda.current_stream()
        if self.wt_pending:
            main.wait_event(self.wt_ready)
            self.wt_pending = False
        self._run_bwd_ops(plan, acc, hook)
        if not acc:
            self.attach_grads()


class _FEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, *params):
        eng = model.hip_engine(x.device)
        ctx.ticket = PlanTicket()
        emb = eng.forward(x, model.training, True, ctx.ticket)  # only reached when a backward pass can follow (see fe_forward)
        ctx.eng = eng
        ctx.plan = eng._last_plan
        ctx.nparams = len(params)
        return emb.clone()

    @staticmethod
    def backward(ctx, demb):
        if ctx.plan.meta.get("owner") is None or ctx.plan.meta["owner"]() is not ctx.ticket:
            raise PfrError("backward: the activations of this forward pass were released (double backward?)")
        ctx.eng.backward(demb, ctx.plan)
        # parameter gradients are delivered by side effect into the flat gradient buffer (p.grad views)
        return (None, None) + (None,) * ctx.nparams


def fe_forward(model, x):
    eng = model.hip_engine(x.device)
    if torch.is_grad_enabled() and any(p.requires_grad for p in eng.param_list):
        return _FEFunction.apply(x, model, *eng.param_list)
    return eng.forward(x, model.training, False).clone()
