"""SwinEngine — executes the Swin-T/S/B/L feature extractor (forward, backward) on the gfx950 kernels.

Counterpart of /root/reference/models/swin.py:196-225 (`SwinTransformer.forward`) + autograd for BASELINE config 4.
Same design as models/_fe_engine.FEEngine: flat fp32 master / gradient buffers (the module's nn.Parameters become views,
state-dict names unchanged), compute-dtype shadow, one pre-built plan of C-ABI calls per input shape.

Mapping (tokens are kept NHWC = [B, H, W, C] end to end; the reference's NCHW↔NHWC permutes between stages vanish):
  PatchMerging (Unfold + Linear, swin.py:155-167)  → stride-f conv on NHWC (weights re-laid out per step)
  LayerNorm / GELU / windowed attention            → csrc/pfr_swin.hip
  to_qkv / to_out / MLP / head Linear              → pfr_conv2d_fwd (GEMM, bias and residual-add fused in the epilogue),
                                                     pfr_conv2d_wgrad, data gradient through pfr_conv2d_fwd
  cyclic shift, window partition, masks            → addressing / analytic mask inside the attention kernel
"""
import os
import weakref

import torch
import torch.nn as nn

from .._hip import lib, dtype_id, PfrError
from .._hip.lib import _TRACER
from .._hip.cplan import CPlan
from ._fe_engine import default_compute_dtype, _ALIGN, _SIDE, _FORK, _SREC, _WAIT, _MWAIT, _side_with_ddp, PlanTicket


class _Lin:
    pass


class _LN:
    pass


class SwinEngine:
    def __init__(self, model, device, compute_dtype=None):
        if not str(device).startswith("cuda"):
            raise PfrError("SwinEngine runs on the HIP device only (no CPU fallback)")
        lib.pfr_version()
        self.device = torch.device(device)
        self.dtype = compute_dtype or default_compute_dtype()
        self.did = dtype_id(self.dtype)
        self.kp = 8 if self.dtype == torch.bfloat16 else 4
        self.model_id = id(model)
        self.plans = {}
        self.side = None          # side stream of the weight-gradient / column-sum launches (see build_plan)
        self.side_events = []
        self.fuse_gelu = True
        self.wt_fork = self.wt_ready = None
        self.wt_pending = False
        self.hook_syncs_side = False
        # replay the launch lists from C (csrc/pfr_plan.hip) instead of the interpreter loop (PFR_C_PLAN=0 keeps the loop; a launch
        # tracer always uses it)
        self.c_plan = os.environ.get("PFR_C_PLAN", "1") != "0"
        # buffers per (shape, dtype) class of the backward pool before one that a side-stream op still reads is re-used (HBM is
        # plentiful; a shallow pool makes the main stream wait for the side stream at almost every layer)
        self.pool_depth = 48
        self.ln_dxsum = True   # bias gradients from the LayerNorm-backward pass that produced their input
        self.side_stream_enabled = os.environ.get("PFR_SIDE_STREAM", "1") != "0"
        self.grad_ready_hook = None
        self._adopt(model)

    # ------------------------------------------------------------------------------------------ parameters
    def _adopt(self, model):
        dev = self.device
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        offs, total = {}, 0
        for name, p in named:
            offs[name] = total
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.n_flat = total
        self.master = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.shadow = self.master if self.dtype == torch.float32 else torch.zeros(total, dtype=self.dtype, device=dev)
        self.offs = offs
        self._views = {}
        self.param_list = []
        for name, p in named:
            o, n = offs[name], p.numel()
            mv = self.master[o:o + n].view(p.shape)
            mv.copy_(p.data.detach().to(dev))
            p.data = mv
            p.grad = None
            self._views[name] = (p, self.grad[o:o + n].view(p.shape))
            self.param_list.append(p)
        for n, p in model.named_parameters():          # frozen masks just move to the device
            if not p.requires_grad:
                p.data = p.data.to(dev)
        self.first_param = named[0][1]

        def lin(prefix, m, conv_f=None, cin=None):
            r = _Lin()
            r.out, r.inp = m.out_features, m.in_features
            r.off = offs[prefix + ".weight"]
            r.g = self.grad[r.off:r.off + r.out * r.inp]
            if m.bias is not None:
                bo = offs[prefix + ".bias"]
                r.bias = self.master[bo:bo + r.out]
                r.dbias = self.grad[bo:bo + r.out]
            else:
                r.bias = r.dbias = None
            r.f = conv_f
            if conv_f is None:
                r.w = self.shadow[r.off:r.off + r.out * r.inp]                       # [out][in] = [out,1,1,in]
                r.wt = torch.zeros(r.inp * r.out, dtype=self.dtype, device=dev)     # [in,1,1,out]
                r.cin = r.cinp = r.inp
            else:                                                                    # patch merging as a conv
                r.cin = cin
                r.cinp = (cin + self.kp - 1) // self.kp * self.kp
                kk = conv_f * conv_f * r.cinp
                r.w = torch.zeros(r.out * kk, dtype=self.dtype, device=dev)          # [out][f][f][cinp]
                r.wt = torch.zeros(r.out * kk, dtype=self.dtype, device=dev)         # [cinp][f][f][out]
                r.g_conv = torch.zeros(r.out * kk, dtype=torch.float32, device=dev)
            return r

        def ln(prefix, m):
            r = _LN()
            r.C, r.eps = m.normalized_shape[0], m.eps
            wo, bo = offs[prefix + ".weight"], offs[prefix + ".bias"]
            r.gamma, r.beta = self.master[wo:wo + r.C], self.master[bo:bo + r.C]
            r.dgamma, r.dbeta = self.grad[wo:wo + r.C], self.grad[bo:bo + r.C]
            return r

        self.stages = []
        cin = model.stage1.patch_partition.linear.in_features // (model.stage1.patch_partition.downscaling_factor ** 2)
        self.in_channels = cin
        for si in range(1, 5):
            st = getattr(model, f"stage{si}")
            pm = st.patch_partition
            rec = {"pm": lin(f"stage{si}.patch_partition.linear", pm.linear, conv_f=pm.downscaling_factor, cin=cin),
                   "f": pm.downscaling_factor, "blocks": [], "off": offs[f"stage{si}.patch_partition.linear.weight"]}
            for li, pair in enumerate(st.layers):
                for bi, blk in enumerate(pair):
                    pre = f"stage{si}.layers.{li}.{bi}"
                    att = blk.attention_block.fn.fn
                    ff = blk.mlp_block.fn.fn
                    if not att.relative_pos_embedding:
                        raise PfrError("HIP Swin path supports relative_pos_embedding=True (the reference default)")
                    b = {"ln1": ln(pre + ".attention_block.fn.norm", blk.attention_block.fn.norm),
                         "qkv": lin(pre + ".attention_block.fn.fn.to_qkv", att.to_qkv),
                         "out": lin(pre + ".attention_block.fn.fn.to_out", att.to_out),
                         "ln2": ln(pre + ".mlp_block.fn.norm", blk.mlp_block.fn.norm),
                         "fc1": lin(pre + ".mlp_block.fn.fn.net.0", ff.net[0]),
                         "fc2": lin(pre + ".mlp_block.fn.fn.net.2", ff.net[2]),
                         "heads": att.heads, "w": att.window_size, "scale": att.scale,
                         "shift": att.window_size // 2 if att.shifted else 0}
                    po = offs[pre + ".attention_block.fn.fn.pos_embedding"]
                    ntab = (2 * att.window_size - 1) ** 2
                    b["pos"] = self.master[po:po + ntab]
                    b["dpos"] = self.grad[po:po + ntab]
                    b["hd"] = att.to_qkv.out_features // (3 * att.heads)
                    b["tab"] = torch.empty(lib.pfr_window_bias_table_floats(att.window_size), dtype=torch.float32, device=dev)
                    rec["blocks"].append(b)
            self.stages.append(rec)
            cin = pm.linear.out_features
        self.head_ln = ln("mlp_head.0", model.mlp_head[0])
        self.head_fc = lin("mlp_head.1", model.mlp_head[1])
        self.emb_dim = self.head_fc.out
        self.cp = self.stages[0]["pm"].cinp
        self.ws = None
        torch.cuda.synchronize(dev)

    def matches(self, model):
        return id(model) == self.model_id and self.first_param.data.data_ptr() == self.master.data_ptr()

    def attach_grads(self):
        for p, gv in self._views.values():
            p.grad = gv

    def _all_lins(self):
        for st in self.stages:
            yield st["pm"]
            for b in st["blocks"]:
                for k in ("qkv", "out", "fc1", "fc2"):
                    yield b[k]
        yield self.head_fc

    def refresh_weights(self, stream, for_backward=True):
        if self.dtype != torch.float32:
            lib.pfr_cast(self.master.data_ptr(), 0, self.shadow.data_ptr(), self.did, self.n_flat, stream)
        for r in self._all_lins():
            if r.f is not None:   # Unfold order (c, kh, kw) → conv layout [out][kh][kw][c(padded)]
                lib.pfr_nchw_to_nhwc(self.master.data_ptr() + 4 * r.off, r.w.data_ptr(), self.did, r.out, r.cin, r.f * r.f, 1,
                                     r.cinp, stream)
        if not for_backward:
            return
        # data-gradient layouts are first needed by backward(): build them on the side stream, concurrent with forward
        sptr = stream
        use_side = self.side_stream_enabled and _TRACER[0] is None and (self.grad_ready_hook is None or _side_with_ddp())
        if use_side:
            if self.side is None:
                self.side = torch.cuda.Stream(device=self.device)
            if self.wt_fork is None:
                self.wt_fork, self.wt_ready = torch.cuda.Event(), torch.cuda.Event()
            self.wt_fork.record(torch.cuda.current_stream())
            self.side.wait_event(self.wt_fork)
            sptr = self.side.cuda_stream
        # one launch for every layer's data-gradient weights (descriptor table built once: the pointers are fixed)
        tab = getattr(self, "_wt_table", None)
        if tab is None:
            import struct
            recs = []
            for r in self._all_lins():
                if r.f is not None:
                    if r is not self.stages[0]["pm"]:
                        recs.append((r.w.data_ptr(), r.wt.data_ptr(), r.out, r.f, r.f, r.cinp))
                else:
                    recs.append((r.w.data_ptr(), r.wt.data_ptr(), r.out, 1, 1, r.inp))
            raw = b"".join(struct.pack("<QQiiii", *rec) for rec in recs)
            tab = self._wt_table = (torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device), len(recs))
        lib.pfr_weight_dgrad_layout_batch(tab[0].data_ptr(), tab[1], self.did, sptr)
        self.wt_pending = use_side
        if use_side:
            self.wt_ready.record(self.side)

    # ------------------------------------------------------------------------------------------ plan
    def build_plan(self, N, H, W, with_backward):
        T, dev, did = self.dtype, self.device, self.did
        fwd, bwd = [], []
        bufs = []
        tab_recs = []

        def A(shape, dtype=None):
            t = torch.empty(shape, dtype=dtype or T, device=dev)
            bufs.append(t)
            return t

        def gemm(ops, x, rows, cin, r, y, bias=True, residual=None, w=None):
            ops.append((lib.pfr_conv2d_fwd, (x.data_ptr(), (w if w is not None else r.w).data_ptr(), y.data_ptr(), did,
                                             dtype_id(y.dtype), rows, 1, 1, cin, r.out, 1, 1, 1, 0, 0, 1, 1, r.out,
                                             r.bias.data_ptr() if (bias and r.bias is not None) else 0,
                                             0 if residual is None else residual.data_ptr(), 0, 0, 0, 0, 0, 0)))

        ws_need = [0]

        def wgrad(ops, x, xshape, dy, dyshape, r, R, stride, out):
            Nq, Hq, Wq, Cq = xshape
            _, OH, OW, Co = dyshape
            KK = R * R * Cq
            splits = lib.pfr_conv2d_wgrad_splits(Nq * OH * OW, Co, KK)
            ws_need[0] = max(ws_need[0], splits * Co * KK)
            side(ops, ("wgrad", (x.data_ptr(), dy.data_ptr(), out.data_ptr(), None, did, Nq, Hq, Wq, Cq, Co, R, R, stride, 0,
                                OH, OW, Co, 0, 0, 0, 1.0, 0)), dy)

        def dgrad_lin(ops, dy, rows, r, dx):
            ops.append((lib.pfr_conv2d_fwd, (dy.data_ptr(), r.wt.data_ptr(), dx.data_ptr(), did, did, rows, 1, 1, r.out, r.inp, 1,
                                             1, 1, 0, 0, 1, 1, r.inp, 0, 0, 0, 0, 0, 0, 0, 0)))

        cs_need = [0]

        pend_cs = []   # column sums whose final merge is deferred to the next flush: (partials, out, partial rows, C, tile height | 0, rows)

        def colsum(ops, x, rows, C, out, dt=None):
            """bias / LayerNorm-parameter / position-table gradient = column sum of x; the partial sums run now (side stream), the
            ~100 tiny final merges of a step are batched into one launch per DDP bucket boundary (flush_colsums)"""
            d = dt if dt is not None else did
            n = lib.pfr_colsum_parts(d, rows, C)
            if n <= 0:   # a few hundred rows: one small kernel does it all
                side(ops, ("side", (lib.pfr_colsum, (x.data_ptr(), d, rows, C, out.data_ptr(), 0, 0))), x)
                return
            ws = A((lib.pfr_colsum_ws_floats(rows, C),), torch.float32)   # its own workspace: alive until the batched final
            side(ops, ("side", (lib.pfr_colsum_partial, (x.data_ptr(), d, rows, C, ws.data_ptr()))), x)
            pend_cs.append((ws, out, n, C, 0, 0))

        def flush_colsums(ops):
            if not pend_cs:
                return
            import struct
            raw = b"".join(struct.pack("<QQiiiiii", ws.data_ptr(), out.data_ptr(), n, C, 0, mt, rws, 0) for ws, out, n, C, mt, rws in pend_cs)
            tab = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
            bufs.append(tab)
            side(ops, ("side", (lib.pfr_colsum_final_batch, (tab.data_ptr(), len(pend_cs), max(e[3] for e in pend_cs)))))
            del pend_cs[:]

        x_nhwc = A((N, H, W, self.cp))
        cur, cshape = x_nhwc, (N, H, W, self.cp)
        saved = []
        for st in self.stages:
            f, pm = st["f"], st["pm"]
            Nn, Hh, Ww, Cc = cshape
            OH, OW = Hh // f, Ww // f
            rows = N * OH * OW
            C = pm.out
            t = A((N, OH, OW, C))
            fwd.append((lib.pfr_conv2d_fwd, (cur.data_ptr(), pm.w.data_ptr(), t.data_ptr(), did, did, N, Hh, Ww, Cc, C, f, f, f, 0,
                                             0, OH, OW, C, pm.bias.data_ptr() if pm.bias is not None else 0, 0, 0, 0, 0, 0, 0, 0)))
            srec = {"in": cur, "inshape": cshape, "t": t, "shape": (N, OH, OW, C), "blocks": []}
            x = t
            for b in st["blocks"]:
                ln1 = A((rows, C)); mu1 = A((rows,), torch.float32); rs1 = A((rows,), torch.float32)
                fwd.append((lib.pfr_layernorm_fwd, (x.data_ptr(), b["ln1"].gamma.data_ptr(), b["ln1"].beta.data_ptr(), ln1.data_ptr(),
                                                    mu1.data_ptr(), rs1.data_ptr(), did, rows, C, float(b["ln1"].eps))))
                qkv = A((rows, 3 * C))
                gemm(fwd, ln1, rows, C, b["qkv"], qkv)
                att = A((rows, C))
                tab_recs.append((b["pos"].data_ptr(), b["tab"].data_ptr(), b["w"], b["shift"]))
                fwd.append((lib.pfr_window_attn_fwd, (qkv.data_ptr(), b["tab"].data_ptr(), att.data_ptr(), did, N, OH, OW,
                                                      b["heads"], b["hd"], b["w"], b["shift"], float(b["scale"]))))
                y = A((rows, C))
                gemm(fwd, att, rows, C, b["out"], y, residual=x)
                ln2 = A((rows, C)); mu2 = A((rows,), torch.float32); rs2 = A((rows,), torch.float32)
                fwd.append((lib.pfr_layernorm_fwd, (y.data_ptr(), b["ln2"].gamma.data_ptr(), b["ln2"].beta.data_ptr(), ln2.data_ptr(),
                                                    mu2.data_ptr(), rs2.data_ptr(), did, rows, C, float(b["ln2"].eps))))
                h1 = A((rows, 4 * C))
                h2 = A((rows, 4 * C))
                if self.fuse_gelu:   # GELU in the fc1 GEMM's epilogue (writes the pre-activation h1 and h2 = gelu(h1))
                    fwd.append((lib.pfr_gemm_act, (ln2.data_ptr(), b["fc1"].w.data_ptr(), h2.data_ptr(), did, rows, C, 4 * C,
                                                   b["fc1"].bias.data_ptr(), 2, h1.data_ptr())))
                else:
                    gemm(fwd, ln2, rows, C, b["fc1"], h1)
                    fwd.append((lib.pfr_gelu_fwd, (h1.data_ptr(), h2.data_ptr(), did, rows * 4 * C)))
                z = A((rows, C))
                gemm(fwd, h2, rows, 4 * C, b["fc2"], z, residual=y)
                srec["blocks"].append(dict(x=x, ln1=ln1, mu1=mu1, rs1=rs1, qkv=qkv, att=att, y=y, ln2=ln2, mu2=mu2, rs2=rs2,
                                           h1=h1, h2=h2, z=z))
                x = z
            saved.append(srec)
            cur, cshape = x, (N, OH, OW, C)
        Nn, Hh, Ww, Cf = cshape
        pooled = A((N, Cf))
        fwd.append((lib.pfr_avgpool_fwd, (cur.data_ptr(), pooled.data_ptr(), did, N, Hh * Ww, Cf)))
        hln = A((N, Cf)); hmu = A((N,), torch.float32); hrs = A((N,), torch.float32)
        fwd.append((lib.pfr_layernorm_fwd, (pooled.data_ptr(), self.head_ln.gamma.data_ptr(), self.head_ln.beta.data_ptr(),
                                            hln.data_ptr(), hmu.data_ptr(), hrs.data_ptr(), did, N, Cf, float(self.head_ln.eps))))
        emb = A((N, self.emb_dim), torch.float32)
        gemm(fwd, hln, N, Cf, self.head_fc, emb)
        # the bias(+mask) tables of every attention block in one launch at the head of the forward pass (their inputs, the relative-position
        # tables, only change in the optimiser step)
        import struct
        tabd = torch.frombuffer(bytearray(b"".join(struct.pack("<QQii", *r) for r in tab_recs)), dtype=torch.uint8).to(dev)
        bufs.append(tabd)
        fwd.insert(0, (lib.pfr_window_bias_table_batch, (tabd.data_ptr(), len(tab_recs))))
        plan = {"fwd": fwd, "bufs": bufs, "x_nhwc": x_nhwc, "emb": emb}
        if not with_backward:
            return plan

        # ================================================================= backward
        # Weight gradients, bias / LayerNorm-parameter / position-table column sums feed nothing before the optimizer: they run on
        # a SIDE stream (≈ 150 short launches per step that would otherwise sit in the dependency chain).  Symbolic ops:
        # ("fork", k) side waits for main's current point; ("srec", k) side records "op k done"; ("wait", k) main waits
        # for op k — emitted before a pooled buffer that op k read is handed out again, before grad-ready marks and at the end.
        pool = {}
        nalloc = {}
        pending = {}      # data_ptr of a pooled buffer -> last side op that reads it
        side_reads = []   # (k, data_ptr) of every side-op input
        nside = [0]

        def G(shape, dtype=None):
            key = (tuple(shape), dtype or T)
            lst = pool.setdefault(key, [])
            for i, t in enumerate(lst):              # a buffer no side op is reading
                if t.data_ptr() not in pending:
                    return lst.pop(i)
            if not lst or nalloc.get(key, 0) < self.pool_depth:    # rotation so that the side stream may lag behind without stalling main
                nalloc[key] = nalloc.get(key, 0) + 1
                return A(shape, dtype)
            t = lst.pop(0)
            bwd.append(("wait", (pending.pop(t.data_ptr()),)))
            return t

        def release(t):
            lo, hi = t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()
            ks = [k for k, ptr in side_reads if lo <= ptr < hi]
            if ks:
                pending[t.data_ptr()] = max(ks)
            side_reads[:] = [(k, ptr) for k, ptr in side_reads if not (lo <= ptr < hi)]
            pool.setdefault((tuple(t.shape), t.dtype), []).append(t)

        def side(ops, op, *reads):
            if ops and ops[-1][0] == "srec":
                # no main-stream launch since the previous side op: it joins that op's fork (every fork is an event recorded on the
                # main stream, i.e. a marker packet in front of the next kernel of the dependency chain)
                k = ops.pop()[1][0]
            else:
                k = nside[0]
                nside[0] += 1
                ops.append(("fork", (k,)))
            ops.append(op)
            ops.append(("srec", (k,)))
            for r in reads:
                side_reads.append((k, r.data_ptr()))

        def ln_bwd(ops, dy, xin, mu, rs, lnrec, dres, dx, rows, C, want_sum=False):
            """→ (partials, rows of partials) of the column sums of dx when want_sum and the kernel can emit them, else None"""
            nb = lib.pfr_layernorm_bwd_blocks(rows)
            # the kernel's per-workgroup partials ARE the partial sets of the deferred final merge (own buffer per LayerNorm: they
            # must survive until the batched merge at the next bucket boundary)
            part = A((2, nb, C), torch.float32)
            dsum = A((nb, C), torch.float32) if (want_sum and self.ln_dxsum and lib.pfr_layernorm_bwd_dxsum_ok(did, C)) else None
            ops.append((lib.pfr_layernorm_bwd_dxsum, (dy.data_ptr(), xin.data_ptr(), mu.data_ptr(), rs.data_ptr(), lnrec.gamma.data_ptr(),
                                                      0 if dres is None else dres.data_ptr(), dx.data_ptr(), part.data_ptr(),
                                                      0 if dsum is None else dsum.data_ptr(), did, rows, C)))
            pend_cs.append((part[0], lnrec.dgamma, nb, C, 0, 0))
            pend_cs.append((part[1], lnrec.dbeta, nb, C, 0, 0))
            return None if dsum is None else (dsum, nb)

        def bias_grad(ops, g, rows, C, dbias, gsum):
            """bias gradient = column sum of g: from the LayerNorm-backward pass that produced g when it left the sums (gsum), else
            a column-sum pass of its own on the side stream"""
            if gsum is not None:
                pend_cs.append((gsum[0], dbias, gsum[1], C, 0, 0))
            else:
                colsum(ops, g, rows, C, dbias)

        demb = A((N, self.emb_dim))
        plan["demb"] = demb
        hf = self.head_fc
        if hf.dbias is not None:
            colsum(bwd, demb, N, hf.out, hf.dbias)
        wgrad(bwd, hln, (N, 1, 1, Cf), demb, (N, 1, 1, hf.out), hf, 1, 1, hf.g)
        dhln = G((N, Cf))
        dgrad_lin(bwd, demb, N, hf, dhln)
        dpooled = G((N, Cf))
        ln_bwd(bwd, dhln, pooled, hmu, hrs, self.head_ln, None, dpooled, N, Cf)
        dz_sum = None   # column sums of dz left by the pass that produced it (None: avgpool / patch-merging data gradient)
        release(dhln)
        dz = G(cshape)
        bwd.append((lib.pfr_avgpool_bwd, (dpooled.data_ptr(), dz.data_ptr(), did, N, Hh * Ww, Cf)))
        release(dpooled)
        bwd.append((None, (offs_head := self.offs["mlp_head.0.weight"],)))
        for si in range(len(self.stages) - 1, -1, -1):
            st, srec = self.stages[si], saved[si]
            Nn, OH, OW, C = srec["shape"]
            rows = N * OH * OW
            for b, sv in zip(reversed(st["blocks"]), reversed(srec["blocks"])):
                # ---- MLP branch: z = fc2(gelu(fc1(ln2(y)))) + y
                bias_grad(bwd, dz.view(rows, C), rows, C, b["fc2"].dbias, dz_sum)
                wgrad(bwd, sv["h2"], (rows, 1, 1, 4 * C), dz, (rows, 1, 1, C), b["fc2"], 1, 1, b["fc2"].g)
                dh2 = G((rows, 4 * C))
                if self.fuse_gelu:   # dh1 = (dz·W2) ∘ gelu'(h1) in the data-gradient GEMM's epilogue
                    # ... which also leaves the per-m-tile column means of dh1: fc1's bias gradient = sum_t rows_t * mean_t, merged
                    # with the other deferred column sums -- no second pass over the widest gradient tensor of the block
                    nsum = lib.pfr_gemm_act_colsum_parts(rows, C, 4 * C, did)
                    if nsum > 0:     # streaming Linear kernel (csrc/pfr_slin.hip): plain column sums per row range
                        stp = A((nsum, 4 * C), torch.float32)
                        bwd.append((lib.pfr_gemm_act_colsums, (dz.data_ptr(), b["fc2"].wt.data_ptr(), dh2.data_ptr(), did, rows, C, 4 * C,
                                                               sv["h1"].data_ptr(), stp.data_ptr())))
                        pend_cs.append((stp, b["fc1"].dbias, nsum, 4 * C, 0, 0))
                    else:
                        mt = lib.pfr_gemm_act_mtile(rows, C, 4 * C, did)
                        nt = (rows + mt - 1) // mt
                        stp = A((nt, 2, 4 * C), torch.float32)
                        bwd.append((lib.pfr_gemm_act_colstats, (dz.data_ptr(), b["fc2"].wt.data_ptr(), dh2.data_ptr(), did, rows, C, 4 * C, 0,
                                                                3, sv["h1"].data_ptr(), stp.data_ptr())))
                        pend_cs.append((stp, b["fc1"].dbias, nt, 4 * C, mt, rows))
                else:
                    dgrad_lin(bwd, dz, rows, b["fc2"], dh2)
                    bwd.append((lib.pfr_gelu_bwd, (sv["h1"].data_ptr(), dh2.data_ptr(), dh2.data_ptr(), did, rows * 4 * C)))
                    colsum(bwd, dh2, rows, 4 * C, b["fc1"].dbias)
                wgrad(bwd, sv["ln2"], (rows, 1, 1, C), dh2, (rows, 1, 1, 4 * C), b["fc1"], 1, 1, b["fc1"].g)
                dln2 = G((rows, C))
                dgrad_lin(bwd, dh2, rows, b["fc1"], dln2)
                release(dh2)
                dy = G((rows, C))
                dy_sum = ln_bwd(bwd, dln2, sv["y"], sv["mu2"], sv["rs2"], b["ln2"], dz, dy, rows, C, want_sum=True)
                release(dln2)
                release(dz)
                # ---- attention branch: y = to_out(attn(to_qkv(ln1(x)))) + x
                bias_grad(bwd, dy, rows, C, b["out"].dbias, dy_sum)
                wgrad(bwd, sv["att"], (rows, 1, 1, C), dy, (rows, 1, 1, C), b["out"], 1, 1, b["out"].g)
                datt = G((rows, C))
                dgrad_lin(bwd, dy, rows, b["out"], datt)
                dqkv = G((rows, 3 * C))
                nblk = N * (OH // b["w"]) * (OW // b["w"]) * b["heads"]
                ntab = (2 * b["w"] - 1) ** 2
                dpart = G((nblk, ntab), torch.float32)
                bwd.append((lib.pfr_window_attn_bwd, (sv["qkv"].data_ptr(), b["tab"].data_ptr(), datt.data_ptr(), dqkv.data_ptr(),
                                                      dpart.data_ptr(), did, N, OH, OW, b["heads"], b["hd"], b["w"], b["shift"],
                                                      float(b["scale"]))))
                colsum(bwd, dpart, nblk, ntab, b["dpos"], 0)
                release(dpart)
                release(datt)
                wgrad(bwd, sv["ln1"], (rows, 1, 1, C), dqkv, (rows, 1, 1, 3 * C), b["qkv"], 1, 1, b["qkv"].g)
                dln1 = G((rows, C))
                dgrad_lin(bwd, dqkv, rows, b["qkv"], dln1)
                release(dqkv)
                dx = G((rows, C))
                dz_sum = ln_bwd(bwd, dln1, sv["x"], sv["mu1"], sv["rs1"], b["ln1"], dy, dx, rows, C, want_sum=True)
                release(dln1)
                release(dy)
                dz = dx
            # ---- patch merging (stride-f conv): bias, weight gradient (conv layout → Unfold layout), data gradient
            pm, f = st["pm"], st["f"]
            Ni, Hi, Wi, Ci = srec["inshape"]
            if pm.dbias is not None:
                bias_grad(bwd, dz, rows, C, pm.dbias, dz_sum)
            g_conv = pm.g_conv
            wgrad(bwd, srec["in"], (Ni, Hi, Wi, Ci), dz, (N, OH, OW, C), pm, f, f, g_conv)
            side(bwd, ("side", (lib.pfr_nhwc_to_nchw_f32, (g_conv.data_ptr(), pm.g.data_ptr(), pm.out, pm.cin, f * f, pm.cinp, 0))))
            if si > 0:
                din = G((Ni, Hi, Wi, Ci))
                bwd.append((lib.pfr_conv2d_fwd, (dz.data_ptr(), pm.wt.data_ptr(), din.data_ptr(), did, did, N, OH, OW, C, Ci, f, f, 1,
                                                 f - 1, {2: 1, 4: 2}[f], Hi, Wi, Ci, 0, 0, 0, 0, 0, 0, 0, 0)))
                release(dz)
                dz = din
            dz_sum = None
            flush_colsums(bwd)
            if nside[0]:   # everything the side stream was given so far is final (end of backward, or a DDP bucket boundary)
                bwd.append(("wait" if si == 0 else "mwait", (nside[0] - 1,)))
            bwd.append((None, (st["off"],)))
        plan["n_side"] = nside[0]
        if self.ws is None or self.ws.numel() < ws_need[0]:
            self.ws = torch.empty(ws_need[0], dtype=torch.float32, device=dev)
        if getattr(self, "cs_ws", None) is None or self.cs_ws.numel() < cs_need[0]:
            self.cs_ws = torch.empty(max(1, cs_need[0]), dtype=torch.float32, device=dev)
        plan["bwd_sym"] = bwd
        return plan

    def _finalize(self, plan):
        res = []
        for fn, args in plan["bwd_sym"]:
            if fn == "wgrad":
                a = list(args)
                a[3] = self.ws.data_ptr()
                res.append((_SIDE, (lib.pfr_conv2d_wgrad, tuple(a))))
            elif fn == "colsum":
                res.append((_SIDE, (lib.pfr_colsum, tuple(args[:-1]) + (self.cs_ws.data_ptr(),))))
            elif fn == "side":
                res.append((_SIDE, args))
            elif fn == "fork":
                res.append((_FORK, args[0]))
            elif fn == "srec":
                res.append((_SREC, args[0]))
            elif fn == "wait":
                res.append((_WAIT, args[0]))
            elif fn == "mwait":
                res.append((_MWAIT, args[0]))
            else:
                res.append((fn, args))
        plan["bwd"] = res
        plan.pop("c_bwd", None)
        plan["ws_ptr"] = (self.ws.data_ptr(), self.cs_ws.data_ptr())

    def get_plan(self, N, H, W, with_backward, slot=0):
        ep = lib.pfr_tuning_epoch()
        if ep != getattr(self, "_tuning_epoch", None):
            # a pfr_set_tuning call changed a knob: the plans baked tile heights / kernel choices in — rebuild those not in flight
            if hasattr(self, "_tuning_epoch"):
                for k, q in list(self.plans.items()):
                    if not self._plan_busy(q):
                        self.plans.pop(k)
            self._tuning_epoch = ep
        key = (N, H, W, with_backward) + ((slot,) if slot else ())
        p = self.plans.get(key)
        if p is None:
            if len(self.plans) >= 6:
                for k, q in list(self.plans.items()):
                    if not self._plan_busy(q):
                        self.plans.pop(k)
                        break
            p = self.build_plan(N, H, W, with_backward)
            if with_backward:
                self._finalize(p)
            self.plans[key] = p
        elif with_backward and p["ws_ptr"] != (self.ws.data_ptr(), self.cs_ws.data_ptr()):
            self._finalize(p)
        return p

    @staticmethod
    def _plan_busy(plan):
        own = plan.get("owner")
        return own is not None and own() is not None

    def acquire_plan(self, N, H, W, with_backward, ticket):
        """one plan instance ("slot") per forward pass that still waits for its backward — see FEEngine.acquire_plan"""
        slot = 0
        while True:
            plan = self.get_plan(N, H, W, with_backward, slot)
            if ticket is None or not self._plan_busy(plan):
                break
            slot += 1
            if slot >= 8:
                raise PfrError("more than 8 forward passes of one shape are waiting for their backward pass")
        if ticket is not None:
            plan["owner"] = weakref.ref(ticket)
        return plan

    def forward(self, x, with_backward, ticket=None):
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise PfrError(f"expected NCHW input with {self.in_channels} channels, got {tuple(x.shape)}")
        x = x.float().contiguous()
        N, _, H, W = x.shape
        plan = self.acquire_plan(N, H, W, with_backward, ticket if with_backward else None)
        stream = torch.cuda.current_stream().cuda_stream
        self.refresh_weights(stream, for_backward=with_backward)
        lib.pfr_nchw_to_nhwc(x.data_ptr(), plan["x_nhwc"].data_ptr(), self.did, N, x.shape[1], H, W, self.cp, stream)
        if not self._run_c(plan, "fwd", stream):
            for fn, args in plan["fwd"]:
                fn(*args, stream)
        self._last = plan
        return plan["emb"]

    def _run_c(self, plan, key, stream, side=0, hook=None, n_events=0):
        """replay plan[key] through the C executor when possible (same contract as FEEngine._run_list)"""
        if not self.c_plan or _TRACER[0] is not None:
            return False
        ck = "c_" + key
        cp = plan.get(ck, False)
        if cp is False:
            cp = plan[ck] = CPlan.compile(plan[key], n_events)
        if cp is None:
            return False
        cp.run(stream, side, hook, self.hook_syncs_side)
        return True

    def backward(self, demb, plan=None):
        plan = plan if plan is not None else self._last
        stream = torch.cuda.current_stream().cuda_stream
        demb = demb.contiguous()
        if demb.numel() != plan["demb"].numel():
            raise PfrError(f"backward: gradient of {tuple(demb.shape)} does not match the plan's embedding buffer "
                           f"{tuple(plan['demb'].shape)}")
        if plan["ws_ptr"] != (self.ws.data_ptr(), self.cs_ws.data_ptr()):
            self._finalize(plan)
        # Every gradient kernel of this engine OVERWRITES its slice of the flat buffer.  When gradients are already present
        # (a second backward before zero_grad: gradient accumulation, list inputs) the previous sum is set aside and added
        # back afterwards — two extra passes over the 110 MB buffer, only on that path.
        prev = self.grad.clone() if self.first_param.grad is not None else None
        plan["owner"] = None
        lib.pfr_cast(demb.data_ptr(), dtype_id(demb.dtype), plan["demb"].data_ptr(), self.did, demb.numel(), stream)
        hook = self.grad_ready_hook
        if prev is not None or any(self._plan_busy(q) for q in self.plans.values()):
            hook = None   # not final yet (accumulating, or another forward pass still waits for its backward)
        main = torch.cuda.current_stream()
        if self.wt_pending:
            main.wait_event(self.wt_ready)
            self.wt_pending = False
        # side stream off: PFR_SIDE_STREAM=0, a launch tracer is active, or gradients are all-reduced (see FEEngine._side_ok)
        use_side = self.side_stream_enabled and _TRACER[0] is None and (hook is None or _side_with_ddp())
        if use_side and self.side is None:
            self.side = torch.cuda.Stream(device=self.device)
        if self._run_c(plan, "bwd", stream, self.side.cuda_stream if use_side else 0,
                       (lambda off: hook(off)) if hook is not None else None, 2 * plan.get("n_side", 0)):
            if prev is not None:
                self.grad.add_(prev)
            self.attach_grads()
            return
        if use_side:
            side, sptr = self.side, self.side.cuda_stream
            ev = self.side_events
            while len(ev) < 2 * plan.get("n_side", 0):
                ev.append(torch.cuda.Event())
        for fn, args in plan["bwd"]:
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
        if prev is not None:
            self.grad.add_(prev)
        self.attach_grads()


class _SwinFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, *params):
        eng = model.hip_engine(x.device)
        ctx.ticket = PlanTicket()
        emb = eng.forward(x, True, ctx.ticket)
        ctx.eng = eng
        ctx.plan = eng._last
        ctx.nparams = len(params)
        return emb.clone()

    @staticmethod
    def backward(ctx, demb):
        if ctx.plan.get("owner") is None or ctx.plan["owner"]() is not ctx.ticket:
            raise PfrError("backward: the activations of this forward pass were released (double backward?)")
        ctx.eng.backward(demb, ctx.plan)
        return (None, None) + (None,) * ctx.nparams


def swin_forward(model, x):
    eng = model.hip_engine(x.device)
    if torch.is_grad_enabled() and any(p.requires_grad for p in eng.param_list):
        return _SwinFunction.apply(x, model, *eng.param_list)
    return eng.forward(x, False).clone()
