"""Flat-buffer fused optimizers with the torch.optim interface (param_groups / step / zero_grad / state_dict), so the
config contract `optimizer(model_) -> ([optim], [sched])` (/root/reference/configs/dog_fe/fe_dogs_config.py:123-133,
body_dog_fe.py:121-131) and torch LR schedulers keep working.  Parameters that live in an FEEngine flat buffer are
updated with ONE kernel launch per contiguous range of a param group; any other CUDA parameter gets one launch."""
import torch

from .._hip import ops, PfrError


class _FusedBase(torch.optim.Optimizer):
    """State lives in `self.state[p]` like in torch.optim (so `state_dict()` / `load_state_dict()` round-trip and follow the
    parameters through `model.to()`): each entry is a VIEW, with the parameter's logical shape and strides, into one flat
    buffer per contiguous run of the group, which is what the kernels update."""
    _STATE_KEYS = ()

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._runs = {}      # group index -> (signature, [run, ...])

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._runs = {}      # loaded tensors are private copies: re-pack them into flat run buffers at the next step

    def _group_runs(self, gi, group):
        params = [p for p in group["params"] if p.grad is not None]
        # the state layout follows the PARAMETERS; a gradient that autograd re-allocated (the ArcFace head's, 20 MB) only moves
        # the run's gradient view — it must not rebuild (re-allocate and copy) every state buffer of the group
        sig = tuple((id(p), p.data_ptr()) for p in params)
        gsig = tuple(p.grad.data_ptr() for p in params)
        cached = self._runs.get(gi)
        if cached is not None and cached[0] == sig:
            if cached[2] != gsig:
                if not self._rebind_grads(cached[1]):
                    cached = None
                else:
                    self._runs[gi] = (sig, cached[1], gsig)
            if cached is not None:
                return cached[1]
        if any(not p.is_cuda for p in params):
            raise PfrError("fused optimizers need CUDA parameters (use torch.optim on the CPU path)")
        runs = self._build_runs(params)
        for r in runs:
            n = r["n"]
            dev = r["pf"].device
            r["state"] = {k: torch.zeros(n, dtype=torch.float32, device=dev) for k in self._STATE_KEYS}
            for p, off in r["members"]:
                st = self.state[p]
                for k in self._STATE_KEYS:
                    view = torch.as_strided(r["state"][k], p.shape, p.stride(), off)
                    old = st.get(k)
                    if old is not None:
                        view.copy_(old.to(dev))     # carried over (loaded checkpoint, or a previous buffer layout)
                    st[k] = view
        self._runs[gi] = (sig, runs, gsig)
        return runs

    def _rebind_grads(self, runs):
        """gradient tensors moved: re-point the flat gradient view of every run (False: the members' gradients no longer form
        the same dense run, rebuild everything)"""
        for r in runs:
            p0, _ = r["members"][0]
            sp = self._storage_span(p0.grad)
            if sp is None:
                return False
            gs = sp[0]
            for p, off in r["members"]:
                s2 = self._storage_span(p.grad)
                if s2 is None or s2[0] != gs + 4 * off or p.grad.untyped_storage().data_ptr() != p0.grad.untyped_storage().data_ptr():
                    return False
            goff = (gs - p0.grad.untyped_storage().data_ptr()) // 4
            r["gs"], r["ge"], r["gbase"] = gs, gs + 4 * r["n"], p0.grad.untyped_storage().data_ptr()
            r["gf"] = torch.empty(0, dtype=torch.float32, device=p0.device).set_(p0.grad.untyped_storage(), goff, (r["n"],), (1,))
        return True

    def _build_runs(self, params):
        """→ runs covering `params` with as few flat tensors as possible.  Neighbours are merged only across the engine's
        ALIGNMENT PADDING (< 64 floats): a gap of a whole 64-element parameter (e.g. a BN bias of another param group lying
        between two weights) must not be swallowed into this group's update."""
        dense = []
        for p in params:
            if p.dtype != torch.float32:
                raise PfrError("fused optimizers need fp32 parameters")
            st_p = self._storage_span(p.data)
            st_g = self._storage_span(p.grad)
            if st_p is None or st_g is None:
                raise PfrError("fused optimizers need parameters/gradients that are dense in memory")
            dense.append((st_p, st_g, p))
        dense.sort(key=lambda t: t[0][0])
        runs = []
        for (pa, pn), (ga, gn), p in dense:
            r = runs[-1] if runs else None
            # merge only across the engine's alignment padding of the PREVIOUS member ((-numel) % 64 floats): a small parameter
            # of another group (or a frozen one) lying in a larger gap must not be swallowed into this group's update
            if r is not None and pa - r["pe"] == 4 * ((-r["last_n"]) % 64) and (ga - r["ge"]) == (pa - r["pe"]) \
                    and r["pbase"] == p.data.untyped_storage().data_ptr() \
                    and r["gbase"] == p.grad.untyped_storage().data_ptr():
                r["members"].append((p, (pa - r["ps"]) // 4))
                r["pe"] = pa + 4 * pn
                r["ge"] = ga + 4 * pn
                r["last_n"] = pn
            else:
                runs.append({"ps": pa, "pe": pa + 4 * pn, "gs": ga, "ge": ga + 4 * pn, "p": p, "members": [(p, 0)], "last_n": pn,
                             "pbase": p.data.untyped_storage().data_ptr(), "gbase": p.grad.untyped_storage().data_ptr()})
        for r in runs:
            n = (r["pe"] - r["ps"]) // 4
            p = r["p"]
            poff = (r["ps"] - r["pbase"]) // 4
            goff = (r["gs"] - r["gbase"]) // 4
            r["n"] = n
            r["pf"] = torch.empty(0, dtype=torch.float32, device=p.device).set_(p.data.untyped_storage(), poff, (n,), (1,))
            r["gf"] = torch.empty(0, dtype=torch.float32, device=p.device).set_(p.grad.untyped_storage(), goff, (n,), (1,))
        return runs

    def _flat_views(self, group):
        """(param_flat, grad_flat, (address, numel)) per run — kept for tests / introspection"""
        return [(r["pf"], r["gf"], (r["ps"], r["n"])) for r in self._build_runs([p for p in group["params"] if p.grad is not None])]

    @staticmethod
    def _storage_span(t):
        """(address, numel) if t occupies one dense block of memory (any permutation of a contiguous tensor)."""
        n = t.numel()
        if n == 0:
            return None
        sizes, strides = list(t.shape), list(t.stride())
        order = sorted(range(len(sizes)), key=lambda i: -strides[i])
        expect = 1
        for i in reversed(order):
            if sizes[i] != 1 and strides[i] != expect:
                return None
            expect *= sizes[i]
        return (t.data_ptr(), n)


class FusedSGD(_FusedBase):
    _STATE_KEYS = ("momentum_buffer",)

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0, weight_decay=0.0, nesterov=False):
        if dampening != 0 or nesterov:
            raise PfrError("FusedSGD supports dampening=0, nesterov=False (what the reference configs use)")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            for r in self._group_runs(gi, group):
                # a zero-initialised buffer makes torch's "first step: buf = d" the general rule buf = momentum*buf + d
                buf = r["state"]["momentum_buffer"] if group["momentum"] != 0 else None
                ops.sgd_step(r["pf"], r["gf"], buf, None, group["lr"], group["momentum"], group["weight_decay"], first_step=False)
        return loss


class FusedAdamW(_FusedBase):
    _STATE_KEYS = ("exp_avg", "exp_avg_sq")

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._t = 0

    def state_dict(self):
        for group in self.param_groups:
            for p in group["params"]:
                if p in self.state:
                    self.state[p]["step"] = torch.tensor(float(self._t))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = [float(st["step"]) for st in self.state.values() if "step" in st]
        self._t = int(max(steps)) if steps else 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._t += 1
        for gi, group in enumerate(self.param_groups):
            for r in self._group_runs(gi, group):
                ops.adamw_step(r["pf"], r["gf"], r["state"]["exp_avg"], r["state"]["exp_avg_sq"], None, group["lr"],
                               group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"], self._t)
        return loss
