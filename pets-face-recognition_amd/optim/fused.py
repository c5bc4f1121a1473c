"""Flat-buffer fused optimizers with the torch.optim interface (param_groups / step / zero_grad / state_dict), so the
config contract `optimizer(model_) -> ([optim], [sched])` (/root/reference/configs/dog_fe/fe_dogs_config.py:123-133,
body_dog_fe.py:121-131) and torch LR schedulers keep working.  Parameters that live in an FEEngine flat buffer are
updated with ONE kernel launch per contiguous range of a param group; any other CUDA parameter gets one launch."""
import torch

from .._hip import ops, PfrError


def _segments(params):
    """Group CUDA fp32 parameters into maximal runs that are contiguous in memory (flat-buffer neighbours)."""
    items = []
    for p in params:
        if p.grad is None:
            continue
        if not (p.is_cuda and p.dtype == torch.float32):
            raise PfrError("fused optimizers need CUDA fp32 parameters")
        items.append(p)
    return items


class _FusedBase(torch.optim.Optimizer):
    def _flat_views(self, group):
        """→ list of (param_flat, grad_flat, key) covering the group's parameters with as few tensors as possible."""
        runs = []
        dense = []
        for p in group["params"]:
            if p.grad is None:
                continue
            st_p = self._storage_span(p.data)
            st_g = self._storage_span(p.grad)
            if st_p is None or st_g is None:
                raise PfrError("fused optimizers need parameters/gradients that are dense in memory")
            dense.append((st_p, st_g, p))
        dense.sort(key=lambda t: t[0][0])
        for (pa, pn), (ga, gn), p in dense:
            # extend the previous run when both param and grad continue it (allowing the engine's alignment padding)
            if runs and 0 <= pa - runs[-1]["pe"] <= 256 and (ga - runs[-1]["ge"]) == (pa - runs[-1]["pe"]) \
                    and runs[-1]["pbase"] == p.data.untyped_storage().data_ptr() \
                    and runs[-1]["gbase"] == p.grad.untyped_storage().data_ptr():
                runs[-1]["pe"] = pa + 4 * pn
                runs[-1]["ge"] = ga + 4 * pn
            else:
                runs.append({"ps": pa, "pe": pa + 4 * pn, "gs": ga, "ge": ga + 4 * pn, "p": p,
                             "pbase": p.data.untyped_storage().data_ptr(), "gbase": p.grad.untyped_storage().data_ptr()})
        out = []
        for r in runs:
            n = (r["pe"] - r["ps"]) // 4
            p = r["p"]
            poff = (r["ps"] - r["pbase"]) // 4
            goff = (r["gs"] - r["gbase"]) // 4
            pf = torch.empty(0, dtype=torch.float32, device=p.device).set_(p.data.untyped_storage(), poff, (n,), (1,))
            gf = torch.empty(0, dtype=torch.float32, device=p.device).set_(p.grad.untyped_storage(), goff, (n,), (1,))
            out.append((pf, gf, (r["ps"], n)))
        return out

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
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0, weight_decay=0.0, nesterov=False):
        if dampening != 0 or nesterov:
            raise PfrError("FusedSGD supports dampening=0, nesterov=False (what the reference configs use)")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._bufs = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            for pf, gf, key in self._flat_views(group):
                k = (gi,) + key
                buf = self._bufs.get(k)
                first = buf is None
                if first and group["momentum"] != 0:
                    buf = torch.zeros_like(pf)
                    self._bufs[k] = buf
                ops.sgd_step(pf, gf, buf, None, group["lr"], group["momentum"], group["weight_decay"], first_step=first)
        return loss


class FusedAdamW(_FusedBase):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._bufs = {}
        self._t = 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._t += 1
        for gi, group in enumerate(self.param_groups):
            for pf, gf, key in self._flat_views(group):
                k = (gi,) + key
                st = self._bufs.get(k)
                if st is None:
                    st = (torch.zeros_like(pf), torch.zeros_like(pf))
                    self._bufs[k] = st
                ops.adamw_step(pf, gf, st[0], st[1], None, group["lr"], group["betas"][0], group["betas"][1], group["eps"],
                               group["weight_decay"], self._t)
        return loss
