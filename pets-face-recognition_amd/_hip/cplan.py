"""Python handle of the C-side launch-plan executor (csrc/pfr_plan.hip, include/pfr_hip.h `pfr_plan_*`).

The engines build a step as a list of (C-ABI function, argument tuple) entries with fixed device pointers; `CPlan.compile`
packs such a list once into a C plan and `run` replays it with one foreign call (~0.3 µs of host time per launch instead of
~20 µs through ctypes + the interpreter).  Entries the executor cannot take (no thunk) make compile() return None and the
caller keeps its Python loop."""
import ctypes
import struct
import weakref

from .lib import lib, PfrError

# entry kinds of the engines' backward lists (models/_fe_engine.py) → plan kinds
SIDE, FORK, SREC, WAIT, MWAIT = 1, 2, 3, 4, 5


def _slot(v, ctype):
    if ctype is ctypes.c_float:
        return struct.unpack("<I", struct.pack("<f", float(v)))[0]
    if v is None:
        return 0
    return int(v) & 0xFFFFFFFFFFFFFFFF


class CPlan:
    def __init__(self, handle, hooks):
        self.handle = handle
        self.hooks = hooks          # plan index -> argument tuple of the host callback at that stop
        self._fin = weakref.finalize(self, lib.pfr_plan_destroy, handle)

    @staticmethod
    def compile(ops, n_events=0):
        """ops: list of (fn, args) [main-stream launch], (SIDE, (fn, args)), (FORK|SREC|WAIT|MWAIT, k), (None, hook_args)."""
        protos = lib.symbols()
        h = lib.pfr_plan_create(int(n_events))
        if not h:
            raise PfrError("pfr_plan_create failed")
        hooks = {}
        ok = True
        for fn, args in ops:
            if fn is None:
                hooks[lib.pfr_plan_size(h)] = args
                lib.pfr_plan_append(h, 6, -1, 0, None, 0)
                continue
            if fn.__class__ is int and fn != SIDE:
                k = int(args)
                kind, ev = {FORK: (2, 2 * k), SREC: (3, 2 * k + 1), WAIT: (4, 2 * k + 1), MWAIT: (5, 2 * k + 1)}[fn]
                lib.pfr_plan_append(h, kind, -1, ev, None, 0)
                continue
            kind = 0
            if fn.__class__ is int:
                kind = 1
                fn, args = args
            name = getattr(fn, "__name__", None)
            ti = lib.pfr_plan_thunk_index(name.encode()) if name else -1
            if ti < 0:
                ok = False
                break
            argtypes = protos[name][1][:-1]      # (the trailing stream is supplied by the executor)
            if len(args) != len(argtypes):
                raise PfrError(f"{name}: {len(args)} arguments for {len(argtypes)} parameters")
            arr = (ctypes.c_ulonglong * len(args))(*[_slot(v, t) for v, t in zip(args, argtypes)])
            lib.pfr_plan_append(h, kind, ti, 0, arr, len(args))
        if not ok:
            lib.pfr_plan_destroy(h)
            return None
        return CPlan(h, hooks)

    def run(self, main_stream, side_stream=0, hook=None, hook_syncs_side=False):
        pos = 0
        while True:
            r = lib.pfr_plan_run(self.handle, pos, -1, main_stream, side_stream or 0, (2 if hook_syncs_side else 1) if hook is not None else 0)
            if r == -1:
                return
            if r <= -2:
                raise PfrError(f"pfr_plan_run failed ({r}): {lib.pfr_last_error().decode()}")
            hook(*self.hooks[r])
            pos = r + 1
