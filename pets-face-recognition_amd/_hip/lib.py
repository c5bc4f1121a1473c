"""ctypes binding of libpfr_hip.so.

The prototypes are parsed from include/pfr_hip.h, so the header IS the binding: every declared symbol must be
exported by the shared library (checked at load time and by tests/test_abi.py).  There is no fallback: if the
library is missing, any use raises.
"""
import ctypes
import os
import re

PFR_F32, PFR_BF16 = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("PFR_LIB_PATH") or os.path.join(_PKG, "csrc", "libpfr_hip.so")   # (override: profiling builds)
HEADER_PATH = os.path.join(_ROOT, "include", "pfr_hip.h")


class PfrError(RuntimeError):
    pass


_CTYPES = {
    "int": ctypes.c_int,
    "long": ctypes.c_long,
    "size_t": ctypes.c_size_t,
    "float": ctypes.c_float,
    "pfr_stream_t": ctypes.c_void_p,
}


def _ctype_of(decl: str):
    decl = decl.strip()
    if "*" in decl:
        if decl.replace("const", "").strip().startswith("char"):
            return ctypes.c_char_p
        return ctypes.c_void_p
    base = decl.replace("const", "").split()[0]
    return _CTYPES[base]


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes], [argnames])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|void\s*\*|int|long)\s+(pfr_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else (ctypes.c_void_p if "void" in ret else (ctypes.c_long if ret == "long" else ctypes.c_int))
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                nm = re.search(r"(\w+)$", a).group(1)
                argnames.append(nm)
                argtypes.append(_ctype_of(a[: -len(nm)]))
        protos[name] = (restype, argtypes, argnames)
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self._protos = None

    def _load(self):
        if self._dll is not None:
            return
        if not os.path.exists(LIB_PATH):
            raise PfrError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or pets-face-recognition_amd/csrc/build.sh). There is no CPU/PyTorch fallback for the HIP path."
            )
        # torch first: its bundled HIP runtime must be the one this process uses (one runtime, shared streams)
        import torch  # noqa: F401
        self._dll = ctypes.CDLL(LIB_PATH)
        self._protos = parse_header()
        for name, (restype, argtypes, _names) in self._protos.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:
                raise PfrError(f"libpfr_hip.so does not export {name} declared in include/pfr_hip.h") from e
            fn.restype = restype
            fn.argtypes = argtypes
        # start-up values of the library's tuning knobs: PFR_TUNING="wgrad9=2,sconv=0" -> pfr_set_tuning (the library reads no environment)
        for item in filter(None, (os.environ.get("PFR_TUNING") or "").replace(";", ",").split(",")):
            k, _, v = item.partition("=")
            if self._dll.pfr_set_tuning(k.strip().encode(), int(v)) != 0:
                raise PfrError(f"PFR_TUNING: {self._dll.pfr_last_error().decode()}")

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        self._load()
        if name not in self._protos:
            raise AttributeError(f"{name} is not declared in include/pfr_hip.h")
        fn = getattr(self._dll, name)
        if self._protos[name][0] is ctypes.c_int and name not in _NO_CHECK:
            def checked(*args, _fn=fn, _name=name):
                tr = _TRACER[0]
                if tr is not None:
                    tr.before(_name, args)
                rc = _fn(*args)
                if tr is not None:
                    tr.after(_name, args)
                if rc != 0:
                    raise PfrError(f"{_name} failed (rc={rc}): {self._dll.pfr_last_error().decode()}")
                return rc
            checked.__name__ = name
            setattr(self, name, checked)
            return checked
        setattr(self, name, fn)
        return fn

    def symbols(self):
        self._load()
        return dict(self._protos)


# queries that return a value rather than an error code
_NO_CHECK = {"pfr_version", "pfr_conv1x1_dgrad2_bn_parts", "pfr_tuning_epoch", "pfr_gram_ws_floats", "pfr_conv1x1_tail_mtile", "pfr_bn_stats_rows_per_part", "pfr_bn_finalize_ws_floats", "pfr_topk_state_bytes", "pfr_layernorm_bwd_blocks", "pfr_window_bias_table_floats", "pfr_colsum_ws_floats", "pfr_conv2d_mtile", "pfr_gemm_act_mtile", "pfr_conv2d_dgrad_bn_parts", "pfr_plan_thunk_index", "pfr_plan_size", "pfr_plan_run", "pfr_conv2d_wgrad_splits", "pfr_colreduce_blocks", "pfr_match_ws_bytes", "pfr_pair_curve_ws_bytes", "pfr_augment_ws_bytes", "pfr_colsum_parts", "pfr_layernorm_bwd_dxsum_ok", "pfr_gemm_act_colsum_parts"}

lib = _Lib()

# optional per-launch tracer (bench.py uses it to bracket every kernel launch with HIP events)
_TRACER = [None]


def set_tracer(tracer):
    _TRACER[0] = tracer


class EventTracer:
    """Brackets every C-ABI launch with a pair of HIP events on the launching stream."""

    def __init__(self):
        import torch
        self._torch = torch
        self.records = []

    def before(self, name, args):
        e = self._torch.cuda.Event(enable_timing=True)
        e.record()
        self._cur = e

    def after(self, name, args):
        e = self._torch.cuda.Event(enable_timing=True)
        e.record()
        self.records.append((name, args, self._cur, e))

    def summary(self):
        """-> {name: [n_launches, total_ms]} (synchronises)"""
        self._torch.cuda.synchronize()
        out = {}
        for name, args, a, b in self.records:
            o = out.setdefault(name, [0, 0.0])
            o[0] += 1
            o[1] += a.elapsed_time(b)
        return out


def is_available() -> bool:
    return os.path.exists(LIB_PATH)


def dtype_id(torch_dtype_):
    import torch

    if torch_dtype_ == torch.float32:
        return PFR_F32
    if torch_dtype_ == torch.bfloat16:
        return PFR_BF16
    raise PfrError(f"unsupported dtype {torch_dtype_}")


def torch_dtype(dtype_id_):
    import torch

    return {PFR_F32: torch.float32, PFR_BF16: torch.bfloat16}[dtype_id_]
