from .lib import lib, PFR_F32, PFR_BF16, dtype_id, torch_dtype, is_available, PfrError, set_tracer, EventTracer  # noqa: F401
