"""RCCL all-reduce through the C-ABI (csrc/pfr_comm.hip) — for hosts without torch.distributed.

`engine/ddp.py` (the Python host's data-parallel wrapper) uses torch.distributed with backend "nccl" (= RCCL); this
class exposes the same collective as a non-Python host of `libpfr_hip.so` would drive it, replacing
DistributedDataParallel's bucket all-reduce (/root/reference/utils/__init__.py:114-119).
"""
import ctypes

import torch

from .lib import PFR_BF16, PFR_F32, PfrError, lib

UNIQUE_ID_BYTES = 128


def unique_id() -> bytes:
    """rank 0: the 128-byte rendezvous id the host distributes to every rank"""
    buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
    lib.pfr_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p))
    return buf.raw


class Communicator:
    """one per process / GPU; `allreduce_(t, average=True)` reduces a contiguous f32 / bf16 CUDA tensor in place"""

    def __init__(self, rank: int, world: int, uid: bytes, device=None):
        if len(uid) != UNIQUE_ID_BYTES:
            raise ValueError(f"unique id must be {UNIQUE_ID_BYTES} bytes")
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        with torch.cuda.device(self.device):
            self._h = lib.pfr_comm_init(rank, world, uid)
        if not self._h:
            raise PfrError(f"pfr_comm_init: {lib.pfr_last_error().decode()}")

    def allreduce_(self, t: torch.Tensor, average: bool = True, stream=None):
        if not t.is_cuda or not t.is_contiguous():
            raise ValueError("allreduce_ needs a contiguous CUDA tensor")
        dt = {torch.float32: PFR_F32, torch.bfloat16: PFR_BF16}.get(t.dtype)
        if dt is None:
            raise TypeError(f"unsupported dtype {t.dtype}")
        s = torch.cuda.current_stream(t.device) if stream is None else stream
        lib.pfr_comm_allreduce(self._h, t.data_ptr(), t.numel(), dt, 1 if average else 0, s.cuda_stream)
        return t

    def close(self):
        if self._h:
            lib.pfr_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
