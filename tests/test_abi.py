"""The C-ABI library loads without a GPU and exports every symbol include/pfr_hip.h declares (no compute calls)."""
import ctypes
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    from pets_face_recognition_amd._hip.lib import LIB_PATH, parse_header, lib
    if not os.path.exists(LIB_PATH):
        import __graft_entry__ as g
        g.build()
    protos = parse_header()
    assert len(protos) >= 40
    dll = ctypes.CDLL(LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), f"{name} declared in include/pfr_hip.h but not exported"
    assert lib.pfr_version() >= 100
    # pure host-side queries work without a device
    assert lib.pfr_conv2d_mtile(256, 56, 56, 64, 64, 3, 3, 1, 1, 56, 56, 1, 1, 0) > 0
    assert lib.pfr_conv2d_wgrad_splits(802816, 64, 576) >= 1
    assert lib.pfr_colreduce_blocks(64, 1, 802816) >= 1
    assert lib.pfr_topk_state_bytes(10, 100) >= 10 * 100 * 8


def test_argument_errors_are_reported_not_crashed():
    from pets_face_recognition_amd._hip import lib, PfrError
    with pytest.raises(PfrError, match="null pointer"):
        lib.pfr_conv2d_fwd(0, 0, 0, 1, 1, 1, 8, 8, 8, 8, 1, 1, 1, 0, 0, 8, 8, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    with pytest.raises(PfrError, match="multiple of 8"):
        lib.pfr_conv2d_fwd(16, 16, 16, 1, 1, 1, 8, 8, 3, 8, 1, 1, 1, 0, 0, 8, 8, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0)


def test_hip_path_refuses_cpu_tensors():
    import torch
    from pets_face_recognition_amd._hip import ops, PfrError
    with pytest.raises(PfrError, match="no CPU fallback"):
        ops.conv2d_fwd(torch.zeros(1, 4, 4, 8), torch.zeros(8, 1, 1, 8))


def test_wgrad_split_model_is_sane_on_cpu():
    """pfr_conv2d_wgrad_splits is host arithmetic (no device work) and sizes the caller's workspace for whichever kernel takes the launch:
    >= 1 split, >= 128 reduction rows per split, the fp32 partial slabs within the 48 MB budget; 3x3-shaped geometries get the halo-staged
    kernel's slab count (256 workgroups / block pairs)"""
    from pets_face_recognition_amd._hip import lib
    for (M, Co, KK) in [(802816, 64, 576), (200704, 128, 1152), (50176, 256, 2304), (12544, 512, 4608), (802816, 256, 64),
                        (50176, 1024, 256), (12544, 2048, 512), (256, 512, 2048), (256, 10000, 512), (401408, 288, 96), (64, 8, 8)]:
        s = lib.pfr_conv2d_wgrad_splits(M, Co, KK)
        assert s >= 1 and (s == 1 or (M + s - 1) // s >= 128)
        assert s == 1 or s * Co * KK * 4 <= 48 << 20
    assert lib.pfr_conv2d_wgrad_splits(200704, 128, 1152) == 64           # 4 block pairs x 64 splits = 256 workgroups (tile kernel alone: 56)
    assert lib.pfr_conv2d_wgrad_splits(50176, 1024, 256) * 16 <= 512      # (tile kernel: whole 512-workgroup rounds)
