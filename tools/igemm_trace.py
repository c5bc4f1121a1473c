"""Workgroup phase timing of the igemm kernel (needs a library built with -DPFR_IGEMM_TRACE):
   hipcc ... -DPFR_IGEMM_TRACE; python tools/igemm_trace.py"""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import ops
dll = ctypes.CDLL(os.environ.get('PFR_LIB_PATH') or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pets-face-recognition_amd', 'csrc', 'libpfr_hip.so'))
CASES = {
    'c1x1_256_1024_h14': (256, 14, 14, 256, 1024, 1, 1, 0),
    'c1x1_64_256_h56': (256, 56, 56, 64, 256, 1, 1, 0),
    'c1x1_128_512_h28': (256, 28, 28, 128, 512, 1, 1, 0),
    'c1x1_1024_256_h14': (256, 14, 14, 1024, 256, 1, 1, 0),
    'c3x3_256_h14': (256, 14, 14, 256, 256, 3, 1, 1),
    'c3x3_64_h56': (256, 56, 56, 64, 64, 3, 1, 1),
    'c3x3_128_h28': (256, 28, 28, 128, 128, 3, 1, 1),
    'c1x1_512_128_h28': (256, 28, 28, 512, 128, 1, 1, 0),
}
FLAGS = int(os.environ.get('DBG', '0'))
dll.pfr_debug_igemm_flags(FLAGS)
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device='cuda')
for name, (N, H, W, C, Co, R, s, p) in CASES.items():
    x = torch.randn(N, H, W, C, device='cuda').bfloat16()
    w = (torch.randn(Co, R, R, C, device='cuda') / (C * R * R) ** 0.5).bfloat16()
    y, part = ops.conv2d_fwd(x, w, stride=s, pad=p, stats=True)
    tr = torch.zeros(1 << 20, 8, dtype=torch.int64, device='cuda')
    big.add_(1.0)
    torch.cuda.synchronize()
    dll.pfr_debug_igemm_trace(ctypes.c_void_p(tr.data_ptr()))
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); ops.conv2d_fwd(x, w, stride=s, pad=p, stats=True, out=y, stats_buf=part); b.record()
    torch.cuda.synchronize()
    dll.pfr_debug_igemm_trace(ctypes.c_void_p(0))
    t = tr.cpu().numpy()
    n = int((t[:, 0] != 0).sum())
    if n == 0:
        print(f"{name}: not a tile-kernel launch (streaming kernel)")
        continue
    t = t[:n].astype('float64') * 0.01  # 100 MHz -> us
    t0 = t[:, 0].min()
    d = t[:, 1:7] - t[:, 0:6]
    life = t[:, 6] - t[:, 0]
    print(f"{name}: {n} WGs, kernel {a.elapsed_time(b)*1e3:.1f} us, span {t[:,6].max()-t0:.1f} us; WG life mean {life.mean():.2f} us")
    print("   phases mean us: setup %.2f | first-load %.2f | mainloop %.2f | epi1 %.2f | epi2(store) %.2f | stats %.2f" % tuple(d.mean(0)))
    # concurrency: average WGs alive
    print("   avg WGs alive: %.1f  (per CU %.2f)" % (life.sum() / (t[:, 6].max() - t0), life.sum() / (t[:, 6].max() - t0) / 256))
