"""Experiment harness (library built with -DPFR_IGEMM_TRACE): time conv layers under debug flags (DBG env)."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import ops
dll = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pets-face-recognition_amd', 'csrc', 'libpfr_hip.so'))
CASES = {
    'c1x1_256_1024_h14': (256, 14, 14, 256, 1024, 1, 1, 0),
    'c1x1_128_512_h28': (256, 28, 28, 128, 512, 1, 1, 0),
    'c1x1_1024_256_h14': (256, 14, 14, 1024, 256, 1, 1, 0),
    'c1x1_512_128_h28': (256, 28, 28, 512, 128, 1, 1, 0),
    'c1x1_2048_512_h7': (256, 7, 7, 2048, 512, 1, 1, 0),
    'c1x1_512_2048_h7': (256, 7, 7, 512, 2048, 1, 1, 0),
    'c3x3_64_h56': (256, 56, 56, 64, 64, 3, 1, 1),
    'c3x3_128_h28': (256, 28, 28, 128, 128, 3, 1, 1),
    'c3x3_256_h14': (256, 14, 14, 256, 256, 3, 1, 1),
    'c3x3_512_h7': (256, 7, 7, 512, 512, 3, 1, 1),
}
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device='cuda')
for flags in [int(a) for a in os.environ.get('DBGS', '0,4').split(',')]:
    dll.pfr_debug_igemm_flags(flags)
    for name, (N, H, W, C, Co, R, s, p) in CASES.items():
        x = torch.randn(N, H, W, C, device='cuda').bfloat16()
        w = (torch.randn(Co, R, R, C, device='cuda') / (C * R * R) ** 0.5).bfloat16()
        y, part = ops.conv2d_fwd(x, w, stride=s, pad=p, stats=True)
        cold = 0.0
        for _ in range(5):
            big.add_(1.0)
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); ops.conv2d_fwd(x, w, stride=s, pad=p, stats=True, out=y, stats_buf=part); b.record()
            torch.cuda.synchronize(); cold += a.elapsed_time(b) / 5
        fl = 2.0 * N * y.shape[1] * y.shape[2] * Co * R * R * C
        print(f'dbg={flags} {name:16s} cold {cold*1e3:8.1f} us {fl/cold/1e9:7.1f} TF/s')
