"""Where a persistent GEMM workgroup spends its time (needs tools/build_trace.sh; run with PFR_LIB_PATH=.../libpfr_hip_trace.so)"""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import ops, lib
from pets_face_recognition_amd._hip.lib import LIB_PATH
dll = ctypes.CDLL(LIB_PATH)
CASES = {
    'c1x1_64_256_h56': (256, 56, 56, 64, 256, 1, 1, 0),
    'c1x1_256_64_h56': (256, 56, 56, 256, 64, 1, 1, 0),
    'c3x3_64_h56': (256, 56, 56, 64, 64, 3, 1, 1),
    'c1x1_128_512_h28': (256, 28, 28, 128, 512, 1, 1, 0),
    'c1x1_256_1024_h14': (256, 14, 14, 256, 1024, 1, 1, 0),
    'c1x1_1024_256_h14': (256, 14, 14, 1024, 256, 1, 1, 0),
    'c3x3_256_h14': (256, 14, 14, 256, 256, 3, 1, 1),
    'c1x1_512_2048_h7': (256, 7, 7, 512, 2048, 1, 1, 0),
}
dll.pfr_debug_igemm_flags(int(os.environ.get("DBG", "0")))
lib.pfr_set_tuning(b"igemm_p", 2)
lib.pfr_set_tuning(b"igemm_ptile", int(os.environ.get("PTILE", "0")))
lib.pfr_set_tuning(b"igemm_ppf", int(os.environ.get("PPF", "0")))
lib.pfr_set_tuning(b"igemm_pkch", int(os.environ.get("PKCH", "4")))
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device='cuda')
for stats in (True, False):
    for name, (N, H, W, C, Co, R, s, p) in CASES.items():
        x = torch.randn(N, H, W, C, device='cuda').bfloat16()
        w = (torch.randn(Co, R, R, C, device='cuda') / (C * R * R) ** 0.5).bfloat16()
        y, part = ops.conv2d_fwd(x, w, stride=s, pad=p, stats=stats)
        tr = torch.zeros(4096, 8, dtype=torch.int64, device='cuda')
        big.add_(1.0)
        torch.cuda.synchronize()
        dll.pfr_debug_igemm_trace(ctypes.c_void_p(tr.data_ptr()))
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); ops.conv2d_fwd(x, w, stride=s, pad=p, stats=stats, out=y, stats_buf=part); b.record()
        torch.cuda.synchronize()
        dll.pfr_debug_igemm_trace(ctypes.c_void_p(0))
        t = tr.cpu().numpy().astype('float64')
        t = t[t[:, 0] != 0]
        n = len(t)
        life = (t[:, 1] - t[:, 0]) * 0.01
        span = (t[:, 1].max() - t[:, 0].min()) * 0.01
        cyc = t[:, 2:5]
        tot = cyc.sum(1)
        clk = tot / life / 1e3   # GHz (cycles of the three phases / WG life)
        print(f"{name} stats={int(stats)}: {n} WGs, kernel {a.elapsed_time(b)*1e3:.1f} us, span {span:.1f}, WG life {life.mean():.1f} us "
              f"(min {life.min():.1f} max {life.max():.1f}); tiles/WG {t[:,5].mean():.1f}, k-steps/tile {t[:,6].sum()/t[:,5].sum():.1f}")
        per_tile = cyc.sum(0) / t[:, 5].sum()
        print("   cycles per tile: wait %.0f | mma %.0f | epilogue %.0f   (share %.0f%% / %.0f%% / %.0f%%)  ~clock %.2f GHz" %
              (per_tile[0], per_tile[1], per_tile[2], *(100 * cyc.sum(0) / cyc.sum()), clk.mean()))
