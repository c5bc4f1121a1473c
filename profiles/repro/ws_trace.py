import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pets_face_recognition_amd._hip import ops, lib
from pets_face_recognition_amd._hip.lib import LIB_PATH
dll = ctypes.CDLL(LIB_PATH)
CASES = {'c3x3_256_h14': (256, 14, 256, 256, 3), 'c1x1_1024_256_h14': (256, 14, 1024, 256, 1), 'c1x1_256_1024_h14': (256, 14, 256, 1024, 1),
         'c1x1_64_256_h56': (256, 56, 64, 256, 1)}
lib.pfr_set_tuning(b"igemm_ws", 2)
for dbg in (0, 50):
    dll.pfr_debug_igemm_flags(dbg)
    for name, (N, H, C, Co, R) in CASES.items():
        x = torch.randn(N, H, H, C, device='cuda').bfloat16()
        w = (torch.randn(Co, R, R, C, device='cuda') / (C * R * R) ** 0.5).bfloat16()
        y, part = ops.conv2d_fwd(x, w, stride=1, pad=(R - 1) // 2, stats=False)
        tr = torch.zeros(4096, 8, dtype=torch.int64, device='cuda')
        torch.cuda.synchronize()
        dll.pfr_debug_igemm_trace(ctypes.c_void_p(tr.data_ptr()))
        a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        a.record(); ops.conv2d_fwd(x, w, stride=1, pad=(R - 1) // 2, stats=False, out=y); e.record()
        torch.cuda.synchronize()
        dll.pfr_debug_igemm_trace(ctypes.c_void_p(0))
        t = tr.cpu().numpy().astype('float64')
        t = t[t[:, 6] != 0]
        ks = t[:, 6].sum()
        print(f"dbg={dbg} {name}: {a.elapsed_time(e)*1e3:.1f} us, {len(t)} WGs, k-steps/WG {t[:,6].mean():.1f}; per k-step cycles: "
              f"MFMA wave work {t[:,0].sum()/ks:.0f} barrier {t[:,1].sum()/ks:.0f} | memory wave gload {t[:,2].sum()/ks:.0f} stores {t[:,3].sum()/ks:.0f} "
              f"vmcnt {t[:,4].sum()/ks:.0f} barrier {t[:,5].sum()/ks:.0f}", flush=True)
