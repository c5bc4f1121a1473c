"""Phase timing of the match's persistent filter GEMM (trace library: tools/build_trace.sh; PFR_LIB_PATH=.../libpfr_hip_trace.so): the LAST tile of every
workgroup — tile set-up, wait for the first k-step, k-loop, filter epilogue (stamps 5, 1, 2, 3, 6 of igemm_kernel)."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pets_face_recognition_amd._hip import lib, dtype_id
dll = ctypes.CDLL(os.environ['PFR_LIB_PATH'])
Q, n, D, K = 10000, 131072, 512, 128
q = torch.nn.functional.normalize(torch.randn(Q, D, device='cuda'), dim=1).bfloat16()
g = torch.nn.functional.normalize(torch.randn(n, D, device='cuda'), dim=1).bfloat16()
state = torch.zeros(lib.pfr_topk_state_bytes(Q, K), dtype=torch.uint8, device='cuda')
cand = torch.empty((Q, 1536), dtype=torch.int64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
# thresholds as in a late chunk: the key of a score few columns beat (0.17 ~ 3.8 sigma of 1/sqrt(512))
thr_off = Q * K * 8 + Q * 4 + 64
import struct
key = struct.unpack('<I', struct.pack('<f', float(os.environ.get('THR', '0.17'))))[0] | 0x80000000
state[thr_off:thr_off + Q * 4].view(torch.int32).fill_(key - (1 << 32))
def run():
    lib.pfr_match_scores_filter(q.data_ptr(), g.data_ptr(), dtype_id(torch.bfloat16), Q, n, D, 0, K, state.data_ptr(), cand.data_ptr(), 1536, 0, st)
for _ in range(3):
    run()
torch.cuda.synchronize()
tr = torch.zeros(4096, 8, dtype=torch.int64, device='cuda')
dll.pfr_debug_igemm_trace(ctypes.c_void_p(tr.data_ptr()))
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record(); run(); b.record(); torch.cuda.synchronize()
dll.pfr_debug_igemm_trace(ctypes.c_void_p(0))
t = tr.cpu().numpy().astype('float64')[:256] * 0.01
ntile = ((Q + 255) // 256) * ((n + 255) // 256)
print(f"kernel {a.elapsed_time(b)*1e3:.1f} us, {ntile} tiles, {ntile/256:.1f} per workgroup -> {a.elapsed_time(b)*1e3/(ntile/256):.2f} us per tile; MFMA floor {2*256*256*D/ (2.5e15/256) * 1e6:.2f} us")
print("last tile of each workgroup, mean us: set-up %.2f | wait first k-step %.2f | k-loop %.2f | epilogue + exit %.2f" % (
    (t[:, 1] - t[:, 5]).mean(), (t[:, 2] - t[:, 1]).mean(), (t[:, 3] - t[:, 2]).mean(), (t[:, 6] - t[:, 3]).mean()))
print("candidates appended per query (this chunk): %.1f" % (state[thr_off + Q * 4: thr_off + Q * 8].view(torch.int32).float().mean().item()))
