import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pets_face_recognition_amd._hip import lib, dtype_id
dev = "cuda:0"
B, H, W, heads, hd, w, shift = 128, 56, 56, 3, 32, 7, (3 if len(sys.argv) > 1 and sys.argv[1] == "s" else 0)
C = heads * hd
qkv = torch.randn(B, H, W, 3 * C, device=dev).bfloat16()
dout = torch.randn(B, H, W, C, device=dev).bfloat16()
dqkv = torch.empty_like(qkv); out = torch.empty_like(dout)
pos = (torch.randn(2 * w - 1, 2 * w - 1, device=dev) * 0.02).contiguous()
tab = torch.empty(lib.pfr_window_bias_table_floats(w), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
lib.pfr_window_bias_table(pos.data_ptr(), tab.data_ptr(), w, shift, st)
nblk = B * (H // w) * (W // w) * heads
dpart = torch.empty(nblk, 169, dtype=torch.float32, device=dev)
for _ in range(3):
    lib.pfr_window_attn_fwd(qkv.data_ptr(), tab.data_ptr(), out.data_ptr(), 1, B, H, W, heads, hd, w, shift, hd ** -0.5, st)
    lib.pfr_window_attn_bwd(qkv.data_ptr(), tab.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dpart.data_ptr(), 1, B, H, W, heads, hd, w, shift, hd ** -0.5, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    lib.pfr_window_attn_bwd(qkv.data_ptr(), tab.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dpart.data_ptr(), 1, B, H, W, heads, hd, w, shift, hd ** -0.5, st)
e1.record(); torch.cuda.synchronize()
print("attn", "bwd us", e0.elapsed_time(e1) * 100, "units", nblk)
e0.record()
for _ in range(10):
    lib.pfr_window_attn_fwd(qkv.data_ptr(), tab.data_ptr(), out.data_ptr(), 1, B, H, W, heads, hd, w, shift, hd ** -0.5, st)
e1.record(); torch.cuda.synchronize()
print("attn fwd us", e0.elapsed_time(e1) * 100)
