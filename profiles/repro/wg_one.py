import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pets_face_recognition_amd._hip import ops
N, H, W, C, Co, R, s, p = (256, 14, 14, 256, 256, 3, 1, 1) if len(sys.argv) < 2 or sys.argv[1] == "a" else (256, 14, 14, 256, 1024, 1, 1, 0)
x = torch.randn(N, H, W, C, device='cuda').bfloat16()
OH = (H + 2 * p - R) // s + 1
dy = torch.randn(N, OH, OH, Co, device='cuda').bfloat16()
ws = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device='cuda')
out = ops.conv2d_wgrad(x, dy, R, R, s, p, workspace=ws)
for _ in range(5):
    ops.conv2d_wgrad(x, dy, R, R, s, p, out=out, workspace=ws)
torch.cuda.synchronize()
