import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pets_face_recognition_amd._hip import ops, lib
mode = sys.argv[1] if len(sys.argv) > 1 else "ws"
lib.pfr_set_tuning(b"igemm_ws", 2 if mode == "ws" else 0)
N, H, C, Co, R = 256, 14, 256, 256, 3
x = torch.randn(N, H, H, C, device='cuda').bfloat16()
w = (torch.randn(Co, R, R, C, device='cuda') / (C * R * R) ** 0.5).bfloat16()
y, part = ops.conv2d_fwd(x, w, stride=1, pad=1, stats=False)
for _ in range(3):
    ops.conv2d_fwd(x, w, stride=1, pad=1, stats=False, out=y)
torch.cuda.synchronize()
