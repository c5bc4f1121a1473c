import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pets_face_recognition_amd._hip import ops, lib
case = sys.argv[1] if len(sys.argv) > 1 else "a"
N, H, C, Co, R, pad = {"a": (256, 14, 256, 256, 3, 1), "b": (256, 14, 256, 1024, 1, 0), "c": (256, 56, 64, 64, 3, 1), "d": (256, 56, 256, 64, 1, 0), "e": (256, 14, 1024, 256, 1, 0), "f": (256, 28, 128, 128, 3, 1), "g": (256, 7, 512, 512, 3, 1)}[case]
x = torch.randn(N, H, H, C, device='cuda').bfloat16()
w = (torch.randn(Co, R, R, C, device='cuda') / (C * R * R) ** 0.5).bfloat16()
y, part = ops.conv2d_fwd(x, w, stride=1, pad=pad, stats=True)
for _ in range(3):
    ops.conv2d_fwd(x, w, stride=1, pad=pad, stats=True, out=y)
torch.cuda.synchronize()
