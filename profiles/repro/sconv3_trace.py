import sys, os, ctypes
sys.path.insert(0, "/root/repo")
import torch
from pets_face_recognition_amd._hip import lib, ops as o
from pets_face_recognition_amd._hip import lib as L
raw = ctypes.CDLL(os.environ["PFR_LIB_PATH"])
dev = "cuda:0"
lib.pfr_set_tuning(b"sconv", 2)
N = 256
x = torch.randn(N, 56, 56, 64, device=dev).bfloat16()
w = (torch.randn(64, 3, 3, 64, device=dev) / 24).bfloat16()
y = torch.empty_like(x)
tr = torch.zeros(1024, 4, dtype=torch.int64, device=dev)
for _ in range(3):
    o.conv2d_fwd(x, w, stride=1, pad=1, out=y)
torch.cuda.synchronize()
raw.pfr_debug_sconv3_trace(ctypes.c_void_p(tr.data_ptr()))
o.conv2d_fwd(x, w, stride=1, pad=1, out=y)
torch.cuda.synchronize()
t = tr.double().cpu()
nb = t[:, 3]
print("per patch (memtime ticks @100MHz?): wait %.1f loop %.1f total %.1f  patches/wave %.1f" % ((t[:, 0] / nb).mean(), (t[:, 1] / nb).mean(), (t[:, 2] / nb).mean(), nb.mean()))
print("total ticks per wave", t[:, 2].mean())
