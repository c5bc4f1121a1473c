import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pets_face_recognition_amd._hip import lib
DEV = "cuda"
N, H, C, Co, join = 2, 56, 256, 64, False
if len(sys.argv) > 1: join = True; C, Co = 64, 256
g = torch.Generator().manual_seed(1)
M = N * H * H
dy = torch.randn(N, H, H, C, generator=g).to(DEV).bfloat16()
wt = (torch.randn(Co, 1, 1, C, generator=g) / C ** 0.5).to(DEV).bfloat16()
bnx = torch.randn(N, H, H, Co, generator=g).to(DEV).bfloat16()
coef = torch.stack([torch.randn(Co, generator=g) * 0.1, torch.rand(Co, generator=g) + 0.5, torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.3]).to(DEV).contiguous()
res = torch.randn(N, H, H, Co, generator=g).to(DEV).bfloat16() if join else None
rmask = torch.randint(0, 256, (M, Co // 8), generator=g, dtype=torch.uint8).to(DEV) if join else None
bmask = torch.randint(0, 256, (M, Co // 8), generator=g, dtype=torch.uint8).to(DEV) if join else None
P = lambda t: 0 if t is None else t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
lib.pfr_set_tuning(b"sconv", 2); lib.pfr_set_tuning(b"bnb", 2)
np_ = lib.pfr_conv2d_dgrad_bn_parts(1, N, H, H, C, Co, 1, 1, 0, H, H)
print("parts", np_, flush=True)
part1 = torch.full((np_, 2, Co), float("nan"), device=DEV)
dx1 = torch.full((N, H, H, Co), float("nan"), device=DEV, dtype=torch.bfloat16)
torch.cuda.synchronize(); print("launch", flush=True)
lib.pfr_conv2d_dgrad_bn(dy.data_ptr(), wt.data_ptr(), dx1.data_ptr(), 1, N, H, H, C, Co, 1, 1, 0, 0, H, H, P(res), P(rmask), 0, bnx.data_ptr(), coef.data_ptr(), P(bmask), part1.data_ptr(), 0, 0, 0, st)
torch.cuda.synchronize(); print("done", float(dx1.float().abs().mean()), float(part1.abs().sum()), flush=True)
