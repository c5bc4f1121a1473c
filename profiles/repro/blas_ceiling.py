"""What a tuned library GEMM reaches on the conv-as-GEMM shapes of ResNet-50 bs256 (ceiling estimate only)."""
import torch, time
dev = "cuda:0"
shapes = [  # (M, N, K) name
    (802816, 64, 256, "l1 conv1 1x1 256->64"), (802816, 256, 64, "l1 conv3 1x1 64->256"), (802816, 64, 576, "l1 conv2 3x3 64 (K=576)"),
    (200704, 128, 1152, "l2 conv2 3x3 128"), (200704, 512, 128, "l2 conv3"), (200704, 128, 512, "l2 conv1"),
    (50176, 256, 2304, "l3 conv2 3x3 256"), (50176, 1024, 256, "l3 conv3"), (50176, 256, 1024, "l3 conv1"),
    (12544, 512, 4608, "l4 conv2 3x3 512"), (12544, 2048, 512, "l4 conv3"), (12544, 512, 2048, "l4 conv1"),
    (8192, 8192, 8192, "square 8k"), (10000, 65536, 512, "match chunk"),
]
for M, N, K, name in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        c = a @ b.t()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tf = 2.0 * M * N * K / ms / 1e9
    gb = (M * K + N * K + M * N) * 2 / ms / 1e6
    print(f"{name:28s} M={M:7d} N={N:5d} K={K:5d}  {ms*1e3:8.1f} us  {tf:7.1f} TF/s  {gb:7.1f} GB/s")
