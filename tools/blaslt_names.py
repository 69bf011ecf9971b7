"""development aid: which hipBLASLt kernels torch.matmul picks for the train step's NT shapes (run under rocprofv3 --kernel-trace --stats)"""
import torch
dt, dev = torch.bfloat16, "cuda"
T = 32768
for (N, K) in [(1536, 512), (512, 512), (2048, 512), (512, 2048), (1024, 512)]:
    A = torch.randn(T, K, device=dev).to(dt); B = torch.randn(N, K, device=dev).to(dt)
    for _ in range(6):
        C = A @ B.t()
    torch.cuda.synchronize()
