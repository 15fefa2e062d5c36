"""What the vendor library (hipBLASLt behind torch.mm / F.linear, bf16) reaches on the four per-layer GEMM shapes:
a reference point for the hand-written 256^2 kernels' main loop (no GELU / LayerNorm / residual epilogue here)."""
import torch
dev = torch.device("cuda:0")
M = 32768
g = torch.Generator().manual_seed(0)
for name, N, K in (("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    W = torch.randn(N, K, generator=g).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
    for label, fn in (("A @ W^T", lambda: torch.mm(A, W.t())), ("linear + bias", lambda: torch.nn.functional.linear(A, W, b))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        ts = []
        for rnd in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        us = sorted(ts)[2]
        print(f"{name:5s} N={N:5d} K={K:5d} {label:14s}: median {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
