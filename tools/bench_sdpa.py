"""What torch's scaled_dot_product_attention (the vendor flash kernels on ROCm) reaches on the attention shape of one
8-tile batch (32 sub-tiles x 16 heads, 1024 tokens, head dim 64, bf16), without a bias and with an additive float bias
(the decomposed rel-pos bias materialised as attn_mask): a reference point for k_attention4p, which fuses the bias."""
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
B, H, T, D = 32, 16, 1024, 64
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(B, H, T, D, generator=g).to(torch.bfloat16).to(dev) for _ in range(3))
bias = (torch.randn(B, H, T, T, generator=g) * 0.1).to(torch.bfloat16).to(dev)
flops = 4.0 * B * H * T * T * D
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    return sorted(ts)[2]
for name, fn in (("no bias", lambda: F.scaled_dot_product_attention(q, k, v)),
                 ("additive bias (attn_mask)", lambda: F.scaled_dot_product_attention(q, k, v, attn_mask=bias))):
    try:
        us = timeit(fn)
        print(f"SDPA {name:26s}: median {us:7.1f} us  {flops / us / 1e6:7.1f} TFLOP/s = {flops / us / 1e6 / 2500:.3f} of peak", flush=True)
    except Exception as e:
        print(f"SDPA {name}: failed: {type(e).__name__}: {str(e)[:200]}", flush=True)
for be in ("flash", "efficient", "math"):
    try:
        from torch.nn.attention import sdpa_kernel, SDPBackend
        b = {"flash": SDPBackend.FLASH_ATTENTION, "efficient": SDPBackend.EFFICIENT_ATTENTION, "math": SDPBackend.MATH}[be]
        with sdpa_kernel(b):
            us = timeit(lambda: F.scaled_dot_product_attention(q, k, v))
        print(f"SDPA backend {be:10s} no bias: median {us:7.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
    except Exception as e:
        print(f"SDPA backend {be}: {type(e).__name__}: {str(e)[:160]}", flush=True)
