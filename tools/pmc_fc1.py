"""The dominant kernel alone -- an mlp.lin1 launch of one 8-tile batch: (32768 / cpx_net_mlp_parts) x 4096 x 1024 (the engine runs the MLP of 32
sub-tiles in two row parts of 16 384 tokens), folded LayerNorm + erf-GELU epilogue, random bf16 operands -- for rocprofv3 --pmc passes
(bench.py starts it twice as a child: FETCH_SIZE, WRITE_SIZE)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpose_amd import _lib, ops
dev = torch.device("cuda:0"); L = _lib.lib()
M, N, K = 32768 // int(L.cpx_net_mlp_parts(32, 0)), 4096, 1024
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
b = torch.randn(N, generator=g).to(dev)
stats = ops.row_stats(A); cs = W.float().sum(1).contiguous()
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(12):
    _lib.check(L.cpx_gemm_ln(A.data_ptr(), W.data_ptr(), M, N, K, ops.EPI["gelu"], b.data_ptr(), None, out.data_ptr(), N,
                             stats.data_ptr(), cs.data_ptr(), None, st))
torch.cuda.synchronize()
