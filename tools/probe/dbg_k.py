import torch, sys
sys.path.insert(0, "/root/repo")
from dclip_amd import ops
dev = torch.device("cuda:0")
def rnd(shape, seed): return torch.randn(shape, generator=torch.Generator().manual_seed(seed))
for (M, N, K) in [(128, 256, 85), (128, 256, 84), (128, 256, 88), (128, 256, 21), (64, 64, 85), (128, 256, 149)]:
    dy, x = rnd((K, M), 1), rnd((K, N), 2)
    want = dy.to(torch.bfloat16).double().t() @ x.to(torch.bfloat16).double()
    dyT, xT = ops.transpose_bf16(dy.to(dev)), ops.transpose_bf16(x.to(dev))
    got = ops.gemm_bf16_wgrad(dyT, xT, K, None).double().cpu()
    ld = (K + 7) // 8 * 8
    a16 = torch.zeros(M, ld, dtype=torch.bfloat16); a16[:, :K] = dy.t().to(torch.bfloat16); a16 = a16.to(dev)
    w16 = torch.zeros(N, ld, dtype=torch.bfloat16); w16[:, :K] = x.t().to(torch.bfloat16); w16 = w16.to(dev)
    got2 = ops.gemm_bf16(a16, w16, k=K).double().cpu()
    e1 = float((got - want).abs().max() / want.abs().max()); e2 = float((got2 - want).abs().max() / want.abs().max())
    same = torch.equal(dyT.cpu(), a16.cpu())
    print(M, N, K, "transposed path err", e1, "cast path err", e2, "operands equal", same, tuple(dyT.shape), tuple(a16.shape))
