"""TEST-ONLY stand-in for dclip_amd.ops on machines without a GPU.

Monkeypatched into `dclip_amd.functional` by tests/test_dist_cpu.py so that the HOST LOGIC of the data-parallel
loss (offsets, gathers, coefficient, LSE exchange — dclip_amd/functional.py, dclip_amd/dist.py) can run under
`gloo` with world_size 2.  Plain torch, same signatures as the ops it replaces.  Never imported by the product.
"""
import torch

NORM_EPS = 1e-12


def normalize_rows_fwd(x):
    n = torch.linalg.vector_norm(x, dim=1).clamp_min(NORM_EPS)
    return x / n[:, None], 1.0 / n


def normalize_rows_bwd(dxhat, xhat, inv, dx=None, accumulate=False):
    dot = (dxhat * xhat).sum(1, keepdim=True)
    clamped = (inv >= 1.0 / NORM_EPS)[:, None]
    return torch.where(clamped, dxhat * inv[:, None], (dxhat - xhat * dot) * inv[:, None])


def contrastive_lse(a_local, b_global, offset, inv_temp):
    z = (a_local @ b_global.t()) * inv_temp
    idx = torch.arange(a_local.shape[0])
    return torch.logsumexp(z, 1), z[idx, idx + offset]


def contrastive_grad(a_local, b_global, lse_row, lse_col, offset, inv_temp, coef):
    z = (a_local @ b_global.t()) * inv_temp
    w = torch.exp(z - lse_row[:, None]) + torch.exp(z - lse_col[None, :])
    idx = torch.arange(a_local.shape[0])
    w[idx, idx + offset] -= 2.0
    return coef * (w @ b_global)


def sub_reduce(a, b, scale, out=None, accumulate=False):
    v = scale * (a - (b if b is not None else 0.0)).sum()
    if out is not None and accumulate:
        out += v
        return out
    return v.clone()


def cosine_loss_fwd(s, t):
    cos = torch.nn.functional.cosine_similarity(s, t, dim=1, eps=1e-12)
    return (1.0 - cos).sum(), cos


def cosine_loss_bwd(s, t, cos, coef, ds=None, accumulate=False):
    ns = torch.linalg.vector_norm(s, dim=1, keepdim=True).clamp_min(NORM_EPS)
    nt = torch.linalg.vector_norm(t, dim=1, keepdim=True).clamp_min(NORM_EPS)
    return -coef * (t / nt - (s / ns) * cos[:, None]) / ns
