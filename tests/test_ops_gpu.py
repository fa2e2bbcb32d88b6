"""GPU parity of each C-ABI op against the CPU oracle (float64 where it matters)."""
import numpy as np
import pytest
import torch

from dclip_amd import config as dcfg, synth
from oracle import dclip_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def relerr(got, want):
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))


# ---------------------------------------------------------------------------------- GEMM

GEMM_SHAPES = [(128, 128, 32), (400, 768, 768), (13, 64, 36), (616, 1536, 512), (257, 132, 100), (64, 2304, 768),
               (1000, 64, 3072)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("layout", [3, 1, 0, 2])
def test_gemm_layouts(dev, M, N, K, layout):
    from dclip_amd import ops
    if not (layout & 1) and M % 4:
        pytest.skip("A [K][M] needs M % 4 == 0")
    a = rnd((M, K), 1)
    b = rnd((N, K), 2)
    want = a.double() @ b.double().t()
    a_in = a if layout & 1 else a.t().contiguous()
    b_in = b if layout & 2 else b.t().contiguous()
    if (layout & 1) and K % 4:          # K-major rows must stay 16-byte aligned: pad the leading dim
        pytest.skip("ragged K with lda == K is not 16-byte aligned")
    got = ops.gemm(a_in.to(dev), b_in.to(dev), layout)
    assert relerr(got, want) < 2e-6 * max(1, K ** 0.5)


def test_gemm_a_is_identity_asymmetric_b(dev):
    """A = I with an asymmetric B catches a transposed C write or a permuted k map."""
    from dclip_amd import ops
    n = 96
    b = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 7.0
    eye = torch.eye(n)
    got = ops.gemm(eye.to(dev), b.to(dev), 3).cpu()          # I @ b^T
    assert torch.equal(got, b.t())
    got = ops.gemm(eye.to(dev), b.to(dev), 1).cpu()          # I @ b
    assert torch.equal(got, b)
    got = ops.gemm(b.to(dev), eye.to(dev), 0).cpu()          # b^T @ I
    assert torch.equal(got, b.t())


def test_gemm_epilogues(dev):
    from dclip_amd import ops
    M, N, K = 300, 256, 128
    a, w, bias, res = rnd((M, K), 1), rnd((N, K), 2, 0.2), rnd((N,), 3), rnd((M, N), 4)
    z = a.double() @ w.double().t() + bias.double()
    aux = torch.empty((M, N), device=dev)
    got = ops.gemm(a.to(dev), w.to(dev), 3, bias=bias.to(dev), aux=aux, epilogue=ops.EPI_GELU)
    assert relerr(aux, z) < 1e-5
    assert relerr(got, O.quick_gelu(z)) < 1e-5
    got = ops.gemm(a.to(dev), w.to(dev), 3, bias=bias.to(dev), residual=res.to(dev))
    assert relerr(got, z + res.double()) < 1e-5
    # DGELU: C = (a w^T) * gelu'(aux)
    h = rnd((M, N), 5, 2.0)
    hd = h.double().requires_grad_(True)
    O.quick_gelu(hd).sum().backward()
    got = ops.gemm(a.to(dev), w.to(dev), 3, aux=h.to(dev), epilogue=ops.EPI_DGELU)
    assert relerr(got, (a.double() @ w.double().t()) * hd.grad) < 1e-5
    # ACCUM + alpha
    c0 = rnd((M, N), 6)
    out = c0.to(dev).clone()
    ops.gemm(a.to(dev), w.to(dev), 3, out=out, epilogue=ops.EPI_ACCUM, alpha=0.5)
    assert relerr(out, c0.double() + 0.5 * (a.double() @ w.double().t())) < 1e-5


@pytest.mark.parametrize("split", [2, 4, 8])
def test_gemm_split_k_wgrad(dev, split):
    from dclip_amd import ops
    rows, n_out, n_in = 1000, 192, 320
    dy, x = rnd((rows, n_out), 1), rnd((rows, n_in), 2)
    want = dy.double().t() @ x.double()
    g0 = rnd((n_out, n_in), 3)
    out = g0.to(dev).clone()
    ops.gemm(dy.to(dev), x.to(dev), 0, out=out, epilogue=ops.EPI_ACCUM, split_k=split)
    assert relerr(out, g0.double() + want) < 1e-5
    out2 = g0.to(dev).clone()
    ops.gemm(dy.to(dev), x.to(dev), 0, out=out2, epilogue=ops.EPI_ACCUM, split_k=split)
    assert torch.equal(out, out2), "split-K must be run-to-run deterministic"


def test_gemm_auto_plan_big_wgrad(dev):
    from dclip_amd import ops
    rows, n_out, n_in = 6400, 768, 768
    dy, x = rnd((rows, n_out), 1, 0.1), rnd((rows, n_in), 2)
    got = ops.gemm(dy.to(dev), x.to(dev), 0)
    assert relerr(got, dy.double().t() @ x.double()) < 2e-5


def test_colsum(dev):
    from dclip_amd import ops
    x = rnd((1234, 768), 1)
    got = ops.colsum(x.to(dev))
    assert relerr(got, x.double().sum(0)) < 1e-5
    acc = torch.ones(768, device=dev)
    ops.colsum(x.to(dev), out=acc, accumulate=True)
    assert relerr(acc, 1 + x.double().sum(0)) < 1e-5


# ---------------------------------------------------------------------------------- LayerNorm

@pytest.mark.parametrize("rows,D", [(7, 64), (400, 768), (33, 512), (10, 1024), (5, 128), (6, 2048)])
def test_layernorm_fwd_bwd(dev, rows, D):
    from dclip_amd import ops
    x, g, b = rnd((rows, D), 1, 2.0) + 0.5, 1 + rnd((D,), 2, 0.1), rnd((D,), 3, 0.1)
    dy, dres = rnd((rows, D), 4), rnd((rows, D), 5)
    xd, gd, bd = (t.double().requires_grad_(True) for t in (x, g, b))
    y = O.layer_norm(xd, gd, bd, 1e-5)
    (y * dy.double()).sum().backward()
    got, mean, rstd = ops.layernorm_fwd(x.to(dev), g.to(dev), b.to(dev), 1e-5)
    assert relerr(got, y) < 1e-5
    dx, dg, db = ops.layernorm_bwd(dy.to(dev), x.to(dev), g.to(dev), mean, rstd, dresidual=dres.to(dev))
    assert relerr(dx, xd.grad + dres.double()) < 2e-5
    assert relerr(dg, gd.grad) < 2e-5
    assert relerr(db, bd.grad) < 2e-5
    dg2 = torch.ones(D, device=dev)
    db2 = torch.ones(D, device=dev)
    ops.layernorm_bwd(dy.to(dev), x.to(dev), g.to(dev), mean, rstd, dgamma=dg2, dbeta=db2, accumulate=True)
    assert relerr(dg2, 1 + gd.grad) < 2e-5


# ---------------------------------------------------------------------------------- attention

def ref_attention(qkv, B, S, H, causal):
    D = H * 64
    q, k, v = (qkv.view(B, S, 3, H, 64)[:, :, i].transpose(1, 2) for i in range(3))
    s = (q @ k.transpose(-1, -2)) * 0.125
    if causal:
        s = s + torch.full((S, S), float("-inf"), dtype=qkv.dtype).triu(1)
    p = O.softmax_lastdim(s)
    return (p @ v).transpose(1, 2).reshape(B * S, D)


@pytest.mark.parametrize("B,S,H,causal", [(2, 50, 2, False), (3, 77, 2, True), (1, 197, 3, False), (2, 10, 1, True),
                                          (1, 257, 2, False), (2, 16, 2, True), (1, 64, 1, False), (1, 130, 1, True),
                                          (2, 65, 2, True), (2, 80, 1, False), (2, 77, 2, False), (1, 81, 1, True),
                                          (2, 1, 1, True), (2, 33, 2, False), (1, 48, 1, True)])
def test_attention_fwd_bwd(dev, B, S, H, causal):
    from dclip_amd import ops
    qkv = rnd((B * S, 3 * H * 64), 1, 1.5)
    dout = rnd((B * S, H * 64), 2)
    qd = qkv.double().requires_grad_(True)
    want = ref_attention(qd, B, S, H, causal)
    (want * dout.double()).sum().backward()
    out, lse = ops.attention_fwd(qkv.to(dev), B, S, H, causal)
    assert relerr(out, want) < 2e-5
    dqkv = ops.attention_bwd(qkv.to(dev), out, dout.to(dev), lse, B, S, H, causal)
    assert relerr(dqkv, qd.grad) < 5e-5


@pytest.mark.parametrize("B,S,H", [(2, 197, 3), (1, 257, 2), (3, 81, 1), (2, 128, 2), (1, 224, 1), (2, 225, 1), (1, 300, 2), (1, 512, 1), (1, 520, 1)])
def test_attention_bwd_long_ds_passing_vs_recompute(dev, monkeypatch, B, S, H):
    """Long non-causal sequences: the default backward forms dS once in the dK/dV kernel and hands it to the dQ kernel
    through the workspace; DCLIP_ATTN_NO_DS=1 keeps the two-kernel split that forms S and dP twice.  Both against fp64,
    dK / dV bit-identical between them (the same kernel body), and the C ABI refuses a short workspace."""
    from dclip_amd import ops, _lib
    qkv = rnd((B * S, 3 * H * 64), 11, 1.5)
    dout = rnd((B * S, H * 64), 12)
    qd = qkv.double().requires_grad_(True)
    (ref_attention(qd, B, S, H, False) * dout.double()).sum().backward()
    out, lse = ops.attention_fwd(qkv.to(dev), B, S, H, False)
    got = ops.attention_bwd(qkv.to(dev), out, dout.to(dev), lse, B, S, H, False)
    monkeypatch.setenv("DCLIP_ATTN_NO_DS", "1")
    old = ops.attention_bwd(qkv.to(dev), out, dout.to(dev), lse, B, S, H, False)
    monkeypatch.delenv("DCLIP_ATTN_NO_DS")
    assert relerr(got, qd.grad) < 5e-5 and relerr(old, qd.grad) < 5e-5
    D = H * 64
    assert relerr(got[:, :D], old[:, :D]) < 2e-5                     # dQ: another kernel, same arithmetic up to order
    assert relerr(got[:, D:], old[:, D:]) < 2e-6                     # dK, dV: only delta's summation order differs
    lib = _lib.load()
    need = int(lib.dclip_attention_bwd_workspace(B, S, H, 0))
    Sp = (S + 31) // 32 * 32
    assert need >= 4 * (B * H * S + (B * H * Sp * Sp if S <= 512 else 0))    # longer: the two-kernel split, delta only
    assert int(lib.dclip_attention_bwd_workspace(B, 50, H, 0)) == 4 * ((B * H * 50 + 63) // 64 * 64)      # short: delta only
    ws = torch.empty(need // 4, dtype=torch.float32, device=dev)
    dq = torch.empty_like(got)
    rc = lib.dclip_attention_bwd_ws(qkv.to(dev).data_ptr(), out.data_ptr(), dout.to(dev).data_ptr(), lse.data_ptr(), dq.data_ptr(),
                                    ws.data_ptr(), need - 4, B, S, H, 0, None)
    assert rc != 0 and "workspace too small" in lib.dclip_last_error().decode()


def test_attention_online_softmax_rescale(dev):
    """A spike in a late key tile forces the running-max rescale branch (S spans several tiles)."""
    from dclip_amd import ops
    B, S, H = 1, 200, 1
    qkv = rnd((B * S, 3 * 64), 3, 0.5)
    qkv[150, 64:128] = qkv[7, 0:64] * 40.0          # key 150 aligned with query 7: huge logit in tile 2
    want = ref_attention(qkv.double(), B, S, H, False)
    out, _ = ops.attention_fwd(qkv.to(dev), B, S, H, False)
    assert relerr(out, want) < 2e-5


# ---------------------------------------------------------------------------------- embeddings

@pytest.mark.parametrize("cfgname", ["tiny", "ViT-B/32", "ViT-L/14"])
def test_patch_embed_path(dev, cfgname):
    from dclip_amd import ops
    v = dcfg.NAMED[cfgname]().vision
    B = 2
    pix = synth.synth_pixel_values(B, v, seed=0)
    w = rnd((v.hidden_size, v.num_channels, v.patch_size, v.patch_size), 1, 0.02)
    cls, pos = rnd((v.hidden_size,), 2), rnd((v.seq_len, v.hidden_size), 3)
    conv = torch.nn.functional.conv2d(pix.double(), w.double(), stride=v.patch_size).flatten(2).transpose(1, 2)
    want = torch.cat([cls.double().expand(B, 1, -1), conv], 1) + pos.double()
    cols = ops.im2col(pix.to(dev), v.patch_size)
    pe = ops.gemm(cols, w.reshape(v.hidden_size, -1).contiguous().to(dev), 3)
    x = ops.vision_assemble_fwd(pe, cls.to(dev), pos.to(dev), B, v.seq_len, v.hidden_size)
    assert relerr(x, want.reshape(B * v.seq_len, -1)) < 1e-5
    dx = rnd((B * v.seq_len, v.hidden_size), 4)
    dp = ops.vision_assemble_bwd(dx.to(dev), B, v.seq_len, v.hidden_size)
    assert torch.equal(dp.cpu(), dx.view(B, v.seq_len, -1)[:, 1:].reshape(-1, v.hidden_size))


def test_text_embed_eos_gather_scatter(dev):
    from dclip_amd import ops
    t = dcfg.tiny().text
    B, T, D = 5, t.max_position_embeddings, t.hidden_size
    ids = synth.synth_input_ids(B, t, seed=3, ragged=True)
    ids[1, 1:] = t.eos_token_id
    tok, pos = rnd((t.vocab_size, D), 1), rnd((T, D), 2)
    x = ops.text_embed_fwd(ids.to(dev), tok.to(dev), pos.to(dev))
    assert torch.equal(x.cpu(), (tok[ids] + pos[:T]).reshape(B * T, D))
    eos = ops.first_eos(ids.to(dev), t.eos_token_id)
    assert torch.equal(eos.cpu().long(), O.first_eos_index(ids, t.eos_token_id).long())
    none = torch.zeros(2, T, dtype=torch.int64)
    assert ops.first_eos(none.to(dev), t.eos_token_id).cpu().tolist() == [0, 0]
    g = ops.gather_rows(x, eos, B, T, D)
    assert torch.equal(g.cpu(), x.cpu().view(B, T, D)[torch.arange(B), eos.cpu().long()])
    g0 = ops.gather_rows(x, None, B, T, D)
    assert torch.equal(g0.cpu(), x.cpu().view(B, T, D)[:, 0])
    dout = rnd((B, D), 5)
    dx = ops.scatter_rows(dout.to(dev), eos, B, T, D).cpu().view(B, T, D)
    want = torch.zeros(B, T, D)
    want[torch.arange(B), eos.cpu().long()] = dout
    assert torch.equal(dx, want)
    dtok = torch.zeros(t.vocab_size, D, device=dev)
    dxx = rnd((B * T, D), 6)
    ops.text_embed_bwd(ids.to(dev), dxx.to(dev), dtok)
    want = torch.zeros(t.vocab_size, D, dtype=torch.float64).index_add_(0, ids.reshape(-1), dxx.double())
    assert relerr(dtok, want) < 1e-5


# ---------------------------------------------------------------------------------- loss pieces

def test_normalize_rows(dev):
    from dclip_amd import ops
    x = rnd((9, 512), 1, 3.0)
    x[4] = 0
    xd = x.double().requires_grad_(True)
    y = O.l2_normalize(xd)
    dy = rnd((9, 512), 2)
    (y * dy.double()).sum().backward()
    xhat, inv = ops.normalize_rows_fwd(x.to(dev))
    assert relerr(xhat, y) < 1e-6
    dx = ops.normalize_rows_bwd(dy.to(dev), xhat, inv)
    assert relerr(dx[[0, 1, 2, 3, 5, 6, 7, 8]], xd.grad[[0, 1, 2, 3, 5, 6, 7, 8]]) < 1e-5
    assert relerr(dx[4], xd.grad[4]) < 1e-5          # clamp active: dx = dy / eps


@pytest.mark.parametrize("Bl,Bg,P,offset", [(8, 8, 512, 0), (37, 37, 768, 0), (256, 256, 512, 0), (64, 256, 512, 128),
                                            (3, 5, 64, 2), (200, 1000, 512, 400),
                                            # the shapes the 8-GPU configs produce per rank (BASELINE c5: 512 local x 4096
                                            # global negatives, rank 3's offset; c4: 128 x 1024): other tile counts, other
                                            # part_m / part_s slab sizes, the 8 MB W workspace
                                            (512, 4096, 512, 1536), (128, 1024, 512, 384)])
def test_contrastive_lse_and_grad(dev, Bl, Bg, P, offset):
    from dclip_amd import ops
    inv_t = 20.0
    a = O.l2_normalize(rnd((Bl, P), 1))
    b = O.l2_normalize(rnd((Bg, P), 2) + 0.3 * torch.cat([torch.zeros(offset, P), a, torch.zeros(Bg - offset - Bl, P)]))
    z = (a.double() @ b.double().t()) * inv_t
    lse, diag = ops.contrastive_lse(a.to(dev), b.to(dev), offset, inv_t)
    assert relerr(lse, torch.logsumexp(z, 1)) < 1e-5
    assert relerr(diag, z[torch.arange(Bl), torch.arange(Bl) + offset]) < 1e-5
    lse_row = torch.logsumexp(z, 1).float()
    lse_col = rnd((Bg,), 3).abs() + 4.0                      # any vector: the op is linear in exp(-lse_col)
    w = torch.exp(z - lse_row.double()[:, None]) + torch.exp(z - lse_col.double()[None, :])
    w[torch.arange(Bl), torch.arange(Bl) + offset] -= 2.0
    want = 0.37 * (w @ b.double())
    got = ops.contrastive_grad(a.to(dev), b.to(dev), lse_row.to(dev), lse_col.to(dev), offset, inv_t, 0.37)
    assert relerr(got, want) < 2e-5


def test_cosine_loss_against_golden(dev, golden):
    from dclip_amd import ops
    g = golden("losses.npz")
    for name in ("b1_p512", "b2_p512", "b8_p512", "b8_p64", "b37_p768"):
        s, t = torch.from_numpy(g[f"{name}.img"]), torch.from_numpy(g[f"{name}.tea"])
        B = s.shape[0]
        loss_sum, cos = ops.cosine_loss_fwd(s.to(dev), t.to(dev))
        assert abs(float(loss_sum) / B - float(g[f"{name}.cos"])) < 1e-5 * max(1.0, abs(float(g[f"{name}.cos"])))
        ds = ops.cosine_loss_bwd(s.to(dev), t.to(dev), cos, 1.0 / B)
        assert relerr(ds, torch.from_numpy(g[f"{name}.d_stu"])) < 1e-4
    z, zt = torch.from_numpy(g["zero_row.img"]), torch.from_numpy(g["zero_row.txt"])
    loss_sum, _ = ops.cosine_loss_fwd(z.to(dev), zt.to(dev))
    assert abs(float(loss_sum) / 4 - float(g["zero_row.cos"])) < 1e-5


# ---------------------------------------------------------------------------------- optimiser tail

def test_fused_adamw_and_clip_match_torch(dev):
    from dclip_amd.optim import FusedAdamW
    shapes = [(768, 768), (2304,), (7,), (512, 3, 4, 4), ()]
    ps = [torch.nn.Parameter(rnd(s, i) if s else torch.tensor(0.3)) for i, s in enumerate(shapes)]
    mine = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ps]
    ref_opt = torch.optim.AdamW(ps, lr=1e-2)
    my_opt = FusedAdamW(mine, lr=1e-2, max_grad_norm=0.5)
    for step in range(4):
        for j, (p, q) in enumerate(zip(ps, mine)):
            g = rnd(p.shape, 100 * step + j, 0.05) if p.dim() else torch.tensor(0.01 * (step + 1))
            p.grad, q.grad = g.clone(), g.clone().to(dev)
        norm = torch.nn.utils.clip_grad_norm_(ps, 0.5)
        ref_opt.step()
        my_opt.step()
        assert abs(float(my_opt.last_grad_norm) - float(norm)) < 1e-5 * float(norm)
    for p, q in zip(ps, mine):
        assert relerr(q, p) < 1e-5 or float((q.cpu() - p).abs().max()) < 1e-7


@pytest.mark.parametrize("wd", [0.0, 0.03])
def test_fused_adam_matches_torch_adam(dev, wd):
    """The teacher trainer's optimizer: torch.optim.Adam (training/train_contrastive_teacher.py:245-248), L2 coupled."""
    from dclip_amd.optim import FusedAdam
    shapes = [(1536, 512), (1536,), (512, 512), (5,)]
    ps = [torch.nn.Parameter(rnd(s, 10 + i)) for i, s in enumerate(shapes)]
    mine = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ps]
    ref_opt = torch.optim.Adam(ps, lr=1e-2, weight_decay=wd)
    my_opt = FusedAdam(mine, lr=1e-2, weight_decay=wd)
    for step in range(4):
        for j, (p, q) in enumerate(zip(ps, mine)):
            g = rnd(p.shape, 200 * step + j, 0.05)
            p.grad, q.grad = g.clone(), g.clone().to(dev)
        ref_opt.step()
        my_opt.step()
    for p, q in zip(ps, mine):
        assert relerr(q, p) < 1e-5 or float((q.cpu() - p).abs().max()) < 1e-7


@pytest.mark.parametrize("M,N,K,split", [(768, 768, 12800, 0), (3072, 768, 12800, 0), (2304, 768, 12800, 4), (132, 64, 77, 1),
                                         (768, 3072, 256, 0), (100, 36, 1000, 3)])
def test_gemm_wgrad_rowsum_is_the_bias_gradient(M, N, K, split):
    """DCLIP_EPI_A_ROWSUM: aux[m] = sum_k A[m,k] from the weight-gradient GEMM's own operand fragments."""
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + N + K)
    dy = torch.randn(K, M, generator=g).to(dev)            # [rows, out]  = A as [K][M]
    x = torch.randn(K, N, generator=g).to(dev)             # [rows, in]
    db = torch.full((M,), float("nan"), device=dev)
    dw = ops.gemm(dy, x, ops.LAYOUT_TN, a_rowsum=db, split_k=split)
    ref_w = dy.double().t() @ x.double()
    ref_b = dy.double().sum(0)
    assert float((dw.double() - ref_w).abs().max() / ref_w.abs().max()) < 1e-5
    assert float((db.double() - ref_b).abs().max() / ref_b.abs().max()) < 1e-5
    assert torch.equal(dw, ops.gemm(dy, x, ops.LAYOUT_TN, split_k=split))        # the product itself is unchanged
    with pytest.raises(ValueError):
        ops.gemm(dy.t().contiguous(), x, ops.LAYOUT_NN, a_rowsum=db)
