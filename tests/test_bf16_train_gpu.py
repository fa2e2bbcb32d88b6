"""bf16 TRAINING path of the student's vision tower (BASELINE configs c3 / c5 quote the step in bf16): helper kernels
against exact references, and the step's loss / gradients against the fp32 path and the fp32 CPU oracle.  bf16 has an
8-bit mantissa: the gate is the loss (1e-3 relative vs the fp32 oracle at real model size) — embedding and gradient
errors are MEASURED and reported, with loose sanity bounds."""
import argparse

import pytest
import torch

from dclip_amd import config as dcfg, synth

pytestmark = pytest.mark.gpu


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("rows,cols", [(64, 64), (85, 132), (12800, 768), (100, 3072), (7, 8), (513, 260)])
@pytest.mark.parametrize("src", ["f32", "bf16"])
def test_transpose_to_bf16(rows, cols, src):
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    x = rnd((rows, cols), rows + cols)
    xin = x.to(dev) if src == "f32" else x.to(torch.bfloat16).to(dev)
    want = x.to(torch.bfloat16)
    if cols % 8 == 0:
        yT, copy = ops.transpose_bf16(xin, want_copy=True)
        assert torch.equal(copy.cpu(), want)
    else:
        yT = ops.transpose_bf16(xin)
    ld = (rows + 7) // 8 * 8
    assert tuple(yT.shape) == (cols, ld)
    assert torch.equal(yT[:, :rows].cpu(), want.t())
    if ld > rows:
        assert float(yT[:, rows:].float().abs().sum()) == 0.0            # zero padded: safe as a GEMM operand tail


def test_rowsum_bf16_and_ln_stats():
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    x = rnd((300, 1288), 5).to(torch.bfloat16)
    got = ops.rowsum_bf16(x.to(dev), 1283).cpu()
    want = x[:, :1283].double().sum(1)
    assert float((got.double() - want).abs().max()) < 1e-3 * float(want.abs().max() + 1)
    h, g, b = rnd((33, 768), 1, 2.0), 1 + rnd((768,), 2, 0.1), rnd((768,), 3, 0.1)
    y, mean, rstd = ops.layernorm_fwd_bf16(h.to(dev), g.to(dev), b.to(dev), 1e-5, save_stats=True)
    y0, mean0, rstd0 = ops.layernorm_fwd(h.to(dev), g.to(dev), b.to(dev), 1e-5)
    assert torch.allclose(mean, mean0, rtol=1e-6, atol=1e-6) and torch.allclose(rstd, rstd0, rtol=1e-6, atol=1e-6)
    assert torch.equal(y, ops.layernorm_fwd_bf16(h.to(dev), g.to(dev), b.to(dev), 1e-5))


@pytest.mark.parametrize("M,N,K", [(400, 768, 768), (12800, 3072, 768), (13, 64, 72)])
def test_gemm_bf16_preact_and_dgelu(M, N, K):
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    a, w, bias = rnd((M, K), 1), rnd((N, K), 2, 0.1), rnd((N,), 3)
    a16, w16 = ops.cast_bf16(a.to(dev)), ops.cast_bf16(w.to(dev))
    pre = a.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().t() + bias.double()
    g16, h16 = ops.gemm_bf16(a16, w16, k=K, bias=bias.to(dev), gelu=True, out_bf16=True, save_preact=True)
    assert float((h16.cpu().double() - pre).abs().max() / pre.abs().max()) < 5e-3            # one bf16 rounding
    hs = h16.cpu().double()
    ref_g = hs * torch.sigmoid(1.702 * hs)                                                     # activation OF THE SAVED value
    assert float((g16.cpu().double() - ref_g).abs().max() / ref_g.abs().max()) < 5e-3
    # dgelu: (dy @ w2^T) * gelu'(h)
    dy, w2 = rnd((M, K), 7), rnd((N, K), 8, 0.1)
    dy16, w2_16 = ops.cast_bf16(dy.to(dev)), ops.cast_bf16(w2.to(dev))
    got = ops.gemm_bf16(dy16, w2_16, k=K, dgelu_of=h16).cpu().double()
    s = torch.sigmoid(1.702 * hs)
    want = (dy.to(torch.bfloat16).double() @ w2.to(torch.bfloat16).double().t()) * (s * (1 + 1.702 * hs * (1 - s)))
    assert float((got - want).abs().max() / want.abs().max()) < 1e-5 * max(1.0, K ** 0.5)


@pytest.mark.parametrize("M,N,K,splits", [(768, 768, 12800, None), (768, 3072, 12800, None), (2304, 768, 12800, 7),
                                           (128, 256, 85, None), (132, 64, 4100, 3), (768, 768, 12800, 1),
                                           (3072, 768, 12800, None), (768, 768, 3200, None), (300, 520, 1280, 3),
                                           (1024, 4096, 3200, 5)])
def test_gemm_bf16_wgrad_split_k(M, N, K, splits):
    """dW = dY^T X on token-contiguous bf16 operands, split-K with a fixed-order reduce (deterministic)."""
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    dy, x = rnd((K, M), 1), rnd((K, N), 2)
    dyT, xT = ops.transpose_bf16(dy.to(dev)), ops.transpose_bf16(x.to(dev))
    got = ops.gemm_bf16_wgrad(dyT, xT, K, splits)
    want = dy.to(torch.bfloat16).double().t() @ x.to(torch.bfloat16).double()
    assert float((got.double().cpu() - want).abs().max() / want.abs().max()) < 2e-6 * max(1.0, K ** 0.5)
    assert torch.equal(got, ops.gemm_bf16_wgrad(dyT, xT, K, splits))


def _named_grads(model):
    return {n: p.grad.detach().double().cpu().reshape(-1) for n, p in model.named_parameters() if p.grad is not None}


def _step(model, pix, ids, t_img, precision):
    from dclip_amd import functional
    for p in model.parameters():
        p.grad = None
    img = model.get_image_features(pixel_values=pix, precision=precision)
    with torch.no_grad():
        txt = model.get_text_features(input_ids=ids)
    loss = functional.cosine_distillation_loss(img, t_img) + functional.contrastive_loss(img, txt)
    loss.backward()
    return float(loss.detach()), img.detach().clone(), _named_grads(model)


@pytest.mark.parametrize("name,mk,B", [("tiny", dcfg.tiny, 6), ("ViT-B/32", dcfg.vit_b32, 8), ("ViT-B/32", dcfg.vit_b32, 256)])
def test_bf16_training_step_against_fp32(name, mk, B):
    """Same weights, same batch: the bf16-GEMM step vs the exact fp32 step of the same library (which the c1 / c2
    goldens pin to the reference).  B = 256 is the benched shape: 12,800 tokens put the backward on its DEFAULT schedule
    (token-major weight gradients, bf16 I/O around the attention kernels, bias gradients out of LayerNorm's backward); the
    small batches take the transposing schedule with fp32 attention I/O."""
    from dclip_amd.clip_model import from_hf_state_dict
    dev = torch.device("cuda:0")
    cfg = mk()
    m = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=0, gain=3.0), device=dev)
    for p in m.text_model.parameters():
        p.requires_grad = False
    m.text_projection.weight.requires_grad = False
    m.logit_scale.requires_grad = False
    pix = synth.synth_pixel_values(B, cfg.vision, seed=0).to(dev)
    ids = synth.synth_input_ids(B, cfg.text, seed=3, ragged=True).to(dev)
    t_img = synth.synth_embeddings(B, cfg.projection_dim, seed=1).to(dev)
    l32, e32, g32 = _step(m, pix, ids, t_img, "fp32")
    l16, e16, g16 = _step(m, pix, ids, t_img, "bf16")
    l16b, _, g16b = _step(m, pix, ids, t_img, "bf16")
    assert l16 == l16b and all(torch.equal(g16[k], g16b[k]) for k in g16)          # deterministic
    assert set(g16) == set(g32)
    cos = {k: float(g16[k] @ g32[k] / (g16[k].norm() * g32[k].norm()).clamp_min(1e-30)) for k in g32}
    nrm = {k: float(g16[k].norm() / g32[k].norm().clamp_min(1e-30)) for k in g32}
    worst = sorted(cos.items(), key=lambda kv: kv[1])[:3]
    emb_rel = float((e16 - e32).abs().max() / e32.abs().max())
    emb_cos = float(torch.nn.functional.cosine_similarity(e16, e32, dim=1).min())
    print(f"[{name} B={B}] loss fp32 {l32:.6f} bf16 {l16:.6f} (rel {abs(l16 - l32) / abs(l32):.2e}); embedding max rel {emb_rel:.2e}, "
          f"min cos {emb_cos:.6f}; grad cosine min {min(cos.values()):.5f} median {sorted(cos.values())[len(cos) // 2]:.5f}; "
          f"grad norm ratio {min(nrm.values()):.3f}..{max(nrm.values()):.3f}; worst {worst}")
    assert abs(l16 - l32) <= 2e-3 * abs(l32)
    assert emb_cos > 0.999
    assert min(cos.values()) > 0.98 and 0.9 < min(nrm.values()) and max(nrm.values()) < 1.1


def test_bf16_student_c3_step_loss_vs_oracle():
    """The gate for the bf16 configs: one c3-shaped step (ViT-B/32 student in bf16, teacher embedding given) — loss within
    1e-3 relative of the fp32 CPU oracle; the optimizer then moves the fp32 master weights and the bf16 copies follow."""
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.optim import FusedAdamW
    from oracle import dclip_oracle as O
    dev = torch.device("cuda:0")
    cfg = dcfg.vit_b32()
    sd = synth.synth_clip_state_dict(cfg, seed=0, gain=3.0)
    student = from_hf_state_dict(cfg, sd, device=dev)
    hp = argparse.Namespace(learning_rate=1e-4, warmup_steps=0, total_steps=10, train_batch_size=4, eval_batch_size=4)
    mod = CLIPImageDistillation(hp, student, None, freeze_mode="north_star", student_precision="bf16").to(dev)
    B = 4
    pix = synth.synth_pixel_values(B, cfg.vision, seed=0)
    ids = synth.synth_input_ids(B, cfg.text, seed=3, ragged=True, min_len=8)
    t_img = synth.synth_embeddings(B, cfg.projection_dim, seed=1)
    batch = {"pixel_values": pix.to(dev), "input_ids": ids.to(dev), "teacher_image_emb": t_img.to(dev)}
    loss = mod.training_step(batch)
    loss.backward()
    with torch.no_grad():
        ref = O.distill_step(sd, cfg, pix, ids, t_img)
    got, want = float(loss.detach()), float(ref["loss"])
    print(f"c3-shaped bf16 student step: loss {got:.6f} vs fp32 oracle {want:.6f} (rel {abs(got - want) / abs(want):.2e})")
    assert abs(got - want) <= 1e-3 * abs(want)
    opt = FusedAdamW([p for p in mod.parameters() if p.requires_grad], lr=1e-3, max_grad_norm=0.5)
    opt.step()
    opt.zero_grad(set_to_none=True)
    loss2 = mod.training_step(batch)
    assert float(loss2.detach()) < got            # the update was seen by the bf16 weight copies (version-keyed cache)


@pytest.mark.parametrize("M,N,K", [(768, 768, 12800), (768, 3072, 12800), (3072, 768, 12800), (2304, 768, 12800), (768, 768, 3200),
                                   (264, 520, 1280), (1024, 4096, 3200), (776, 1032, 25600)])
def test_gemm_bf16_wgrad_tokmajor(M, N, K):
    """dW = dY^T X straight from the token-major bf16 operands (no transposes): against fp64 of the rounded inputs, equal
    to itself on repetition, and within fp32-accumulation noise of the transposing path."""
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    dy, x = rnd((K, M), 1).to(torch.bfloat16), rnd((K, N), 2).to(torch.bfloat16)
    got = ops.gemm_bf16_wgrad_tokmajor(dy.to(dev), x.to(dev))
    if got is None:                       # the library declined the shape (too few work items for 256 CUs)
        from dclip_amd import _lib
        assert _lib.load().dclip_gemm_bf16_wgrad_tokmajor_plan(M, N, K) == 0 and (M, N, K) in ((768, 768, 3200), (264, 520, 1280))
        return
    want = dy.double().t() @ x.double()
    assert float((got.double().cpu() - want).abs().max() / want.abs().max()) < 2e-6 * max(1.0, K ** 0.5)
    assert torch.equal(got, ops.gemm_bf16_wgrad_tokmajor(dy.to(dev), x.to(dev)))
    other = ops.gemm_bf16_wgrad(ops.transpose_bf16(dy.to(dev)), ops.transpose_bf16(x.to(dev)), K)
    assert float((got - other).abs().max() / want.abs().max()) < 2e-6 * max(1.0, K ** 0.5)


def test_colsum_bf16():
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    x = rnd((12800, 3072), 5).to(torch.bfloat16)
    got = ops.colsum_bf16(x.to(dev))
    want = x.double().sum(0)
    assert float((got.double().cpu() - want).abs().max() / want.abs().max()) < 1e-5


def test_bf16_layer_backward_schedules_agree(monkeypatch):
    """The token-major weight-gradient schedule (default; with it q/k/v, the attention output and their gradients travel as
    bf16) and the transposing one (DCLIP_BF16_WGRAD_TN=0; fp32 attention I/O) give the same gradients up to fp32
    accumulation order and those extra bf16 roundings."""
    from dclip_amd.clip_model import from_hf_state_dict
    dev = torch.device("cuda:0")
    cfg = dcfg.vit_b32()
    sd = synth.synth_clip_state_dict(cfg, seed=3)
    B = 256                                                          # 12,800 tokens: the token-major form applies to all four shapes
    pix = synth.synth_pixel_values(B, cfg.vision, seed=0).to(dev)
    ids = synth.synth_input_ids(B, cfg.text, seed=3, ragged=True).to(dev)
    t_img = synth.synth_embeddings(B, cfg.projection_dim, seed=1).to(dev)
    grads = []
    from dclip_amd import engine
    for flag in ("1", "0"):
        monkeypatch.setenv("DCLIP_BF16_WGRAD_TN", flag)
        engine._TOKMAJOR_PLAN.clear()                                # the schedule choice is cached per (M, D, I)
        m = from_hf_state_dict(cfg, sd, device=dev)
        for p_ in m.text_model.parameters():
            p_.requires_grad = False
        m.text_projection.weight.requires_grad = False
        m.logit_scale.requires_grad = False
        _loss, _img, g = _step(m, pix, ids, t_img, "bf16")
        grads.append(g)
    worst = 1.0
    for n in grads[0]:
        a, b = grads[0][n], grads[1][n]
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        worst = min(worst, cos)
        assert cos > 0.99995, (n, cos)
    print(f"token-major + bf16 attention I/O vs transposing + fp32 attention I/O: min gradient cosine {worst:.7f}")


def test_bf16_student_under_graph_replay_follows_the_optimizer():
    """the bf16 weight copies must be refreshed INSIDE the captured step.  Replay, optimizer step,
    replay — the second replay's loss and gradients equal the eager module's second step (same kernels, same order:
    bit-identical); before the fix it multiplied by the copies made during the capture's warm-up."""
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.graph import GraphedStep
    from dclip_amd.optim import FusedAdamW
    dev = torch.device("cuda:0")
    cfg = dcfg.tiny()
    B = 6

    def make():
        student = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=0, gain=3.0), device=dev)
        hp = argparse.Namespace(learning_rate=1e-2, warmup_steps=0, total_steps=100, train_batch_size=B, eval_batch_size=B)
        mod = CLIPImageDistillation(hp, student, None, freeze_mode="north_star", student_precision="bf16").to(dev)
        return mod, FusedAdamW([p for p in mod.parameters() if p.requires_grad], lr=1e-2, max_grad_norm=0.5)

    batch = {"pixel_values": synth.synth_pixel_values(B, cfg.vision, seed=20).to(dev),
             "input_ids": synth.synth_input_ids(B, cfg.text, seed=21, ragged=True).to(dev),
             "teacher_image_emb": synth.synth_embeddings(B, cfg.projection_dim, seed=22).to(dev)}
    (eager, opt_e), (graphed, opt_g) = make(), make()
    g = GraphedStep(graphed, batch)
    names = [n for n, p in eager.named_parameters() if p.requires_grad]
    losses = []
    for it in range(3):
        for p in eager.parameters():
            p.grad = None
        le = eager.training_step(batch)
        le.backward()
        lg = g.step(batch)
        assert torch.equal(le.detach(), lg.detach()), (it, float(le), float(lg))
        ge, gg = dict(eager.named_parameters()), dict(graphed.named_parameters())
        for n in names:
            assert torch.equal(ge[n].grad, gg[n].grad), (it, n)
        losses.append(float(le.detach()))
        opt_e.step()
        opt_g.step()
    assert losses[1] != losses[0] and losses[2] != losses[1]          # the updates are visible through the bf16 copies
    for n in names:
        assert torch.equal(dict(eager.named_parameters())[n], dict(graphed.named_parameters())[n]), n


@pytest.mark.parametrize("B,S,H,causal", [(8, 50, 12, False), (3, 64, 2, False), (5, 17, 3, False), (4, 50, 8, True), (2, 1, 1, False)])
def test_attention_io16_equals_the_fp32_kernels_on_rounded_operands(B, S, H, causal):
    """The bf16-I/O attention kernels are the fp32 kernels with bf16 loads and stores: on operands that ARE bf16 values
    they must reproduce the fp32 kernels' results rounded to bf16 (same products, same order)."""
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    D = 64 * H
    qkv16 = (rnd((B * S, 3 * D), 11) * 1.5).to(torch.bfloat16).to(dev)
    out16, lse16 = ops.attention_fwd_io16(qkv16, B, S, H, causal)
    out32, lse32 = ops.attention_fwd(qkv16.float(), B, S, H, causal)
    assert torch.equal(lse16.reshape(-1), lse32.reshape(-1))
    assert torch.equal(out16, out32.to(torch.bfloat16))
    dout16 = rnd((B * S, D), 12).to(torch.bfloat16).to(dev)
    dq16 = ops.attention_bwd_io16(qkv16, out16, dout16, lse16, B, S, H, causal)
    dq32 = ops.attention_bwd(qkv16.float(), out16.float(), dout16.float(), lse16, B, S, H, causal)
    assert dq16.dtype == torch.bfloat16 and torch.equal(dq16, dq32.to(torch.bfloat16))
    # and against an fp64 attention of the same rounded operands
    q, k, v = [t.reshape(B, S, H, 64).permute(0, 2, 1, 3).double().cpu() for t in qkv16.float().split(D, dim=1)]
    sc = q @ k.transpose(-1, -2) * 0.125
    if causal:
        sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
    ref = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B * S, D)
    assert float((out16.double().cpu() - ref).abs().max() / ref.abs().max()) < 6e-3           # one bf16 rounding of the output


def test_layernorm_bwd_ex_bf16_copy_and_column_sums():
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    rows, D = 12800, 768
    x, dy, res = rnd((rows, D), 1, 2.0).to(dev), rnd((rows, D), 2).to(dev), rnd((rows, D), 3).to(dev)
    g, b = (1.0 + 0.1 * rnd((D,), 4)).to(dev), rnd((D,), 5).to(dev)
    _, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
    dx0, dg0, db0 = ops.layernorm_bwd(dy, x, g, mean, rstd, dresidual=res)
    cs = torch.empty(D, dtype=torch.float32, device=dev)
    dx1, dg1, db1, dx16 = ops.layernorm_bwd(dy, x, g, mean, rstd, dresidual=res, want_bf16=True, dx_colsum=cs)
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert dx16.dtype == torch.bfloat16 and torch.equal(dx16, dx0.to(torch.bfloat16))
    want = dx0.double().sum(0)
    assert float((cs.double() - want).abs().max() / want.abs().max()) < 1e-5
    # without parameter gradients (a frozen LayerNorm): the column sums alone
    cs2 = torch.empty_like(cs)
    dx2, _, _ = ops.layernorm_bwd(dy, x, g, mean, rstd, dresidual=res, need_param_grads=False, dx_colsum=cs2)
    assert torch.equal(dx2, dx0) and torch.equal(cs2, cs)


def test_mt_weights_bf16_equals_the_per_weight_kernels():
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072), (72, 40), (8, 8)]
    ws = [rnd(s_, 20 + i).to(dev) for i, s_ in enumerate(shapes)]
    recs = []
    for w in ws:
        recs.append((w, torch.zeros_like(ops.cast_bf16(w)), torch.zeros_like(ops.transpose_bf16(w))))
    table, n, tiles = ops.mt_weights_table(recs)
    ops.mt_weights_bf16(table, n, tiles)
    for w, w16, w16T in recs:
        assert torch.equal(w16, ops.cast_bf16(w)) and torch.equal(w16T, ops.transpose_bf16(w))
    # a record with only one of the two outputs
    w = ws[0]
    only = torch.zeros_like(ops.transpose_bf16(w))
    table, n, tiles = ops.mt_weights_table([(w, None, only)])
    ops.mt_weights_bf16(table, n, tiles)
    assert torch.equal(only, ops.transpose_bf16(w))


@pytest.mark.parametrize("B,S,H,causal", [(8, 50, 12, False), (3, 64, 2, False), (5, 17, 3, False), (4, 50, 8, True), (2, 1, 1, False),
                                          (3, 33, 2, True)])
def test_attention_bf16_mfma_training_pair(B, S, H, causal):
    """bf16 MFMA attention forward (+ lse) and backward for the training student: against an fp64 attention of the same
    bf16-valued operands and its autograd gradients, and against the fp32-arithmetic bf16-I/O kernels."""
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    D = 64 * H
    qkv16 = (rnd((B * S, 3 * D), 21) * 1.5).to(torch.bfloat16).to(dev)
    dout16 = rnd((B * S, D), 22).to(torch.bfloat16).to(dev)
    out16, lse = ops.attention_fwd_bf16_lse(qkv16, B, S, H, causal)
    assert torch.equal(out16, ops.attention_fwd_bf16(qkv16, B, S, H, causal))           # same kernel with / without the lse output
    dq16 = ops.attention_bwd_bf16(qkv16, out16, dout16, lse, B, S, H, causal)
    assert torch.equal(dq16, ops.attention_bwd_bf16(qkv16, out16, dout16, lse, B, S, H, causal))     # deterministic
    # fp64 reference
    x = qkv16.double().cpu().requires_grad_(True)
    q, k, v = [t.reshape(B, S, H, 64).permute(0, 2, 1, 3) for t in x.split(D, dim=1)]
    sc = q @ k.transpose(-1, -2) * 0.125
    if causal:
        sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
    ref = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B * S, D)
    ref.backward(dout16.double().cpu())
    ref_lse = torch.logsumexp(sc, -1).reshape(B * H, S)
    assert float((lse.double().cpu() - ref_lse).abs().max()) < 2e-3
    assert float((out16.double().cpu() - ref.detach()).abs().max() / ref.detach().abs().max()) < 1.5e-2
    g, w = dq16.double().cpu(), x.grad
    for part, name in enumerate(("dq", "dk", "dv")):
        a, b_ = g[:, part * D:(part + 1) * D].reshape(-1), w[:, part * D:(part + 1) * D].reshape(-1)
        if float(b_.abs().max()) < 1e-9:          # S = 1: the softmax is constant, dq = dk = 0 exactly; the kernel leaves rounding noise
            assert float(a.abs().max()) < 1e-3, name
            continue
        cos = float(a @ b_ / (a.norm() * b_.norm()).clamp_min(1e-30))
        rel = float((a - b_).abs().max() / b_.abs().max().clamp_min(1e-30))
        assert cos > 0.9998 and rel < 3e-2, (name, cos, rel)
    # and the fp32-arithmetic kernels on the same operands (they differ by the bf16 rounding of P and dS only)
    o2, l2 = ops.attention_fwd_io16(qkv16, B, S, H, causal)
    assert float((out16.float() - o2.float()).abs().max() / o2.float().abs().max()) < 1.5e-2
    assert float((lse - l2.reshape(B * H, S)).abs().max()) < 2e-3
    if S <= 64:
        d2 = ops.attention_bwd_io16(qkv16, o2, dout16, l2, B, S, H, causal).float()[:, 2 * D:].reshape(-1)      # dv (never zero)
        a = dq16.float()[:, 2 * D:].reshape(-1)
        assert float(a @ d2 / (a.norm() * d2.norm())) > 0.9998


def test_bf16_patch_embedding_runs_on_the_bf16_mfmas(monkeypatch):
    """The bf16 student's patch embedding (convolution as a GEMM) and its weight gradient on the bf16 kernels (default at
    the benched shape) against the same step with that one GEMM pair in fp32 (DCLIP_BF16_PATCH=0), and against the all-fp32
    step: the loss moves by < 5e-4, no gradient tensor turns (the gates of test_bf16_training_step_against_fp32 hold for
    both)."""
    from dclip_amd import engine
    from dclip_amd.clip_model import from_hf_state_dict
    dev = torch.device("cuda:0")
    cfg = dcfg.vit_b32()
    B = 256
    m = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=0, gain=3.0), device=dev)
    for p in m.text_model.parameters():
        p.requires_grad = False
    m.text_projection.weight.requires_grad = False
    m.logit_scale.requires_grad = False
    pix = synth.synth_pixel_values(B, cfg.vision, seed=0).to(dev)
    ids = synth.synth_input_ids(B, cfg.text, seed=3, ragged=True).to(dev)
    t_img = synth.synth_embeddings(B, cfg.projection_dim, seed=1).to(dev)
    l32, e32, g32 = _step(m, pix, ids, t_img, "fp32")
    monkeypatch.setenv("DCLIP_BF16_PATCH", "0")
    engine._PATCH_BF16_PLAN.clear()
    l_off, e_off, g_off = _step(m, pix, ids, t_img, "bf16")
    assert list(engine._PATCH_BF16_PLAN.values()) == [False]
    monkeypatch.delenv("DCLIP_BF16_PATCH")
    engine._PATCH_BF16_PLAN.clear()
    l_on, e_on, g_on = _step(m, pix, ids, t_img, "bf16")
    assert list(engine._PATCH_BF16_PLAN.values()) == [True]                 # 12,544 patch rows: the bf16 path is the default
    key = "vision_model.embeddings.patch_embedding.weight"
    cos = lambda a, b: float(a @ b / (a.norm() * b.norm()).clamp_min(1e-30))
    c_on = {k: cos(g_on[k], g32[k]) for k in g32}
    c_off = {k: cos(g_off[k], g32[k]) for k in g32}
    print(f"patch embedding bf16: loss fp32 {l32:.6f} | bf16 layers only {l_off:.6f} | + bf16 patch GEMMs {l_on:.6f}; patch-weight "
          f"gradient cosine vs fp32 {c_off[key]:.6f} -> {c_on[key]:.6f}; min over tensors {min(c_off.values()):.5f} -> {min(c_on.values()):.5f}; "
          f"embedding min cos {float(torch.nn.functional.cosine_similarity(e_on, e32, dim=1).min()):.6f}")
    assert abs(l_on - l_off) <= 5e-4 * abs(l32) and abs(l_on - l32) <= 2e-3 * abs(l32)
    assert c_on[key] > 0.999 and min(c_on.values()) > 0.98
    assert float(torch.nn.functional.cosine_similarity(e_on, e32, dim=1).min()) > 0.999
