"""Opt-in bf16 path for frozen towers: GEMM kernel vs a bf16-rounded fp64 reference, and the tower's measured error."""
import pytest
import torch

from dclip_amd import config as dcfg, synth

pytestmark = pytest.mark.gpu


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (400, 768, 768), (257, 132, 588), (2048, 3072, 768), (13, 64, 72),
                                   (12800, 768, 3072), (128, 256, 85), (64, 64, 21), (130, 132, 149)])   # odd K too
def test_gemm_bf16_matches_rounded_inputs(M, N, K):
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    a, w, bias = rnd((M, K), 1), rnd((N, K), 2, 0.1), rnd((N,), 3)
    if K % 4 == 0:
        a16, w16 = ops.cast_bf16(a.to(dev)), ops.cast_bf16(w.to(dev))
    else:            # the cast kernel takes rows of a multiple of 4 floats: pad on the host (zero tail, ld % 8 == 0)
        ld = (K + 7) // 8 * 8
        a16 = torch.zeros(M, ld, dtype=torch.bfloat16)
        w16 = torch.zeros(N, ld, dtype=torch.bfloat16)
        a16[:, :K], w16[:, :K] = a.to(torch.bfloat16), w.to(torch.bfloat16)
        a16, w16 = a16.to(dev), w16.to(dev)
    assert a16.shape[1] % 8 == 0 and torch.equal(a16[:, :K].cpu(), a.to(torch.bfloat16))
    if a16.shape[1] > K:
        assert float(a16[:, K:].float().abs().sum()) == 0.0
    want = a.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().t() + bias.double()
    got = ops.gemm_bf16(a16, w16, k=K, bias=bias.to(dev))
    err = float((got.double().cpu() - want).abs().max() / want.abs().max())
    assert err < 2e-6 * max(1.0, K ** 0.5), err              # only fp32 accumulation error is left
    res = rnd((M, N), 4)
    got2 = ops.gemm_bf16(a16, w16, k=K, bias=bias.to(dev), residual=res.to(dev))
    assert float((got2.double().cpu() - (want + res.double())).abs().max() / want.abs().max()) < 1e-5
    g16 = ops.gemm_bf16(a16, w16, k=K, bias=bias.to(dev), gelu=True, out_bf16=True)
    ref = (want * torch.sigmoid(1.702 * want)).to(torch.bfloat16)
    assert float((g16.cpu().double() - ref.double()).abs().max() / ref.double().abs().max()) < 1e-2


def test_layernorm_bf16_and_cast():
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    x, g, b = rnd((33, 768), 1, 2.0), 1 + rnd((768,), 2, 0.1), rnd((768,), 3, 0.1)
    y = ops.layernorm_fwd_bf16(x.to(dev), g.to(dev), b.to(dev), 1e-5)
    want = torch.nn.functional.layer_norm(x.double(), (768,), g.double(), b.double(), 1e-5)
    assert float((y.cpu().double() - want).abs().max()) < 2e-2
    assert y.dtype == torch.bfloat16


@pytest.mark.parametrize("name,mk", [("tiny", dcfg.tiny), ("ViT-B/32", dcfg.vit_b32)])
def test_frozen_vision_tower_bf16_error(name, mk):
    """Measured, reported error of the opt-in bf16 tower against the fp32 tower (same weights, same inputs)."""
    from dclip_amd.clip_model import from_hf_state_dict
    dev = torch.device("cuda:0")
    cfg = mk()
    m = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=0, gain=3.0), device=dev)
    pix = synth.synth_regions(4, 1, cfg.vision, seed=2)[:, 0].to(dev)
    with torch.no_grad():
        f32 = m.get_image_features(pixel_values=pix)
        b16 = m.get_image_features(pixel_values=pix, precision="bf16")
        b16_again = m.get_image_features(pixel_values=pix, precision="bf16")
    assert torch.equal(b16, b16_again)
    rel = float((b16 - f32).abs().max() / f32.abs().max())
    cos = torch.nn.functional.cosine_similarity(b16, f32, dim=1).min()
    print(f"{name}: bf16 tower max rel err {rel:.2e}, min cosine {float(cos):.6f}")
    assert rel < 3e-2 and float(cos) > 0.999
    # with gradients enabled and trainable parameters the same call is the bf16 TRAINING path (tests/test_bf16_train_gpu.py)
    assert m.get_image_features(pixel_values=pix, precision="bf16").requires_grad


def test_frozen_text_tower_bf16_error():
    from dclip_amd.clip_model import from_hf_state_dict
    dev = torch.device("cuda:0")
    cfg = dcfg.vit_b32()
    m = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=0, gain=3.0), device=dev)
    ids = synth.synth_input_ids(6, cfg.text, seed=3, ragged=True).to(dev)
    with torch.no_grad():
        f32 = m.get_text_features(input_ids=ids)
        b16 = m.get_text_features(input_ids=ids, precision="bf16")
        s32, t32, e32 = m.text_token_level(ids)
        s16, t16, e16 = m.text_token_level(ids, precision="bf16")
    assert torch.equal(e32, e16)
    for nm, a, b in (("sentence", f32, b16), ("sentence(token pass)", s32, s16)):
        rel = float((a - b).abs().max() / a.abs().max())
        cos = float(torch.nn.functional.cosine_similarity(a, b, dim=1).min())
        print(f"text {nm}: bf16 max rel err {rel:.2e}, min cosine {cos:.6f}")
        assert rel < 3e-2 and cos > 0.999
    # word tokens that the teacher reads: rows 1..eos-1 of each caption
    for b in range(ids.shape[0]):
        n = int(e32[b])
        cos = torch.nn.functional.cosine_similarity(t32[b, 1:n], t16[b, 1:n], dim=1)
        assert float(cos.min()) > 0.999


def test_meta_teacher_bf16_towers_close_to_fp32():
    """compute_global_embedding with tower_precision='bf16' vs 'fp32': same cross-attention weights."""
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    dev = torch.device("cuda:0")
    cfg = dcfg.vit_b32()
    clip = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=0, gain=3.0), device=dev)
    E = cfg.projection_dim
    outs = {}
    for prec in ("fp32", "bf16"):
        t = PatchTextAggregation(embed_dim=E, num_heads=E // 64, clip_model=clip, tower_precision=prec).to(dev)
        t.cross_modal_attention.load_state_dict(synth.synth_cross_modal_state_dict(E, seed=5))
        regions = synth.synth_regions(3, 4, cfg.vision, seed=2).to(dev)
        ids = synth.synth_input_ids(3, cfg.text, seed=3, ragged=True).to(dev)
        with torch.no_grad():
            outs[prec] = t.compute_global_embedding_tensors(regions, ids, torch.tensor([4, 2, 0], dtype=torch.int32))
    cos = torch.nn.functional.cosine_similarity(outs["fp32"], outs["bf16"], dim=1)
    print("meta-teacher bf16 towers: min cosine", float(cos.min()))
    assert float(cos.min()) > 0.999


@pytest.mark.parametrize("pingpong", ["1", "0"])
def test_gemm_bf16_big_tile_kernel_matches(pingpong):
    """The 256x256 LDS-DMA kernels (ping-pong schedule, and the lock-step one behind DCLIP_BF16_PP=0), forced onto small
    shapes (DCLIP_BF16_BIG_MIN=1 is read once per process: this test runs the comparison in a child process), against
    the fp64 product of the rounded inputs.  K = 64 .. 3072 covers 1, 2, 3 (odd), 12 and 48 K-tiles."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent("""
        import torch, sys
        sys.path.insert(0, %r)
        from dclip_amd import ops
        dev = torch.device("cuda:0")
        for M, N, K in [(256, 256, 64), (1000, 520, 128), (257, 260, 192), (2048, 3072, 768), (4100, 768, 3072), (77, 768, 768),
                        (511, 508, 320), (300, 1028, 256)]:
            g = torch.Generator().manual_seed(M + N + K)
            a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1
            bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
            a16, w16 = ops.cast_bf16(a.to(dev)), ops.cast_bf16(w.to(dev))
            want = a.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().t() + bias.double()
            got = ops.gemm_bf16(a16, w16, k=K, bias=bias.to(dev))
            err = float((got.double().cpu() - want).abs().max() / want.abs().max())
            assert err < 2e-6 * max(1.0, K ** 0.5), (M, N, K, err)
            got2 = ops.gemm_bf16(a16, w16, k=K, bias=bias.to(dev), residual=res.to(dev))
            assert float((got2.double().cpu() - (want + res.double())).abs().max() / want.abs().max()) < 1e-5
            g16 = ops.gemm_bf16(a16, w16, k=K, bias=bias.to(dev), gelu=True, out_bf16=True)
            ref = (want * torch.sigmoid(1.702 * want)).to(torch.bfloat16)
            assert float((g16.cpu().double() - ref.double()).abs().max() / ref.double().abs().max()) < 1e-2
            # training epilogues: GELU that also saves the bf16 pre-activation; product times quick_gelu'(saved)
            y, h = ops.gemm_bf16(a16, w16, k=K, bias=bias.to(dev), gelu=True, save_preact=True)
            h_ref = want.to(torch.bfloat16)
            assert float((h.cpu().double() - h_ref.double()).abs().max() / h_ref.double().abs().max()) < 1e-2
            hd = h.double().cpu()
            assert float((y.double().cpu() - hd * torch.sigmoid(1.702 * hd)).abs().max()) < 1e-4 * float(hd.abs().max())
            sg = torch.sigmoid(1.702 * hd)
            dref = (want - bias.double()) * (sg * (1.0 + 1.702 * hd * (1.0 - sg)))
            for o16 in (False, True):
                d = ops.gemm_bf16(a16, w16, k=K, dgelu_of=h, out_bf16=o16)
                tol = 1e-2 if o16 else 1e-5
                assert float((d.double().cpu() - dref).abs().max() / dref.abs().max()) < tol, (M, N, K, o16)
        print("OK")
    """) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DCLIP_BF16_BIG_MIN="1", DCLIP_BF16_PP=pingpong)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


@pytest.mark.parametrize("B,S,H,causal", [(2, 50, 2, False), (2, 77, 2, True), (1, 257, 2, False), (1, 197, 1, False),
                                          (3, 64, 1, True), (2, 1, 1, False), (1, 130, 2, True), (2, 33, 3, False),
                                          (2, 288, 1, False), (1, 288, 2, True), (1, 289, 1, False), (2, 32, 2, True),
                                          (1, 320, 2, True), (3, 96, 2, False), (3, 257, 3, False), (2, 257, 2, True)])     # <= 288: whole-head kernel (257: shared last query), beyond: tiled
def test_attention_fwd_bf16(B, S, H, causal):
    """bf16 q/k/v, fp32 softmax, bf16 P and output: against an fp64 attention of the same rounded inputs."""
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    D = H * 64
    qkv = (rnd((B * S, 3 * D), 7 + S, 1.2)).to(torch.bfloat16)
    q, k, v = (qkv.double().view(B, S, 3, H, 64)[:, :, i].transpose(1, 2) for i in range(3))
    s = (q @ k.transpose(-1, -2)) * 0.125
    if causal:
        s = s + torch.full((S, S), float("-inf"), dtype=torch.float64).triu(1)
    want = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B * S, D)
    got = ops.attention_fwd_bf16(qkv.to(dev), B, S, H, causal)
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == (B * S, D)
    err = float((got.double().cpu() - want).abs().max() / want.abs().max())
    assert err < 1.5e-2, err          # bf16 rounding of P (2^-9 relative) and of the output


@pytest.mark.parametrize("B,S,p", [(3, 64, 16), (2, 224, 32), (5, 32, 4)])
def test_im2col_bf16_equals_im2col_then_round(B, S, p):
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    pix = rnd((B, 3, S, S), 11).to(dev)
    want = ops.im2col(pix, p).to(torch.bfloat16)
    got = ops.im2col_bf16(pix, p)
    assert got.dtype == torch.bfloat16 and torch.equal(got[:, :want.shape[1]], want)
    ref = pix.cpu().unfold(2, p, p).unfold(3, p, p).permute(0, 2, 3, 1, 4, 5).reshape(B * (S // p) ** 2, 3 * p * p)
    assert torch.equal(ops.im2col(pix, p).cpu(), ref)                    # the vectorised fp32 path


def test_bf16_weight_cache_follows_the_fused_optimizer():
    """FusedAdamW updates parameters through raw pointers; the bf16 weight copies (keyed on tensor versions) must be
    rebuilt afterwards."""
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.optim import FusedAdamW
    dev = torch.device("cuda:0")
    cfg = dcfg.tiny()
    m = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=0, gain=3.0), device=dev)
    pix = synth.synth_pixel_values(3, cfg.vision, seed=2).to(dev)
    with torch.no_grad():
        before = m.get_image_features(pixel_values=pix, precision="bf16")
    params = [p for p in m.vision_model.parameters()] + [m.visual_projection.weight]
    m.get_image_features(pixel_values=pix).square().sum().backward()
    FusedAdamW(params, lr=1e-2).step()
    with torch.no_grad():
        after16 = m.get_image_features(pixel_values=pix, precision="bf16")
        after32 = m.get_image_features(pixel_values=pix)
    assert float((after16 - before).abs().max()) > 1e-3                      # the update is visible on the bf16 path
    assert float((after16 - after32).abs().max() / after32.abs().max()) < 3e-2


def test_pingpong_gemm_repeat_launches_are_bit_identical():
    """Race screen of the counted-vmcnt / raw-barrier schedule (tools/bf16_gemm_race_screen.py): every shape launched many
    times, alone and beside a bandwidth-heavy copy on another stream; every result bit-identical to the first, the first
    checked against fp64."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "bf16_gemm_race_screen.py"), "40"], capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0 and "RACE SCREEN clean" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


@pytest.mark.parametrize("M,N,K", [(1000, 520, 64), (2048, 1024, 128), (777, 260, 192), (5000, 768, 768), (256, 256, 64),
                                   (4096, 2304, 768)])
def test_persistent_pingpong_gemm_against_the_one_tile_kernel(M, N, K, monkeypatch):
    """gemm_bf16_ppp_kernel (a workgroup walks several tiles; C leaves straight from the accumulators, the residual is the
    accumulators' initial value): forced on for small problems here (it normally serves >= 512 tiles), edge tiles in M and
    N, one / two / many K-tiles, every epilogue it takes.  Against the one-tile ping-pong kernel: bit-identical for bias /
    GELU / saved pre-activation (same products, same order), fp32-rounding-close for the residual (added first, not last);
    and against fp64 of the rounded operands."""
    import os
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    a16, w16 = rnd((M, K), 1).to(torch.bfloat16).to(dev), rnd((N, K), 2, 0.1).to(torch.bfloat16).to(dev)
    bias, res = rnd((N,), 3).to(dev), rnd((M, N), 4).to(dev)
    want = a16.double().cpu() @ w16.double().cpu().t() + bias.double().cpu()

    def run():
        return (ops.gemm_bf16(a16, w16, bias=bias), ops.gemm_bf16(a16, w16), ops.gemm_bf16(a16, w16, bias=bias, out_bf16=True),
                ops.gemm_bf16(a16, w16, bias=bias, residual=res), ops.gemm_bf16(a16, w16, bias=bias, gelu=True, out_bf16=True),
                ops.gemm_bf16(a16, w16, bias=bias, gelu=True, out_bf16=True, save_preact=True))

    monkeypatch.setenv("DCLIP_BF16_BIG_MIN", "1")         # both runs below take the 256x256 kernels
    monkeypatch.setenv("DCLIP_BF16_PERSIST", "0")
    one = run()
    monkeypatch.setenv("DCLIP_BF16_PERSIST", "1")
    monkeypatch.setenv("DCLIP_BF16_PERSIST_MIN", "1")
    for rep in range(3):                                  # repeated: a race on the counted waits would show as a difference
        per = run()
        assert torch.equal(per[0], one[0]) and torch.equal(per[1], one[1]) and torch.equal(per[2], one[2])
        assert torch.equal(per[4], one[4]) and torch.equal(per[5][0], one[5][0]) and torch.equal(per[5][1], one[5][1])
        assert float((per[3] - one[3]).abs().max()) <= 2e-6 * float(one[3].abs().max())
    assert float((per[0].double().cpu() - want).abs().max() / want.abs().max()) < 2e-6 * max(1.0, K ** 0.5)
    assert float((per[3].double().cpu() - (want + res.double().cpu())).abs().max() / want.abs().max()) < 1e-5


@pytest.mark.parametrize("B,S,H,use_rows", [(3, 50, 12, False), (2, 257, 16, False), (5, 77, 8, True), (2, 1, 1, False),
                                            (2, 300, 2, True), (1, 512, 1, False), (7, 197, 3, False), (4, 77, 12, True)])
def test_attention_row_fwd_bf16(B, S, H, use_rows):
    """The last layer's one-row attention of a frozen bf16 tower (CLS row against all keys; first-EOS row against keys
    0..row) against fp64 on the same bf16-rounded q | k | v, and against the fp32 one-row kernels it replaces there."""
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(S * 31 + H)
    qkv = (torch.randn(B * S, 3 * H * 64, generator=g) * 1.5).to(torch.bfloat16)
    rows = torch.randint(0, S, (B,), generator=g, dtype=torch.int32) if use_rows else None
    if use_rows:
        rows[0] = S - 1
        rows[-1] = 0
    x = qkv.double().view(B, S, 3, H, 64)
    want = torch.empty(B, H * 64, dtype=torch.float64)
    for b in range(B):
        r = int(rows[b]) if use_rows else 0
        n = r + 1 if use_rows else S
        q, k, v = x[b, r, 0], x[b, :n, 1], x[b, :n, 2]                # [H,64], [n,H,64]
        p = torch.softmax(torch.einsum("hd,nhd->hn", q, k) * 0.125, dim=-1)
        want[b] = torch.einsum("hn,nhd->hd", p, v).reshape(-1)
    got = ops.attention_row_fwd_bf16(qkv.to(dev), None if rows is None else rows.to(dev), B, S, H)
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == (B, H * 64)
    err = float((got.double().cpu() - want).abs().max() / want.abs().max())
    assert err < 6e-3, err                                            # the bf16 rounding of the output
    q32 = qkv.float().to(dev)
    old = ops.attention_row_fwd(q32, rows.to(dev), B, S, H) if use_rows else ops.attention_cls_fwd(q32, B, S, H)[0]
    assert float((got.float() - old).abs().max() / old.abs().max()) < 6e-3
    with pytest.raises(ValueError):
        ops.attention_row_fwd_bf16(qkv.to(dev)[:, :-64], None, B, S, H)
