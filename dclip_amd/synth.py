"""Deterministic synthetic weights and inputs.

No pretrained weights or BPE vocabulary exist offline, so every test, fixture and
bench run uses seeded random tensors.  Weights follow the std scheme of HF's
`CLIPPreTrainedModel._init_weights` (hf:modeling_clip.py:404-452) but are drawn
here with an explicit `torch.Generator` in a fixed key order, so the SAME state
dict can be loaded into the HF oracle model (in the build container) and into the
HIP model (on the GPU box) without shipping hundreds of MB of fixtures.
"""
from __future__ import annotations

import torch

from .config import ClipConfig, VisionConfig, TextConfig


def _normal(gen, shape, std):
    return torch.randn(shape, generator=gen, dtype=torch.float32, device=gen.device) * std


def _tower_layers(sd, gen, prefix, hidden, inter, n_layers, jitter):
    in_proj_std = hidden ** -0.5 * (2 * n_layers) ** -0.5
    out_proj_std = hidden ** -0.5
    fc_std = (2 * hidden) ** -0.5
    for i in range(n_layers):
        p = f"{prefix}.encoder.layers.{i}"
        for nm, std in (("q_proj", in_proj_std), ("k_proj", in_proj_std), ("v_proj", in_proj_std),
                        ("out_proj", out_proj_std)):
            sd[f"{p}.self_attn.{nm}.weight"] = _normal(gen, (hidden, hidden), std)
            sd[f"{p}.self_attn.{nm}.bias"] = _normal(gen, (hidden,), jitter)
        for ln in ("layer_norm1", "layer_norm2"):
            sd[f"{p}.{ln}.weight"] = 1.0 + _normal(gen, (hidden,), jitter)
            sd[f"{p}.{ln}.bias"] = _normal(gen, (hidden,), jitter)
        sd[f"{p}.mlp.fc1.weight"] = _normal(gen, (inter, hidden), fc_std)
        sd[f"{p}.mlp.fc1.bias"] = _normal(gen, (inter,), jitter)
        sd[f"{p}.mlp.fc2.weight"] = _normal(gen, (hidden, inter), in_proj_std)
        sd[f"{p}.mlp.fc2.bias"] = _normal(gen, (hidden,), jitter)


def synth_clip_state_dict(cfg: ClipConfig, seed: int = 0, jitter: float = 0.02, gain: float = 1.0, device=None):
    """HF-keyed fp32 state dict for `cfg` (keys: SURVEY.md §8b-keys).

    `device`: draw on that device with ITS generator (seconds faster for ViT-L/14's 428 M parameters) — same shapes and
    stds, but NOT the values of the CPU draw that fixtures and tests are keyed to: for benchmarks of frozen towers only.

    `jitter` perturbs biases / LayerNorm affine params away from HF's 0/1 init so
    parity tests exercise them.  `gain` scales the q/k/v/fc weights: the HF init is
    so small that attention is near-uniform; tests use gain>1 to get peaked softmax.
    """
    gen = (torch.Generator() if device is None else torch.Generator(device=device)).manual_seed(seed)
    v, t = cfg.vision, cfg.text
    sd = {}
    sd["logit_scale"] = torch.tensor(cfg.logit_scale_init_value, dtype=torch.float32)
    sd["text_model.embeddings.token_embedding.weight"] = _normal(gen, (t.vocab_size, t.hidden_size), 0.02)
    sd["text_model.embeddings.position_embedding.weight"] = _normal(
        gen, (t.max_position_embeddings, t.hidden_size), 0.02)
    _tower_layers(sd, gen, "text_model", t.hidden_size, t.intermediate_size, t.num_hidden_layers, jitter)
    sd["text_model.final_layer_norm.weight"] = 1.0 + _normal(gen, (t.hidden_size,), jitter)
    sd["text_model.final_layer_norm.bias"] = _normal(gen, (t.hidden_size,), jitter)
    sd["vision_model.embeddings.class_embedding"] = _normal(gen, (v.hidden_size,), v.hidden_size ** -0.5)
    sd["vision_model.embeddings.patch_embedding.weight"] = _normal(
        gen, (v.hidden_size, v.num_channels, v.patch_size, v.patch_size), 0.02)
    sd["vision_model.embeddings.position_embedding.weight"] = _normal(gen, (v.seq_len, v.hidden_size), 0.02)
    sd["vision_model.pre_layrnorm.weight"] = 1.0 + _normal(gen, (v.hidden_size,), jitter)
    sd["vision_model.pre_layrnorm.bias"] = _normal(gen, (v.hidden_size,), jitter)
    _tower_layers(sd, gen, "vision_model", v.hidden_size, v.intermediate_size, v.num_hidden_layers, jitter)
    sd["vision_model.post_layernorm.weight"] = 1.0 + _normal(gen, (v.hidden_size,), jitter)
    sd["vision_model.post_layernorm.bias"] = _normal(gen, (v.hidden_size,), jitter)
    sd["visual_projection.weight"] = _normal(gen, (cfg.projection_dim, v.hidden_size), v.hidden_size ** -0.5)
    sd["text_projection.weight"] = _normal(gen, (cfg.projection_dim, t.hidden_size), t.hidden_size ** -0.5)
    if gain != 1.0:
        for k in sd:
            if any(s in k for s in ("q_proj.weight", "k_proj.weight", "v_proj.weight", "fc1.weight",
                                    "fc2.weight", "out_proj.weight")):
                sd[k] = sd[k] * gain
    return sd


def synth_cross_modal_state_dict(embed_dim: int = 512, seed: int = 0, prefix: str = ""):
    """The 12 tensors of `cross_modal_attention.*` (SURVEY.md §8b teacher checkpoint)."""
    gen = torch.Generator().manual_seed(seed)
    e = embed_dim
    sd = {}
    for d in ("text_to_image", "image_to_text"):
        sd[f"{prefix}{d}.in_proj_weight"] = _normal(gen, (3 * e, e), (2.0 / (4 * e)) ** 0.5)
        sd[f"{prefix}{d}.in_proj_bias"] = _normal(gen, (3 * e,), 0.02)
        sd[f"{prefix}{d}.out_proj.weight"] = _normal(gen, (e, e), e ** -0.5)
        sd[f"{prefix}{d}.out_proj.bias"] = _normal(gen, (e,), 0.02)
    for n in ("norm_text", "norm_image"):
        sd[f"{prefix}{n}.weight"] = 1.0 + _normal(gen, (e,), 0.02)
        sd[f"{prefix}{n}.bias"] = _normal(gen, (e,), 0.02)
    return sd


def synth_pixel_values(batch: int, cfg: VisionConfig, seed: int = 0) -> torch.Tensor:
    gen = torch.Generator().manual_seed(seed)
    return torch.randn((batch, cfg.num_channels, cfg.image_size, cfg.image_size), generator=gen)


def synth_regions(batch: int, regions: int, cfg: VisionConfig, seed: int = 2) -> torch.Tensor:
    """Region crops in [0,1]: the reference feeds ToTensor() output without mean/std
    (training/image_tokenizer.py:28-32)."""
    gen = torch.Generator().manual_seed(seed)
    return torch.rand((batch, regions, cfg.num_channels, cfg.image_size, cfg.image_size), generator=gen)


def synth_input_ids(batch: int, cfg: TextConfig, seed: int = 3, ragged: bool = False,
                    min_len: int = 3) -> torch.Tensor:
    """`[B, T]` int64: BOS, L-2 random word ids, EOS, then EOS padding (the CLIP tokenizer
    pads with its EOS id).  L = T unless `ragged`."""
    gen = torch.Generator().manual_seed(seed)
    T = cfg.max_position_embeddings
    hi = min(cfg.bos_token_id, cfg.eos_token_id)
    ids = torch.randint(1, hi, (batch, T), generator=gen, dtype=torch.int64)
    if ragged:
        lens = torch.randint(min_len, T + 1, (batch,), generator=gen)
    else:
        lens = torch.full((batch,), T, dtype=torch.int64)
    ids[:, 0] = cfg.bos_token_id
    for b in range(batch):
        ids[b, int(lens[b]) - 1:] = cfg.eos_token_id
    return ids


def synth_embeddings(batch: int, dim: int, seed: int = 1) -> torch.Tensor:
    gen = torch.Generator().manual_seed(seed)
    return torch.randn((batch, dim), generator=gen)


def synth_photo(height: int, width: int, seed: int = 0):
    """HWC uint8 RGB test image: smooth colour gradients + a few hard edges + noise, so that resampling filters
    (antialiasing, negative bicubic lobes, clipping at 0 / 255) all matter.  numpy RandomState: stable across hosts."""
    import numpy as np
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    img = np.zeros((height, width, 3), dtype=np.float64)
    for c in range(3):
        fx, fy, ph = rs.uniform(0.5, 6.0), rs.uniform(0.5, 6.0), rs.uniform(0, 6.28)
        img[..., c] = 127.5 + 110.0 * np.sin(fx * xx / width * 6.28 + fy * yy / height * 6.28 + ph)
    for _ in range(6):                                   # saturated rectangles: hard edges at 0 and 255
        y0, x0 = rs.randint(0, height), rs.randint(0, width)
        y1, x1 = min(height, y0 + rs.randint(1, max(2, height // 3))), min(width, x0 + rs.randint(1, max(2, width // 3)))
        img[y0:y1, x0:x1] = rs.choice([0.0, 255.0], size=3)
    img += rs.normal(0.0, 12.0, size=img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)
