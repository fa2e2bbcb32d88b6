"""Tower shapes for the DCLIP distillation step.

The reference never states shapes itself; it takes whatever HF `CLIPModel` it is
handed (training/CLIP_image_distill_training.py:22-23).  These dataclasses carry
the same fields HF's `CLIPVisionConfig` / `CLIPTextConfig` carry for that model
(hf:configuration_clip.py) so a state dict with HF key names maps 1:1.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict


@dataclass(frozen=True)
class VisionConfig:
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    image_size: int = 224
    patch_size: int = 32
    num_channels: int = 3
    layer_norm_eps: float = 1e-5

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid

    @property
    def seq_len(self) -> int:
        return self.num_patches + 1

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def patch_dim(self) -> int:
        return self.num_channels * self.patch_size * self.patch_size


@dataclass(frozen=True)
class TextConfig:
    hidden_size: int = 512
    intermediate_size: int = 2048
    num_hidden_layers: int = 12
    num_attention_heads: int = 8
    max_position_embeddings: int = 77
    vocab_size: int = 49408
    bos_token_id: int = 49406
    eos_token_id: int = 49407
    layer_norm_eps: float = 1e-5

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


@dataclass(frozen=True)
class ClipConfig:
    vision: VisionConfig = field(default_factory=VisionConfig)
    text: TextConfig = field(default_factory=TextConfig)
    projection_dim: int = 512
    logit_scale_init_value: float = 2.6592
    name: str = "ViT-B/32"

    def to_dict(self) -> dict:
        return asdict(self)


def vit_b32() -> ClipConfig:
    return ClipConfig(name="ViT-B/32")


def vit_b16() -> ClipConfig:
    return ClipConfig(vision=VisionConfig(patch_size=16), name="ViT-B/16")


def vit_l14() -> ClipConfig:
    return ClipConfig(
        vision=VisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                            num_attention_heads=16, patch_size=14),
        text=TextConfig(hidden_size=768, intermediate_size=3072, num_attention_heads=12),
        projection_dim=768, name="ViT-L/14")


def tiny(image_size: int = 64, patch_size: int = 16, layers: int = 2, seq: int = 16,
         vocab: int = 512, proj: int = 64) -> ClipConfig:
    """A head_dim=64 toy used by the parity tests (weights small enough to commit)."""
    return ClipConfig(
        vision=VisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=layers,
                            num_attention_heads=2, image_size=image_size, patch_size=patch_size),
        text=TextConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=layers,
                        num_attention_heads=2, max_position_embeddings=seq, vocab_size=vocab,
                        bos_token_id=vocab - 2, eos_token_id=vocab - 1),
        projection_dim=proj, name="tiny")


NAMED = {"ViT-B/32": vit_b32, "ViT-B/16": vit_b16, "ViT-L/14": vit_l14, "tiny": tiny}


def from_hf(hf_config) -> ClipConfig:
    """Build from an HF `CLIPConfig` object (duck-typed; transformers is optional)."""
    v, t = hf_config.vision_config, hf_config.text_config
    return ClipConfig(
        vision=VisionConfig(v.hidden_size, v.intermediate_size, v.num_hidden_layers,
                            v.num_attention_heads, v.image_size, v.patch_size, v.num_channels,
                            v.layer_norm_eps),
        text=TextConfig(t.hidden_size, t.intermediate_size, t.num_hidden_layers,
                        t.num_attention_heads, t.max_position_embeddings, t.vocab_size,
                        t.bos_token_id, t.eos_token_id, t.layer_norm_eps),
        projection_dim=hf_config.projection_dim,
        logit_scale_init_value=hf_config.logit_scale_init_value, name="hf")
