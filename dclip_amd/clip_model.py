"""`HipCLIPModel`: the student / teacher CLIP towers behind the call surface the reference uses on HF
`CLIPModel`:

    model.get_image_features(pixel_values=Tensor[B,3,H,W]) -> Tensor[B,P]      (CLIP_image_distillation.py:601)
    model.get_text_features(input_ids=LongTensor[B,T], attention_mask=ignored) -> Tensor[B,P]   (:616)
    model.vision_model.named_parameters() / model.text_model.parameters()       (:504, :754)
    model.state_dict() / load_state_dict()  with HF key names (q_proj/k_proj/v_proj kept SEPARATE on disk)

Both calls return plain tensors (transformers 4.x semantics, which the reference's `.float()` calls assume).
In memory q/k/v are one fused [3D, D] parameter named `...self_attn.qkv_proj.{weight,bias}` — it contains
"proj", so the reference's freeze rule `if "proj" not in name: requires_grad = False` (:504-506) selects
exactly the same tensors as it does on the HF module.  state-dict hooks split / merge the fused tensor.

All arithmetic runs in the HIP library; there is no PyTorch fallback (importing works on CPU for
checkpoint handling, calling a tower without the GPU library raises).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import engine, functional
from .config import ClipConfig, TextConfig, VisionConfig


class _Affine(nn.Module):
    """Holder with HF-compatible `.weight` / `.bias` names (LayerNorm or Linear parameters)."""

    def __init__(self, w_shape, bias: bool = True, ones: bool = False):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(w_shape) if ones else torch.zeros(w_shape))
        if bias:
            self.bias = nn.Parameter(torch.zeros(w_shape[0]))
        else:
            self.register_parameter("bias", None)


class _Table(nn.Module):
    def __init__(self, rows, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(rows, dim))


class HipSelfAttention(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim
        self.qkv_proj = _Affine((3 * dim, dim))
        self.out_proj = _Affine((dim, dim))
        self._register_state_dict_hook(self._split_qkv)
        self._register_load_state_dict_pre_hook(self._merge_qkv)

    @staticmethod
    def _split_qkv(module, state_dict, prefix, local_metadata):
        D = module.dim
        for kind in ("weight", "bias"):
            fused = state_dict.pop(f"{prefix}qkv_proj.{kind}")
            for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
                state_dict[f"{prefix}{n}.{kind}"] = fused[i * D:(i + 1) * D]
        return state_dict

    def _merge_qkv(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for kind in ("weight", "bias"):
            keys = [f"{prefix}{n}.{kind}" for n in ("q_proj", "k_proj", "v_proj")]
            if all(k in state_dict for k in keys):
                state_dict[f"{prefix}qkv_proj.{kind}"] = torch.cat([state_dict.pop(k) for k in keys], dim=0)


class HipMLP(nn.Module):
    def __init__(self, dim, inter):
        super().__init__()
        self.fc1 = _Affine((inter, dim))
        self.fc2 = _Affine((dim, inter))


class HipEncoderLayer(nn.Module):
    def __init__(self, dim, inter):
        super().__init__()
        self.self_attn = HipSelfAttention(dim)
        self.layer_norm1 = _Affine((dim,), ones=True)
        self.mlp = HipMLP(dim, inter)
        self.layer_norm2 = _Affine((dim,), ones=True)

    def params(self) -> engine.LayerParams:
        a, m = self.self_attn, self.mlp
        return engine.LayerParams(self.layer_norm1.weight, self.layer_norm1.bias, a.qkv_proj.weight, a.qkv_proj.bias,
                                  a.out_proj.weight, a.out_proj.bias, self.layer_norm2.weight, self.layer_norm2.bias,
                                  m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)


class HipEncoder(nn.Module):
    def __init__(self, n, dim, inter):
        super().__init__()
        self.layers = nn.ModuleList([HipEncoderLayer(dim, inter) for _ in range(n)])


class _VisionEmbeddings(nn.Module):
    def __init__(self, v: VisionConfig):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.zeros(v.hidden_size))
        self.patch_embedding = nn.Module()
        self.patch_embedding.weight = nn.Parameter(torch.zeros(v.hidden_size, v.num_channels, v.patch_size, v.patch_size))
        self.position_embedding = _Table(v.seq_len, v.hidden_size)


class _TextEmbeddings(nn.Module):
    def __init__(self, t: TextConfig):
        super().__init__()
        self.token_embedding = _Table(t.vocab_size, t.hidden_size)
        self.position_embedding = _Table(t.max_position_embeddings, t.hidden_size)


class HipVisionTransformer(nn.Module):
    def __init__(self, v: VisionConfig):
        super().__init__()
        if v.head_dim != 64:
            raise ValueError("the HIP attention kernel is built for head_dim 64 (every CLIP ViT-B/L config)")
        self.config = v
        self.embeddings = _VisionEmbeddings(v)
        self.pre_layrnorm = _Affine((v.hidden_size,), ones=True)          # sic: HF key spelling
        self.encoder = HipEncoder(v.num_hidden_layers, v.hidden_size, v.intermediate_size)
        self.post_layernorm = _Affine((v.hidden_size,), ones=True)


class HipTextTransformer(nn.Module):
    def __init__(self, t: TextConfig):
        super().__init__()
        if t.head_dim != 64:
            raise ValueError("the HIP attention kernel is built for head_dim 64")
        self.config = t
        self.embeddings = _TextEmbeddings(t)
        self.encoder = HipEncoder(t.num_hidden_layers, t.hidden_size, t.intermediate_size)
        self.final_layer_norm = _Affine((t.hidden_size,), ones=True)


class HipCLIPModel(nn.Module):
    def __init__(self, config: Optional[ClipConfig] = None):
        super().__init__()
        self.config = config or ClipConfig()
        c = self.config
        self.logit_scale = nn.Parameter(torch.tensor(c.logit_scale_init_value))
        self.text_model = HipTextTransformer(c.text)
        self.vision_model = HipVisionTransformer(c.vision)
        self.visual_projection = _Affine((c.projection_dim, c.vision.hidden_size), bias=False)
        self.text_projection = _Affine((c.projection_dim, c.text.hidden_size), bias=False)

    # convenient aliases used by tests / smoke
    @property
    def visual_projection_weight(self):
        return self.visual_projection.weight

    def vision_params(self) -> engine.VisionParams:
        vm = self.vision_model
        e = vm.embeddings
        return engine.VisionParams(e.class_embedding, e.patch_embedding.weight, e.position_embedding.weight,
                                   vm.pre_layrnorm.weight, vm.pre_layrnorm.bias,
                                   [l.params() for l in vm.encoder.layers],
                                   vm.post_layernorm.weight, vm.post_layernorm.bias, self.visual_projection.weight)

    def text_params(self) -> engine.TextParams:
        tm = self.text_model
        return engine.TextParams(tm.embeddings.token_embedding.weight, tm.embeddings.position_embedding.weight,
                                 [l.params() for l in tm.encoder.layers],
                                 tm.final_layer_norm.weight, tm.final_layer_norm.bias, self.text_projection.weight)

    # ------------------------------------------------------------------ reference call surface
    def get_image_features(self, pixel_values: torch.Tensor = None, precision: str = "fp32", **kwargs) -> torch.Tensor:
        """precision="bf16" (opt-in, frozen use only): GEMM inputs in bf16 on the bf16 MFMA path, everything else fp32."""
        if pixel_values is None:
            raise ValueError("You have to specify pixel_values")
        v = self.config.vision
        if pixel_values.shape[-1] != v.image_size or pixel_values.shape[-2] != v.image_size:
            raise ValueError(f"Input image size ({pixel_values.shape[-2]}*{pixel_values.shape[-1]}) doesn't match "
                             f"model ({v.image_size}*{v.image_size}).")          # hf:modeling_clip.py:204-207
        p = self.vision_params()
        if precision == "bf16":
            if torch.is_grad_enabled() and any(t.requires_grad for t in p.tensors()):
                # TRAINING in bf16 (configs c3 / c5): forward, dgrad and wgrad GEMMs on the bf16 MFMA kernels, fp32
                # master weights and fp32 everything else (engine.vision_fwd_bf16_train)
                return functional.VisionTowerBf16Fn.apply(pixel_values.float(), v, v.num_hidden_layers, self._bf16_cache(),
                                                          *p.tensors())
            pd = engine.VisionParams.from_tensors([t.detach() for t in p.tensors()], v.num_hidden_layers)
            return engine.vision_fwd_bf16(pd, pixel_values.float().contiguous(), v, self._bf16_cache())
        if precision != "fp32":
            raise ValueError(f"precision {precision!r}")
        return functional.VisionTowerFn.apply(pixel_values.float(), v, v.num_hidden_layers, *p.tensors())

    def _bf16_cache(self) -> dict:
        """bf16 copies of the GEMM weights: persistent buffers, each refreshed in place when ITS parameter's version
        counter has moved (optimizer step, load_state_dict) — engine._w16 / _w16t.  A frozen tower's copies are made once."""
        c = getattr(self, "_bf16_w", None)
        if c is None:
            c = {}
            object.__setattr__(self, "_bf16_w", c)
        return c

    def invalidate_bf16_of_trainable(self) -> int:
        """Mark the bf16 copies of every TRAINABLE parameter stale (frozen towers keep theirs).  graph.GraphedStep calls
        this between its eager warm-up and the capture: the weight casts / transposes are then part of the captured step
        and every replay converts the CURRENT fp32 masters — without it a replay would multiply by the copies made
        at warm-up while the optimizer keeps updating the masters.  Returns the number of entries marked."""
        c = getattr(self, "_bf16_w", None)
        if not c:
            return 0
        ptrs = {p.data_ptr() for p in self.parameters() if p.requires_grad}
        n = 0
        for e in c.values():
            if isinstance(e, list) and e[2] in ptrs:
                e[1] = -1
                n += 1
        return n

    def get_text_features(self, input_ids: torch.Tensor = None, attention_mask=None, precision: str = "fp32",
                          **kwargs) -> torch.Tensor:
        """`attention_mask` is accepted and ignored: under the causal mask trailing pads cannot influence the
        first-EOS row that is pooled.  precision="bf16": see get_image_features."""
        if input_ids is None:
            raise ValueError("You have to specify input_ids")
        t = self.config.text
        if input_ids.shape[-1] > t.max_position_embeddings:
            raise ValueError(f"Sequence length must be less than max_position_embeddings (got `sequence length`: "
                             f"{input_ids.shape[-1]} and max_position_embeddings: {t.max_position_embeddings}")
        p = self.text_params()
        if precision == "bf16":
            if torch.is_grad_enabled() and any(x.requires_grad for x in p.tensors()):
                raise RuntimeError("precision='bf16' is a forward-only path for frozen towers: call it under torch.no_grad()")
            return engine.text_fwd_frozen_bf16(self.text_params_detached(), input_ids.long().contiguous(), t,
                                               self._bf16_cache())
        if precision != "fp32":
            raise ValueError(f"precision {precision!r}")
        return functional.TextTowerFn.apply(input_ids.long(), t, t.num_hidden_layers, *p.tensors())

    @torch.no_grad()
    def text_token_level(self, input_ids: torch.Tensor, precision: str = "fp32"):
        """Frozen pass used by the meta-teacher: (sentence [B,P], tokens [B,T,P], first-EOS index [B])."""
        t = self.config.text
        ids = input_ids.long().contiguous()
        if precision == "bf16":
            sent, tokens, eos = engine.text_token_level_bf16(self.text_params_detached(), ids, t, self._bf16_cache())
        elif precision == "fp32":
            sent, tokens, eos = engine.text_token_level(self.text_params_detached(), ids, t)
        else:
            raise ValueError(f"precision {precision!r}")
        return sent, tokens.view(input_ids.shape[0], input_ids.shape[1], -1), eos

    def text_params_detached(self) -> engine.TextParams:
        p = self.text_params()
        return engine.TextParams.from_tensors([t.detach() for t in p.tensors()], len(p.layers))

    @torch.no_grad()
    def hidden_states(self, pixel_values=None, input_ids=None) -> List[torch.Tensor]:
        """Per-layer hidden states (parity tests): [embeddings-after-pre-LN, layer 1, ...] for vision,
        [embeddings, layer 1, ...] (before final_layer_norm) for text."""
        out: List[torch.Tensor] = []
        if pixel_values is not None:
            v = self.config.vision
            p = engine.VisionParams.from_tensors([t.detach() for t in self.vision_params().tensors()],
                                                 v.num_hidden_layers)
            engine.vision_fwd(p, pixel_values.float().contiguous(), v, False, out)
            return [h.view(pixel_values.shape[0], v.seq_len, -1) for h in out]
        t = self.config.text
        engine.text_encoder_fwd(self.text_params_detached(), input_ids.long().contiguous(), t, False, out)
        return [h.view(input_ids.shape[0], input_ids.shape[1], -1) for h in out]


def from_hf_state_dict(config: ClipConfig, state_dict, device=None) -> HipCLIPModel:
    """Build a model from an HF-keyed state dict (e.g. `CLIPModel.from_pretrained(local_dir).state_dict()`)."""
    m = HipCLIPModel(config)
    missing, unexpected = m.load_state_dict(state_dict, strict=False)
    unexpected = [k for k in unexpected if "position_ids" not in k]       # 4.x checkpoints carry these buffers
    if missing or unexpected:
        raise KeyError(f"state dict mismatch: missing={missing[:5]} unexpected={unexpected[:5]}")
    return m.to(device) if device is not None else m
