"""The meta-teacher: `CrossModalAttention` and `PatchTextAggregation`, same names, constructor arguments,
methods and checkpoint keys as training/patch_text_aggregation.py, on the HIP kernels.

What is kept from the reference:
  * `CrossModalAttention(embed_dim, num_heads)`: two packed-projection multi-head attentions and two LayerNorms;
    `state_dict()` holds exactly the 12 tensors `{text_to_image,image_to_text}.{in_proj_weight,in_proj_bias,
    out_proj.weight,out_proj.bias}`, `{norm_text,norm_image}.{weight,bias}`.
  * both directions attend from the ORIGINAL inputs; no key_padding_mask — zero-padded rows are attended,
    LayerNorm'd and pooled (:33,:42,:555-620).
  * `aggregation`: cosine-to-mean softmax pooling with temperature 2.0 (:243-265); final 0.5/0.5 mix (:647).
  * region crops go to the CLIP vision tower in [0,1] without mean/std (training/image_tokenizer.py:28-32; N5).
  * tokenizers are plain objects, not nn.Modules, so the teacher checkpoint contains only `cross_modal_attention.*`.

What is deliberately different: the per-sample Python loop (:297-553) is replaced by ONE batched frozen vision
forward over all regions and ONE batched frozen text forward that yields token-level and sentence embeddings
together; the KNN / projection tokenizer (disabled when its paths are empty, :78-96) and YOLO detection are out of
scope — boxes / region crops are inputs.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import ops
from .clip_model import HipCLIPModel, _Affine


# ------------------------------------------------------------------------------------------------ cross-modal block

class _PackedMHA(nn.Module):
    """Parameter holder with nn.MultiheadAttention's names and default initialisation."""

    def __init__(self, embed_dim: int):
        super().__init__()
        e = embed_dim
        self.in_proj_weight = nn.Parameter(torch.empty(3 * e, e))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * e))
        self.out_proj = _Affine((e, e))
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.kaiming_uniform_(self.out_proj.weight, a=math.sqrt(5))


def _mha_dir_fwd(xq, xkv, B, Lq, Lk, E, H, in_w, in_b, out_w, out_b, ln_w, ln_b):
    q = ops.gemm(xq, in_w[:E], ops.LAYOUT_NT, bias=in_b[:E])
    kv = ops.gemm(xkv, in_w[E:], ops.LAYOUT_NT, bias=in_b[E:])
    a, lse = ops.cross_attention_fwd(q, kv, B, Lq, Lk, H)
    pre = ops.gemm(a, out_w, ops.LAYOUT_NT, bias=out_b, residual=xq)
    y, mean, rstd = ops.layernorm_fwd(pre, ln_w, ln_b, 1e-5)
    return y, (q, kv, a, lse, pre, mean, rstd)


def _mha_dir_bwd(dy, xq, xkv, saved, B, Lq, Lk, E, H, in_w, out_w, ln_w, need_x: bool):
    q, kv, a, lse, pre, mean, rstd = saved
    dpre, d_ln_w, d_ln_b = ops.layernorm_bwd(dy, pre, ln_w, mean, rstd)
    d_out_w = ops.gemm(dpre, a, ops.LAYOUT_TN)
    d_out_b = ops.colsum(dpre)
    da = ops.gemm(dpre, out_w, ops.LAYOUT_NN)
    dq, dkv = ops.cross_attention_bwd(q, kv, a, da, lse, B, Lq, Lk, H)
    d_in_w = torch.empty_like(in_w)
    d_in_b = torch.empty((3 * E,), dtype=torch.float32, device=in_w.device)
    ops.gemm(dq, xq, ops.LAYOUT_TN, out=d_in_w[:E])
    ops.gemm(dkv, xkv, ops.LAYOUT_TN, out=d_in_w[E:])
    ops.colsum(dq, out=d_in_b[:E])
    ops.colsum(dkv, out=d_in_b[E:])
    dxq = dxkv = None
    if need_x:
        dxq = ops.gemm(dq, in_w[:E], ops.LAYOUT_NN, residual=dpre)
        dxkv = ops.gemm(dkv, in_w[E:], ops.LAYOUT_NN)
    return (d_in_w, d_in_b, d_out_w, d_out_b, d_ln_w, d_ln_b), dxq, dxkv


class CrossModalFn(torch.autograd.Function):
    """(text [B,T,E], patches [B,R,E]) -> (LN(text + MHA(text <- patches)), LN(patches + MHA(patches <- text)))."""

    @staticmethod
    def forward(ctx, text, patches, heads, *params):
        (t_in_w, t_in_b, t_out_w, t_out_b, i_in_w, i_in_b, i_out_w, i_out_b, nt_w, nt_b, ni_w, ni_b) = \
            [p.detach().contiguous() for p in params]
        B, T, E = text.shape
        R = patches.shape[1]
        t2 = text.detach().float().contiguous().view(B * T, E)
        p2 = patches.detach().float().contiguous().view(B * R, E)
        t_out, sv_t = _mha_dir_fwd(t2, p2, B, T, R, E, heads, t_in_w, t_in_b, t_out_w, t_out_b, nt_w, nt_b)
        i_out, sv_i = _mha_dir_fwd(p2, t2, B, R, T, E, heads, i_in_w, i_in_b, i_out_w, i_out_b, ni_w, ni_b)
        if any(ctx.needs_input_grad):
            ctx.saved = (t2, p2, sv_t, sv_i, (t_in_w, t_out_w, i_in_w, i_out_w, nt_w, ni_w))
        ctx.dims = (B, T, R, E, heads)
        return t_out.view(B, T, E), i_out.view(B, R, E)

    @staticmethod
    def backward(ctx, d_t, d_i):
        B, T, R, E, H = ctx.dims
        t2, p2, sv_t, sv_i, (t_in_w, t_out_w, i_in_w, i_out_w, nt_w, ni_w) = ctx.saved
        need_x = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        d_t = d_t.contiguous().view(B * T, E)
        d_i = d_i.contiguous().view(B * R, E)
        gt, dxq_t, dxkv_t = _mha_dir_bwd(d_t, t2, p2, sv_t, B, T, R, E, H, t_in_w, t_out_w, nt_w, need_x)
        gi, dxq_i, dxkv_i = _mha_dir_bwd(d_i, p2, t2, sv_i, B, R, T, E, H, i_in_w, i_out_w, ni_w, need_x)
        d_text = d_patch = None
        if need_x:
            d_text = ops.axpby(dxkv_i, dxq_t, 1.0, 1.0).view(B, T, E)
            d_patch = ops.axpby(dxkv_t, dxq_i, 1.0, 1.0).view(B, R, E)
        grads = [d_text, d_patch, None, gt[0], gt[1], gt[2], gt[3], gi[0], gi[1], gi[2], gi[3], gt[4], gt[5], gi[4], gi[5]]
        flag = getattr(ctx, "replaced_flag", None)
        if flag is not None:
            # the batch was replaced by zeros downstream (NaN / Inf guard, :649-651): in the reference that result is
            # a fresh tensor with no graph, so nothing reaches this block; the saved activations hold the NaNs, so the
            # gradients are zeroed rather than trusted
            for g in grads:
                if g is not None:
                    ops.sanitize_groups(g.view(1, -1) if g.dim() == 1 else g, g.numel() // g.shape[-1], flag)
        # parameter order: t.{in_w,in_b,out_w,out_b}, i.{...}, norm_text.{w,b}, norm_image.{w,b}
        return tuple(grads)


class CrossModalAttention(nn.Module):
    def __init__(self, embed_dim, num_heads):
        super().__init__()
        if embed_dim % num_heads or embed_dim // num_heads != 64:
            raise ValueError("the HIP attention kernel is built for head_dim 64 (embed_dim / num_heads)")
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.text_to_image = _PackedMHA(embed_dim)
        self.image_to_text = _PackedMHA(embed_dim)
        self.norm_text = _Affine((embed_dim,), ones=True)
        self.norm_image = _Affine((embed_dim,), ones=True)

    def _params(self):
        t, i = self.text_to_image, self.image_to_text
        return (t.in_proj_weight, t.in_proj_bias, t.out_proj.weight, t.out_proj.bias,
                i.in_proj_weight, i.in_proj_bias, i.out_proj.weight, i.out_proj.bias,
                self.norm_text.weight, self.norm_text.bias, self.norm_image.weight, self.norm_image.bias)

    def forward(self, text_embedding, image_embedding):
        """Bidirectional attention between text and image tokens (training/patch_text_aggregation.py:21-46)."""
        return CrossModalFn.apply(text_embedding, image_embedding, self.num_heads, *self._params())


class GlobalPoolFn(torch.autograd.Function):
    """0.5 * aggregation(attended_text) + 0.5 * aggregation(attended_image) (:643-647), one kernel per side."""

    @staticmethod
    def forward(ctx, at, ai, temperature):
        at, ai = at.detach().contiguous(), ai.detach().contiguous()
        out, wt = ops.aggregation_fwd(at, temperature, out_scale=0.5)
        _, wi = ops.aggregation_fwd(ai, temperature, out=out, out_scale=0.5, accumulate=True)
        # :649-651 — any NaN / Inf anywhere in the [B,E] result replaces the WHOLE batch by zeros
        _, flag = ops.sanitize_groups(out, out.shape[0])
        ctx.replaced_flag = flag
        ctx.save_for_backward(at, ai, wt, wi, flag)
        ctx.temperature = temperature
        return out

    @staticmethod
    def backward(ctx, d_out):
        at, ai, wt, wi, flag = ctx.saved_tensors
        d_out = d_out.contiguous()
        d_at = ops.aggregation_bwd(at, wt, d_out, ctx.temperature, 0.5)
        d_ai = ops.aggregation_bwd(ai, wi, d_out, ctx.temperature, 0.5)
        # a replaced batch is a fresh zeros tensor in the reference: NO gradient reaches the block (the saved
        # activations hold the NaNs, so the gradients themselves are zeroed, not just d_out)
        ops.sanitize_groups(d_at, d_at.shape[0] * d_at.shape[1], flag)
        ops.sanitize_groups(d_ai, d_ai.shape[0] * d_ai.shape[1], flag)
        return d_at, d_ai, None


class AggregationFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, temperature):
        x = x.detach().contiguous()
        out, w = ops.aggregation_fwd(x, temperature)
        ctx.save_for_backward(x, w)
        ctx.temperature = temperature
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, w = ctx.saved_tensors
        return ops.aggregation_bwd(x, w, d_out.contiguous(), ctx.temperature, 1.0), None


# ------------------------------------------------------------------------------------------------ tokenizers

class CLIPTextTokenizer:
    """Frozen teacher text side (training/text_tokenizer.py).  `model` is a HipCLIPModel; `tokenizer` an HF
    CLIPTokenizer loaded from a LOCAL path, or None when callers pass token ids (synthetic configs).
    The BERT / GloVe word-complexity machinery of the reference (:26-143) is never called on this path."""

    def __init__(self, model: HipCLIPModel, tokenizer=None, max_chunk_size: int = 77, precision: str = "fp32"):
        self.model = model
        self.tokenizer = tokenizer
        self.max_chunk_size = max_chunk_size
        self.precision = precision          # "bf16": opt-in bf16-input GEMMs for this frozen tower

    @property
    def device(self):
        return next(self.model.parameters()).device

    def _ids(self, text, keep_host: bool = False) -> torch.Tensor:
        if isinstance(text, torch.Tensor):
            ids = text if text.dim() == 2 else text.unsqueeze(0)
        else:
            if self.tokenizer is None:
                raise RuntimeError("caption strings need a CLIPTokenizer (load one from a local path); "
                                   "or pass token ids")
            ids = self.tokenizer(text, return_tensors="pt", padding=True, truncation=True, max_length=77).input_ids
        ids = ids.long().contiguous()
        return ids if keep_host else ids.to(self.device)

    @torch.no_grad()
    def token_level_ids(self, input_ids: torch.Tensor):
        """(sentence [B,P], tokens [B,T,P], eos [B]) from one frozen forward."""
        return self.model.text_token_level(self._ids(input_ids), precision=self.precision)

    @torch.no_grad()
    def aggregate_text_ids(self, input_ids: torch.Tensor) -> torch.Tensor:
        return self.token_level_ids(input_ids)[0]

    @torch.no_grad()
    def get_embeddings(self, text, return_token_level=True) -> List[torch.Tensor]:
        """:171-216 — list of word-token embeddings (BOS/EOS dropped), or [sentence embedding]."""
        sent, tokens, eos = self.token_level_ids(self._ids(text)[:1])
        n = max(int(eos[0]) - 1, 0)
        if return_token_level and n > 0:
            return [tokens[0, 1 + i] for i in range(n)]
        return [sent[0]]

    @torch.no_grad()
    def aggregate_text(self, text) -> torch.Tensor:
        """:220-235 — the sentence-level embedding [E]."""
        return self.aggregate_text_ids(self._ids(text)[:1])[0]


class CLIPPatchTokenizer:
    """Frozen teacher region side (training/image_tokenizer.py:19-124) without the detector: boxes are inputs."""

    def __init__(self, clip_model: HipCLIPModel, precision: str = "fp32"):
        self.clip_model = clip_model
        self.precision = precision

    @property
    def device(self):
        return next(self.clip_model.parameters()).device

    def get_weighted_bounding_boxes(self, image_path):
        raise NotImplementedError("YOLO detection is outside the distillation step: pass cached boxes "
                                  "(weighted_boxes_batch) as the reference's training loop does")

    get_weighted_bounding_boxes_batch = get_weighted_bounding_boxes

    def patch_transform(self, pil_patch) -> torch.Tensor:
        """Resize((S,S)) + ToTensor(): bilinear PIL resize, HWC uint8 -> CHW float in [0,1], no mean/std (:28-32)."""
        import numpy as np
        from PIL import Image
        s = self.clip_model.config.vision.image_size
        arr = np.asarray(pil_patch.convert("RGB").resize((s, s), Image.BILINEAR), dtype=np.float32) / 255.0
        return torch.from_numpy(arr).permute(2, 0, 1).contiguous()

    def crop_boxes_gpu(self, images: Sequence, boxes_per_image: Sequence[Sequence], images_u8: Optional[torch.Tensor] = None,
                       dims: Optional[torch.Tensor] = None) -> tuple:
        """PIL images (or HWC uint8 arrays) + per-image box lists -> (regions [B,Rmax,3,S,S] in [0,1], counts [B]).
        The crops are cut, resized and converted on the GPU, bit-exact with `patch_transform` (Pillow).
        `images_u8` [B,Hmax,Wmax,3] + `dims` [B,2] (already on the device, data.GpuCollate) replace `images`."""
        import numpy as np
        dev = self.device
        s = self.clip_model.config.vision.image_size
        if images_u8 is None:
            arrs = [np.asarray(im.convert("RGB") if hasattr(im, "convert") else im, dtype=np.uint8) for im in images]
            hmax, wmax = max(a.shape[0] for a in arrs), max(a.shape[1] for a in arrs)
            batch = np.zeros((len(arrs), hmax, wmax, 3), dtype=np.uint8)
            for b, a in enumerate(arrs):
                batch[b, :a.shape[0], :a.shape[1]] = a
            images_u8 = torch.from_numpy(batch).to(dev)
            dims = torch.tensor([a.shape[:2] for a in arrs], dtype=torch.int32).to(dev)
        B = images_u8.shape[0]
        if len(boxes_per_image) != B:
            raise ValueError("crop_boxes_gpu: one box list per image")
        flat, counts = [], []
        for b, boxes in enumerate(boxes_per_image):
            # A zero / negative-extent box (YOLO coordinates are int-truncated, training/image_tokenizer.py:56) makes
            # PIL's crop + Resize raise inside encode_weighted_bounding_boxes; the reference catches that around the
            # WHOLE image (training/patch_text_aggregation.py:479-486): `patch_embed_list = []`, i.e. the image keeps
            # the single zero patch row of an image without boxes (:489-491).  Same here: count 0, no crop is cut.
            if any(x2 <= x1 or y2 <= y1 for (x1, y1, x2, y2), _conf in boxes):
                counts.append(0)
                continue
            counts.append(len(boxes))
            for (x1, y1, x2, y2), _conf in boxes:
                flat.append((b, int(x1), int(y1), int(x2), int(y2)))
        rmax = max(max(counts), 1)
        regions = torch.zeros((B, rmax, 3, s, s), dtype=torch.float32, device=dev)
        if flat:
            bx = torch.tensor(flat, dtype=torch.int32)
            crops = ops.crop_resize(images_u8, dims, bx.to(dev), s, int((bx[:, 4] - bx[:, 2]).max()),
                                    int((bx[:, 3] - bx[:, 1]).max()))
            # one scatter for the whole batch (flat slot b*rmax + r of crop number i), not one copy per image
            slots = [b * rmax + r for b, n in enumerate(counts) for r in range(n)]
            regions.view(B * rmax, 3, s, s).index_copy_(0, torch.tensor(slots, dtype=torch.int64, device=dev), crops)
        return regions, torch.tensor(counts, dtype=torch.int32)

    @torch.no_grad()
    def encode_regions(self, regions: torch.Tensor) -> torch.Tensor:
        """[N,3,S,S] in [0,1] -> [N,E] (one batched frozen forward)."""
        return self.clip_model.get_image_features(pixel_values=regions.to(self.device).float(), precision=self.precision)

    @torch.no_grad()
    def encode_weighted_bounding_boxes(self, image, weighted_boxes, full_resolution=False):
        """:86-124 — list of (clip_embedding, confidence)."""
        if full_resolution:
            raise NotImplementedError("full_resolution crops are rejected by CLIP without interpolate_pos_encoding "
                                      "(hf:modeling_clip.py:204-207); the reference never enables it (SURVEY §5)")
        if not weighted_boxes:
            return []
        crops = torch.stack([self.patch_transform(image.crop(box)) for box, _ in weighted_boxes])
        embs = self.encode_regions(crops)
        return [(e, conf) for e, (_, conf) in zip(embs, weighted_boxes)]


# ------------------------------------------------------------------------------------------------ the teacher

class PatchTextAggregation(nn.Module):
    def __init__(self, embed_dim=512, num_heads=8, similarity_threshold=0.85, projection_model_path=None,
                 faiss_index_path=None, embeddings_json_path=None, clip_model: Optional[HipCLIPModel] = None,
                 tokenizer=None, tower_precision: str = "fp32", owns_clip: bool = False,
                 text_twin: Optional[HipCLIPModel] = None):
        """`tower_precision` ("fp32" default = the reference's arithmetic; "bf16" opt-in) selects how the FROZEN
        region / text towers multiply; the trainable cross_modal_attention always runs in fp32.
        `owns_clip`: the towers are this teacher's private frozen copy (they follow `.to()` / `.cuda()` although they
        stay out of `state_dict()`).  `text_twin`: a model whose text tower held the SAME weights as `clip_model`'s
        when this teacher was built (the student a snapshot was taken from) — see shares_text_tower_with."""
        super().__init__()
        if tower_precision not in ("fp32", "bf16"):
            raise ValueError(f"tower_precision {tower_precision!r}")
        if all([projection_model_path, faiss_index_path, embeddings_json_path]):
            raise NotImplementedError("the KNN + projection tokenizer is outside the distillation step "
                                      "(README.md:21: leave these paths blank)")
        if clip_model is None:
            raise ValueError("pass clip_model (a HipCLIPModel holding the teacher's CLIP towers); nothing is "
                             "downloaded by name here")
        self.embed_dim = embed_dim
        self.similarity_threshold = similarity_threshold
        # plain attributes on purpose: the towers must not enter teacher.state_dict()
        object.__setattr__(self, "_clip", clip_model)
        object.__setattr__(self, "_owns_clip", bool(owns_clip))
        object.__setattr__(self, "_text_twin", text_twin)
        object.__setattr__(self, "_text_twin_versions",
                           None if text_twin is None else self._text_versions(text_twin))
        self.text_tokenizer = CLIPTextTokenizer(clip_model, tokenizer, precision=tower_precision)
        self.patch_tokenizer = CLIPPatchTokenizer(clip_model, precision=tower_precision)
        self.cross_modal_attention = CrossModalAttention(embed_dim, num_heads)
        self.knn_cache = {}
        self.use_knn_projection = False
        self.last_sentence_embedding = None
        self.advanced_tokenizer = None
        self.full_resolution = False

    @property
    def device(self):
        return self.cross_modal_attention.norm_text.weight.device

    @staticmethod
    def _text_versions(model: HipCLIPModel):
        return tuple(p._version for p in list(model.text_model.parameters()) + [model.text_projection.weight])

    def _apply(self, fn, *args, **kwargs):
        # a private tower copy is a plain attribute (it must stay out of state_dict()), so nn.Module would not move it
        if self._owns_clip:
            self._clip._apply(fn, *args, **kwargs)
        return super()._apply(fn, *args, **kwargs)

    def shares_text_tower_with(self, student: HipCLIPModel) -> bool:
        """True when one frozen text forward serves teacher and student: the student's text tower is frozen and the
        teacher's text tower either IS it, or is a snapshot of it and the student's text weights have not been
        written since (parameter version counters recorded at snapshot time)."""
        frozen = not any(p.requires_grad for p in student.text_model.parameters()) \
            and not student.text_projection.weight.requires_grad
        if not frozen:
            return False
        if self._clip is student:
            return True
        return self._text_twin is student and self._text_twin_versions == self._text_versions(student)

    def load_caches(self, knn_cache_path=None):
        """:104-124 — the KNN cache only feeds the (out-of-scope) KNN tokenizer; kept so callers do not break."""
        self.knn_cache = {}
        return self

    def cross_attention(self, text_embedding, patch_embedding):
        return self.cross_modal_attention(text_embedding, patch_embedding)

    def aggregation(self, attended_text, temperature=2.0):
        return AggregationFn.apply(attended_text, temperature)

    # ---- tensor-in variant (synthetic configs, and the body of the path-based method)
    def compute_global_embedding_tensors(self, regions: torch.Tensor, input_ids: torch.Tensor,
                                         region_counts: Optional[torch.Tensor] = None,
                                         max_tokens: Optional[int] = None) -> torch.Tensor:
        """regions [B,R,3,S,S] in [0,1] (rows >= region_counts[b] ignored), input_ids [B,T] -> [B,E].
        Gradients flow only into cross_modal_attention (the towers run frozen, as in the reference:
        training/image_tokenizer.py:119, training/text_tokenizer.py:185)."""
        dev = self.device
        B, R = regions.shape[:2]
        with torch.no_grad():
            # The two frozen towers share nothing until the tokens are packed: on the GPU the text tower is launched on a
            # stream of its own (scratch lane 4) beside the region tower — its 19,712-row GEMMs have 154-616 tiles for 256
            # CUs, the region tower's LayerNorm / attention launches leave the matrix pipes idle (DCLIP_TEACHER_TEXT_STREAM=0:
            # one stream).  Not inside a HIP-graph capture (the fork would have to be part of every caller's warm-up).
            import os
            side = None
            if regions.is_cuda and input_ids.is_cuda and not torch.cuda.is_current_stream_capturing() \
                    and os.environ.get("DCLIP_TEACHER_TEXT_STREAM", "1") != "0":
                if getattr(self, "_text_side_stream", None) is None:
                    object.__setattr__(self, "_text_side_stream", torch.cuda.Stream(device=dev))
                side = self._text_side_stream
                here = torch.cuda.current_stream(dev)
                side.wait_stream(here)
                with torch.cuda.stream(side), ops.workspace_lane(4):
                    sent, tokens, eos = self.text_tokenizer.token_level_ids(input_ids)
            emb = self.patch_tokenizer.encode_regions(regions.reshape(B * R, *regions.shape[2:])).view(B, R, -1)
            if region_counts is not None:
                counts = region_counts.to(dev).to(torch.int32).contiguous()
                rmax = max(int(region_counts.max()), 1)        # an image without boxes keeps ONE zero row (:489-491)
                emb = ops.mask_rows(emb.contiguous(), counts)[:, :rmax].contiguous()
            if side is None:
                sent, tokens, eos = self.text_tokenizer.token_level_ids(input_ids)
            else:
                here.wait_stream(side)
                for t_ in (sent, tokens, eos):
                    t_.record_stream(here)
            self.last_sentence_embedding = sent       # text_projection(final_LN(h)[first EOS]) of THIS call's captions
            if max_tokens is None:
                max_tokens = max(int(eos.max()) - 1, 1)       # host sync; pass max_tokens to avoid it
            text = ops.pack_tokens(tokens.contiguous(), sent, eos, max_tokens)
        return self.global_embedding_from_tokens(text, emb)

    def global_embedding_from_tokens(self, text: torch.Tensor, patches: torch.Tensor) -> torch.Tensor:
        """[B,Tmax,E], [B,Rmax,E] (zero-padded) -> [B,E]: :634-651, with the reference's NaN / Inf guards: a region
        embedding that is not finite becomes a zero row (:497-499), a caption with any non-finite token embedding
        becomes all zeros (:542-544), a non-finite result zeroes the whole batch (:649-651)."""
        if not (text.requires_grad or patches.requires_grad):       # frozen tower outputs: guard in place on copies
            text, patches = text.detach().clone().contiguous(), patches.detach().clone().contiguous()
            ops.sanitize_groups(patches, 1)
            ops.sanitize_groups(text, text.shape[1])
        at, ai = self.cross_modal_attention(text, patches)
        out = GlobalPoolFn.apply(at, ai, 2.0)
        if out.grad_fn is not None and at.grad_fn is not None:
            at.grad_fn.replaced_flag = out.grad_fn.replaced_flag       # lets the block's backward honour guard (3)
        return out

    # ---- the reference's path-based signature
    def compute_global_embedding_batch(self, image_paths, texts, weighted_boxes_batch=None, images_u8=None, dims=None):
        """:268-656 with cached boxes: images are decoded on the host; crops, resize and everything after run batched
        on the GPU.  `images_u8` / `dims` (data.GpuCollate) hand over images that are already decoded and uploaded.
        `texts`: caption strings (needs a tokenizer) or an [B,T] id tensor."""
        from PIL import Image
        if weighted_boxes_batch is None:
            raise NotImplementedError("no detector here: pass weighted_boxes_batch (the reference's loader does)")
        if isinstance(weighted_boxes_batch, list):
            boxes = weighted_boxes_batch
        else:
            boxes = [weighted_boxes_batch.get(p, []) for p in image_paths]
        images = None
        if images_u8 is None:
            images = []
            for path in image_paths:
                try:
                    images.append(Image.open(path).convert("RGB"))
                except Exception:
                    images.append(Image.new("RGB", (224, 224)))        # the reference's fallback (:302)
        regions, counts = self.patch_tokenizer.crop_boxes_gpu(images, boxes, images_u8, dims)
        ids = self.text_tokenizer._ids(texts if isinstance(texts, torch.Tensor) else list(texts), keep_host=True)
        max_tokens = None
        if not ids.is_cuda:
            # token ids still on the host (tokenizer output): the longest caption's word-token count is known without
            # asking the GPU — no stream synchronisation in the middle of the step
            eos_id = self._clip.config.text.eos_token_id
            first_eos = (ids == eos_id).int().argmax(dim=1)
            max_tokens = max(int(first_eos.max()) - 1, 1)
            ids = ids.to(self.device)
        return self.compute_global_embedding_tensors(regions, ids, counts, max_tokens)
