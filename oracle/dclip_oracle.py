"""CPU oracle for the DCLIP distillation step.  TEST INFRASTRUCTURE — NOT PRODUCT.

A restatement, in elementary torch-CPU tensor ops, of the arithmetic the reference
runs on its hot path.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this module; nothing under `dclip_amd/`
does, and the product path raises if the HIP library is missing.

Pinning: every function here is checked in `tests/test_oracle_golden.py` against
`tests/golden/*.npz`, which `oracle/make_golden.py` produced by running the
reference's own code in the build container (its `CrossModalAttention` /
`aggregation` imported from training/patch_text_aggregation.py, its two loss
functions lifted from training/CLIP_image_distillation.py:532-576, and HF
`transformers` 5.15.0 `CLIPModel` — the third-party library the reference calls for
the towers, version unpinned by the reference: see DESIGN.md "Oracle").

All functions are dtype-generic: pass float64 tensors for a high-precision
reference, float32 to mirror the reference's `precision=32` run.  Gradients come
from torch autograd over these same ops.

Citations are `/root/reference/` paths, or `hf:` for the installed transformers.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------- primitives

def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.LayerNorm over the last dim (biased variance) — hf:modeling_clip.py:364-366;
    training/patch_text_aggregation.py:18-19."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * w + b


def quick_gelu(x: Tensor) -> Tensor:
    """x * sigmoid(1.702 x) — hf:activations.py:122-123 (CLIP `hidden_act="quick_gelu"`)."""
    return x * (1.0 / (1.0 + torch.exp(-1.702 * x)))


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    y = x @ w.transpose(-1, -2)
    return y if b is None else y + b


def softmax_lastdim(s: Tensor) -> Tensor:
    m = s.max(dim=-1, keepdim=True).values
    e = torch.exp(s - m)
    return e / e.sum(dim=-1, keepdim=True)


def l2_normalize(x: Tensor, eps: float = 1e-12) -> Tensor:
    """F.normalize(x, dim=1): x / max(||x||, eps) — CLIP_image_distillation.py:545-546, :569-570.
    vector_norm (not sqrt(sum)) so that a zero row back-propagates dy/eps like ATen does, not NaN."""
    n = torch.linalg.vector_norm(x, dim=-1, keepdim=True)
    return x / torch.clamp(n, min=eps)


# --------------------------------------------------------------------------- CLIP towers (HF arithmetic)

def clip_self_attention(x: Tensor, p: Dict[str, Tensor], pre: str, heads: int, causal: bool) -> Tensor:
    """hf:modeling_clip.py:298-335 (projections), :259-277 (eager softmax(QK^T*dh^-.5 + mask)V).
    Causal mask as in the text tower (hf:modeling_clip.py:546-551)."""
    B, S, D = x.shape
    dh = D // heads
    q = linear(x, p[f"{pre}.q_proj.weight"], p[f"{pre}.q_proj.bias"]).view(B, S, heads, dh).transpose(1, 2)
    k = linear(x, p[f"{pre}.k_proj.weight"], p[f"{pre}.k_proj.bias"]).view(B, S, heads, dh).transpose(1, 2)
    v = linear(x, p[f"{pre}.v_proj.weight"], p[f"{pre}.v_proj.bias"]).view(B, S, heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (dh ** -0.5)
    if causal:
        mask = torch.full((S, S), float("-inf"), dtype=x.dtype).triu(1)
        s = s + mask
    a = softmax_lastdim(s) @ v
    a = a.transpose(1, 2).reshape(B, S, D)
    return linear(a, p[f"{pre}.out_proj.weight"], p[f"{pre}.out_proj.bias"])


def clip_encoder_layer(x: Tensor, p: Dict[str, Tensor], pre: str, heads: int, causal: bool, eps: float) -> Tensor:
    """Pre-LN block — hf:modeling_clip.py:362-383; MLP :346-350."""
    h = layer_norm(x, p[f"{pre}.layer_norm1.weight"], p[f"{pre}.layer_norm1.bias"], eps)
    x = x + clip_self_attention(h, p, f"{pre}.self_attn", heads, causal)
    h = layer_norm(x, p[f"{pre}.layer_norm2.weight"], p[f"{pre}.layer_norm2.bias"], eps)
    h = quick_gelu(linear(h, p[f"{pre}.mlp.fc1.weight"], p[f"{pre}.mlp.fc1.bias"]))
    return x + linear(h, p[f"{pre}.mlp.fc2.weight"], p[f"{pre}.mlp.fc2.bias"])


def vision_tower(p: Dict[str, Tensor], pixel_values: Tensor, cfg, return_hidden: bool = False):
    """`CLIPModel.get_image_features` — call sites training/CLIP_image_distillation.py:601,
    training/image_tokenizer.py:120.  Math: patch conv (no bias) as im2col matmul
    (hf:modeling_clip.py:209-210), CLS concat + position (:212-217), pre_layrnorm (:642),
    encoder, CLS -> post_layernorm (:650-651), visual_projection without bias (:751).
    `cfg` is a `dclip_amd.config.VisionConfig`-like object; returns `[B, P]`."""
    v = cfg
    B = pixel_values.shape[0]
    ps, g = v.patch_size, v.grid
    w = p["vision_model.embeddings.patch_embedding.weight"].reshape(v.hidden_size, -1)
    # [B,C,g,ps,g,ps] -> [B,g,g,C,ps,ps] -> [B, g*g, C*ps*ps]  (row-major patch order, as conv+flatten(2))
    cols = pixel_values.reshape(B, v.num_channels, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, -1)
    patches = cols @ w.t()
    cls = p["vision_model.embeddings.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, patches], dim=1) + p["vision_model.embeddings.position_embedding.weight"]
    x = layer_norm(x, p["vision_model.pre_layrnorm.weight"], p["vision_model.pre_layrnorm.bias"], v.layer_norm_eps)
    hidden = [x]
    for i in range(v.num_hidden_layers):
        x = clip_encoder_layer(x, p, f"vision_model.encoder.layers.{i}", v.num_attention_heads, False,
                               v.layer_norm_eps)
        hidden.append(x)
    pooled = layer_norm(x[:, 0, :], p["vision_model.post_layernorm.weight"],
                        p["vision_model.post_layernorm.bias"], v.layer_norm_eps)
    emb = linear(pooled, p["visual_projection.weight"])
    return (emb, hidden) if return_hidden else emb


def first_eos_index(input_ids: Tensor, eos_token_id: int) -> Tensor:
    """First position of the EOS id (pads may equal EOS) — hf:modeling_clip.py:574-581."""
    return (input_ids == eos_token_id).int().argmax(dim=-1)


def text_tower(p: Dict[str, Tensor], input_ids: Tensor, cfg, return_hidden: bool = False):
    """`CLIPModel.get_text_features` — call sites training/CLIP_image_distillation.py:616,
    training/text_tokenizer.py:187-193.  token+position embedding (hf:modeling_clip.py:232-256),
    causal pre-LN encoder (:541-566), final_layer_norm (:568-569), first-EOS pooling (:574-581),
    text_projection without bias (:713).  Returns `[B, P]` (and the final-LN'd hidden states)."""
    t = cfg
    B, T = input_ids.shape
    x = p["text_model.embeddings.token_embedding.weight"][input_ids] \
        + p["text_model.embeddings.position_embedding.weight"][:T]
    for i in range(t.num_hidden_layers):
        x = clip_encoder_layer(x, p, f"text_model.encoder.layers.{i}", t.num_attention_heads, True,
                               t.layer_norm_eps)
    x = layer_norm(x, p["text_model.final_layer_norm.weight"], p["text_model.final_layer_norm.bias"],
                   t.layer_norm_eps)
    pooled = x[torch.arange(B), first_eos_index(input_ids, t.eos_token_id)]
    emb = linear(pooled, p["text_projection.weight"])
    return (emb, x) if return_hidden else emb


def teacher_token_embeddings(p: Dict[str, Tensor], input_ids: Tensor, cfg) -> Tuple[Tensor, Tensor, Tensor]:
    """`CLIPTextTokenizer.get_embeddings(text, return_token_level=True)` for a batch of padded ids
    (training/text_tokenizer.py:171-216).  The reference runs one caption at a time, unpadded, and keeps
    hidden rows i with 0 < i < len-1 (BOS and EOS dropped, :202), each through `text_projection` (:206).
    Under the causal mask rows < len are unaffected by trailing pads, so the batched, padded
    computation is identical.  A caption with no word tokens yields its sentence embedding (:210-212).

    Returns (tokens [B, Tmax, P] zero-padded as training/patch_text_aggregation.py:606-620 pads them,
    n_tokens [B], sentence [B, P])."""
    sent, hidden = text_tower(p, input_ids, cfg, return_hidden=True)
    B = input_ids.shape[0]
    eos = first_eos_index(input_ids, cfg.eos_token_id)           # = len-1
    n_tok = torch.clamp(eos - 1, min=0)
    proj = linear(hidden, p["text_projection.weight"])           # [B,T,P]
    Tmax = int(max(int(n_tok.max()), 1))
    out = torch.zeros(B, Tmax, proj.shape[-1], dtype=proj.dtype)
    counts = []
    for b in range(B):
        n = int(n_tok[b])
        if n == 0:
            out[b, 0] = sent[b]
            counts.append(1)
        else:
            out[b, :n] = proj[b, 1:1 + n]
            counts.append(n)
    return out, torch.tensor(counts), sent


# --------------------------------------------------------------------------- meta-teacher

def torch_mha(query: Tensor, key_value: Tensor, p: Dict[str, Tensor], pre: str, heads: int) -> Tensor:
    """`nn.MultiheadAttention(E, heads)` forward, batch-first restatement of the seq-first call at
    training/patch_text_aggregation.py:33,:42 — packed in_proj [3E,E] (+bias), softmax(QK^T/sqrt(dh))V,
    out_proj; dropout 0; NO key_padding_mask."""
    B, Lq, E = query.shape
    Lk = key_value.shape[1]
    dh = E // heads
    W, bvec = p[f"{pre}.in_proj_weight"], p[f"{pre}.in_proj_bias"]
    q = linear(query, W[:E], bvec[:E]).view(B, Lq, heads, dh).transpose(1, 2)
    k = linear(key_value, W[E:2 * E], bvec[E:2 * E]).view(B, Lk, heads, dh).transpose(1, 2)
    v = linear(key_value, W[2 * E:], bvec[2 * E:]).view(B, Lk, heads, dh).transpose(1, 2)
    a = softmax_lastdim((q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))) @ v
    a = a.transpose(1, 2).reshape(B, Lq, E)
    return linear(a, p[f"{pre}.out_proj.weight"], p[f"{pre}.out_proj.bias"])


def cross_modal_attention(p: Dict[str, Tensor], text: Tensor, patches: Tensor, heads: int = 8,
                          prefix: str = "") -> Tuple[Tensor, Tensor]:
    """`CrossModalAttention.forward` — training/patch_text_aggregation.py:21-46.  Both directions read the
    ORIGINAL text / patches (not sequential); LayerNorm eps is nn.LayerNorm's default 1e-5."""
    t_out = torch_mha(text, patches, p, f"{prefix}text_to_image", heads)
    t_out = layer_norm(text + t_out, p[f"{prefix}norm_text.weight"], p[f"{prefix}norm_text.bias"], 1e-5)
    i_out = torch_mha(patches, text, p, f"{prefix}image_to_text", heads)
    i_out = layer_norm(patches + i_out, p[f"{prefix}norm_image.weight"], p[f"{prefix}norm_image.bias"], 1e-5)
    return t_out, i_out


def aggregation(x: Tensor, temperature: float = 2.0) -> Tensor:
    """`PatchTextAggregation.aggregation` — training/patch_text_aggregation.py:243-265.
    F.cosine_similarity clamps EACH norm at eps=1e-8 (torch >= 1.12 semantics: x·y / (max(|x|,eps)·max(|y|,eps)))."""
    m = x.mean(dim=1, keepdim=True)
    nx = torch.clamp(torch.sqrt((x * x).sum(-1)), min=1e-8)
    nm = torch.clamp(torch.sqrt((m * m).sum(-1)), min=1e-8)
    sim = (x * m).sum(-1) / (nx * nm)
    w = softmax_lastdim(sim / temperature)
    return (x * w.unsqueeze(-1)).sum(dim=1)


def global_embedding(p: Dict[str, Tensor], text: Tensor, patches: Tensor, heads: int = 8, prefix: str = "") -> Tensor:
    """Tail of `compute_global_embedding_batch` — training/patch_text_aggregation.py:634-647:
    cross-attention, aggregate each side, 0.5/0.5 mix.  Inputs are already zero-padded
    ([B,Tmax,E], [B,Rmax,E]); padded rows are attended to and pooled."""
    # NaN / Inf guards: a non-finite region embedding -> zero row (:497-499); a caption with any non-finite token
    # embedding -> all zeros (:542-544); a non-finite result -> the whole batch zeros (:649-651)
    row_ok = torch.isfinite(patches).all(dim=-1, keepdim=True)
    patches = torch.where(row_ok, patches, torch.zeros_like(patches))
    cap_ok = torch.isfinite(text).flatten(1).all(dim=1).view(-1, 1, 1)
    text = torch.where(cap_ok, text, torch.zeros_like(text))
    at, ai = cross_modal_attention(p, text, patches, heads, prefix)
    out = 0.5 * aggregation(at) + 0.5 * aggregation(ai)
    if not bool(torch.isfinite(out).all()):
        out = torch.zeros_like(out)
    return out


def pad_regions(region_embs: Sequence[Tensor], embed_dim: int) -> Tensor:
    """Zero-pad per-image region embeddings to Rmax — training/patch_text_aggregation.py:555-581;
    an image with no boxes contributes ONE zero row (:489-491)."""
    rows = [r if r.shape[0] > 0 else torch.zeros(1, embed_dim, dtype=r.dtype) for r in region_embs]
    rmax = max(r.shape[0] for r in rows)
    out = torch.zeros(len(rows), rmax, embed_dim, dtype=rows[0].dtype)
    for b, r in enumerate(rows):
        out[b, :r.shape[0]] = r
    return out


# --------------------------------------------------------------------------- losses

def contrastive_loss(image_emb: Tensor, text_emb: Tensor, temperature: float = 0.05) -> Tensor:
    """`compute_contrastive_loss` — training/CLIP_image_distillation.py:532-562 (duplicate at
    training/train_contrastive_teacher.py:251-261).  Temperature is the constant 0.05, not logit_scale."""
    i = l2_normalize(image_emb)
    t = l2_normalize(text_emb)
    z = (i @ t.t()) / temperature
    n = z.shape[0]

    def ce_diag(logits):
        m = logits.max(dim=1, keepdim=True).values
        lse = m.squeeze(1) + torch.log(torch.exp(logits - m).sum(dim=1))
        return (lse - logits.diagonal()).sum() / n

    return (ce_diag(z) + ce_diag(z.t())) / 2.0


def cosine_distillation_loss(student: Tensor, teacher: Tensor) -> Tensor:
    """`cosine_distillation_loss` — training/CLIP_image_distillation.py:564-576."""
    return (1.0 - (l2_normalize(student) * l2_normalize(teacher)).sum(dim=1)).mean()


# --------------------------------------------------------------------------- the step

def distill_step(student: Dict[str, Tensor], cfg, pixel_values: Tensor, input_ids: Tensor,
                 teacher_image_emb: Tensor, teacher_text_emb: Optional[Tensor] = None,
                 temperature: float = 0.05) -> Dict[str, Tensor]:
    """`CLIPImageDistillation.training_step` arithmetic — training/CLIP_image_distillation.py:580-634 —
    with the teacher image embedding given (it is computed under no_grad, :597-600).  If
    `teacher_text_emb` is None the teacher sentence embedding is the (frozen) student text tower's own
    output, the north_star regime in which both text towers share one forward."""
    s_img = vision_tower(student, pixel_values, cfg.vision)
    s_txt = text_tower(student, input_ids, cfg.text)
    t_txt = s_txt.detach() if teacher_text_emb is None else teacher_text_emb
    l_img = cosine_distillation_loss(s_img, teacher_image_emb)
    l_txt = cosine_distillation_loss(s_txt, t_txt)
    l_con = contrastive_loss(s_img, s_txt, temperature)
    return {"loss": l_img + l_txt + 1.0 * l_con, "loss_image": l_img, "loss_text": l_txt,
            "loss_contrastive": l_con, "image_emb": s_img, "text_emb": s_txt}


def bridge_weight(student_dim: int, teacher_dim: int, seed: int = 0) -> Tensor:
    """Config c5's DECLARED teacher→student bridge (no reference semantics: training/CLIP_image_distillation.py:573
    would raise on the width mismatch; training/patch_text_aggregation.py:51 "FIX THIS IF GOING FROM VIT L TO VIT B").
    Seeded Gaussian [student_dim, teacher_dim], std teacher_dim^-1/2 — restated here independently of the product's
    `dclip_amd.CLIP_image_distillation.bridge_weight`; tests compare the two bit for bit."""
    gen = torch.Generator().manual_seed(1_000_003 + int(seed))
    return torch.randn((student_dim, teacher_dim), generator=gen, dtype=torch.float32) * float(teacher_dim) ** -0.5


def teacher_targets(teacher: Dict[str, Tensor], tcfg, cm: Dict[str, Tensor], regions: Tensor, region_counts: Sequence[int],
                    input_ids: Tensor, heads: int, prefix: str = "") -> Tuple[Tensor, Tensor]:
    """The meta-teacher's two targets for one batch, from a teacher CLIP state dict `teacher` of config `tcfg`:
    global image embedding (a5 → a8 → a6 → a7: frozen region forward in [0,1] without mean/std, zero padding to Rmax,
    token-level text, cross-modal attention, aggregation, 0.5/0.5 — training/patch_text_aggregation.py:268-656) and
    the sentence embedding of each caption (`aggregate_text`, training/text_tokenizer.py:220-235)."""
    B, R = regions.shape[:2]
    E = tcfg.projection_dim
    embs = []
    for b in range(B):
        n = int(region_counts[b])
        embs.append(vision_tower(teacher, regions[b, :n], tcfg.vision) if n > 0 else torch.zeros(0, E, dtype=regions.dtype))
    patches = pad_regions(embs, E)
    tokens, _n, sent = teacher_token_embeddings(teacher, input_ids, tcfg.text)
    return global_embedding(cm, tokens, patches, heads, prefix), sent


def distill_step_bridged(student: Dict[str, Tensor], cfg, pixel_values: Tensor, input_ids: Tensor,
                         teacher_image_emb: Tensor, teacher_text_emb: Tensor, bridge_w: Tensor,
                         temperature: float = 0.05) -> Dict[str, Tensor]:
    """Config c5: `distill_step` with both (frozen, wider) teacher targets taken through the bridge first."""
    return distill_step(student, cfg, pixel_values, input_ids, linear(teacher_image_emb, bridge_w),
                        linear(teacher_text_emb, bridge_w), temperature)


def teacher_step(cm: Dict[str, Tensor], text_tokens: Tensor, region_embs: Tensor, sentence_emb: Tensor,
                 heads: int = 8, temperature: float = 0.05, prefix: str = "") -> Dict[str, Tensor]:
    """Teacher-trainer step — training/train_contrastive_teacher.py:340-357: meta-teacher image embedding
    vs CLIP sentence embedding under the symmetric InfoNCE of :251-261."""
    img = global_embedding(cm, text_tokens, region_embs, heads, prefix)
    return {"loss": contrastive_loss(img, sentence_emb, temperature), "image_emb": img}


def to_dtype(sd: Dict[str, Tensor], dtype) -> Dict[str, Tensor]:
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
