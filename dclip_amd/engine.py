"""Explicit forward / backward schedules of the CLIP towers over the C-ABI kernels.

No tracing compiler and no autograd inside a tower: each function below is the fixed launch
sequence for one pre-LN transformer stack (hf:modeling_clip.py:362-383) and its hand-derived
backward.  `functional.py` wraps a whole tower as ONE torch.autograd.Function so that the
reference's training scripts (`loss.backward()`, torch optimizers) keep working unchanged.

Shapes: activations are [M = B*S, D] row-major.  Per layer the forward keeps
(x, ln1, qkv, attn, lse, x1, ln2, h, g + LN statistics) for the backward; with 288 GB of HBM per
MI355X nothing is recomputed (12 layers x ~630 MB at B=256, S=50, D=768).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import os

import torch

from . import _lib, ops


@dataclass
class LayerParams:
    ln1_w: torch.Tensor
    ln1_b: torch.Tensor
    qkv_w: torch.Tensor      # [3D, D] = rows [q | k | v], the in-memory fusion of q_proj/k_proj/v_proj
    qkv_b: torch.Tensor
    out_w: torch.Tensor
    out_b: torch.Tensor
    ln2_w: torch.Tensor
    ln2_b: torch.Tensor
    fc1_w: torch.Tensor
    fc1_b: torch.Tensor
    fc2_w: torch.Tensor
    fc2_b: torch.Tensor

    FIELDS = ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w",
              "fc2_b")

    def tensors(self):
        return [getattr(self, f) for f in self.FIELDS]


def layer_fwd(x, p: LayerParams, B: int, S: int, H: int, causal: bool, eps: float, save: bool):
    ln1, m1, r1 = ops.layernorm_fwd(x, p.ln1_w, p.ln1_b, eps, save_stats=save)
    qkv = ops.gemm(ln1, p.qkv_w, ops.LAYOUT_NT, bias=p.qkv_b)
    attn, lse = ops.attention_fwd(qkv, B, S, H, causal)
    x1 = ops.gemm(attn, p.out_w, ops.LAYOUT_NT, bias=p.out_b, residual=x)
    ln2, m2, r2 = ops.layernorm_fwd(x1, p.ln2_w, p.ln2_b, eps, save_stats=save)
    h = torch.empty((x.shape[0], p.fc1_w.shape[0]), dtype=torch.float32, device=x.device) if save else None
    g = ops.gemm(ln2, p.fc1_w, ops.LAYOUT_NT, bias=p.fc1_b, aux=h, epilogue=ops.EPI_GELU)
    x2 = ops.gemm(g, p.fc2_w, ops.LAYOUT_NT, bias=p.fc2_b, residual=x1)
    saved = (x, m1, r1, ln1, qkv, attn, lse, x1, m2, r2, ln2, h, g) if save else None
    return x2, saved


def _fresh(_name: str, shape, device) -> torch.Tensor:
    return torch.empty(shape, dtype=torch.float32, device=device)


def _galloc(alloc, name: str, shape, device) -> torch.Tensor:
    """Where a parameter gradient is written.  `alloc(name, shape)` (data parallel: dist.GradSync hands out the
    parameter's slice of its persistent all-reduce bucket, so the wgrad GEMM writes straight into it) may return None."""
    t = alloc(name, shape) if alloc is not None else None
    return t if t is not None else _fresh(name, shape, device)


def linear_param_grads(dy, x, need_w: bool, need_b: bool, gr: Dict[str, torch.Tensor], wkey: str, bkey: str, alloc=None):
    """dW = dy^T x and db = colsum(dy) of one nn.Linear.  When both are wanted the bias gradient comes out of the
    weight-gradient GEMM (DCLIP_EPI_A_ROWSUM): dy is streamed once, no separate column-sum launch."""
    dev = dy.device
    if need_w and need_b:
        gr[bkey] = _galloc(alloc, bkey, (dy.shape[1],), dev)
        gr[wkey] = ops.gemm(dy, x, ops.LAYOUT_TN, out=_galloc(alloc, wkey, (dy.shape[1], x.shape[1]), dev), a_rowsum=gr[bkey])
    elif need_w:
        gr[wkey] = ops.gemm(dy, x, ops.LAYOUT_TN, out=_galloc(alloc, wkey, (dy.shape[1], x.shape[1]), dev))
    elif need_b:
        gr[bkey] = ops.colsum(dy, out=_galloc(alloc, bkey, (dy.shape[1],), dev))


def _ln_bwd(dy, x, gamma, mean, rstd, dresidual, want: bool, gr, wkey: str, bkey: str, alloc):
    """LayerNorm backward; dγ / dβ land where `alloc` says."""
    if want:
        dg = _galloc(alloc, wkey, tuple(gamma.shape), dy.device)
        db = _galloc(alloc, bkey, tuple(gamma.shape), dy.device)
        dx, dg, db = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dresidual=dresidual, dgamma=dg, dbeta=db, accumulate=False)
        gr[wkey], gr[bkey] = dg, db
        return dx
    dx, _, _ = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dresidual=dresidual, need_param_grads=False)
    return dx


def layer_bwd(dx2, p: LayerParams, saved, B: int, S: int, H: int, causal: bool, need: Dict[str, bool], alloc=None):
    """Returns (dx, grads) with grads keyed like LayerParams.FIELDS (missing = not needed).  `alloc(field, shape)`:
    see _galloc."""
    x, m1, r1, ln1, qkv, attn, lse, x1, m2, r2, ln2, h, g = saved
    gr: Dict[str, torch.Tensor] = {}
    linear_param_grads(dx2, g, bool(need.get("fc2_w")), bool(need.get("fc2_b")), gr, "fc2_w", "fc2_b", alloc)
    dh = ops.gemm(dx2, p.fc2_w, ops.LAYOUT_NN, aux=h, epilogue=ops.EPI_DGELU)
    linear_param_grads(dh, ln2, bool(need.get("fc1_w")), bool(need.get("fc1_b")), gr, "fc1_w", "fc1_b", alloc)
    dln2 = ops.gemm(dh, p.fc1_w, ops.LAYOUT_NN)
    del dh
    dx1 = _ln_bwd(dln2, x1, p.ln2_w, m2, r2, dx2, bool(need.get("ln2_w") or need.get("ln2_b")), gr, "ln2_w", "ln2_b", alloc)
    linear_param_grads(dx1, attn, bool(need.get("out_w")), bool(need.get("out_b")), gr, "out_w", "out_b", alloc)
    dattn = ops.gemm(dx1, p.out_w, ops.LAYOUT_NN)
    dqkv = ops.attention_bwd(qkv, attn, dattn, lse, B, S, H, causal)
    linear_param_grads(dqkv, ln1, bool(need.get("qkv_w")), bool(need.get("qkv_b")), gr, "qkv_w", "qkv_b", alloc)
    dln1 = ops.gemm(dqkv, p.qkv_w, ops.LAYOUT_NN)
    del dqkv
    dx = _ln_bwd(dln1, x, p.ln1_w, m1, r1, dx1, bool(need.get("ln1_w") or need.get("ln1_b")), gr, "ln1_w", "ln1_b", alloc)
    return dx, gr


# --------------------------------------------------------------------------------------------- last vision layer
# After the final encoder layer the model reads ONLY the CLS row of each image (hf:modeling_clip.py:650-651).  In
# that layer everything downstream of the attention — out_proj, LayerNorm2, fc1, quick-GELU, fc2 and both residual
# adds — is row-wise, so it is evaluated for the B CLS rows instead of all B*S rows, and the attention itself for
# one query row per (image, head).  In the backward the incoming gradient is nonzero only on those rows, so the
# same restriction is exact there too (the other rows' contributions to every weight gradient are products with
# zero).  K and V, hence the qkv projection, LayerNorm1 and their gradients, stay full size.  Results are identical
# to the unpruned schedule (tests/test_model_gpu.py); 9/12 of that layer's GEMM work disappears.

def last_layer_fwd_cls(x, p: LayerParams, B: int, S: int, H: int, eps: float, save: bool):
    D = x.shape[1]
    ln1, m1, r1 = ops.layernorm_fwd(x, p.ln1_w, p.ln1_b, eps, save_stats=save)
    qkv = ops.gemm(ln1, p.qkv_w, ops.LAYOUT_NT, bias=p.qkv_b)
    attn, lse = ops.attention_cls_fwd(qkv, B, S, H)                       # [B, D]
    x_cls = ops.gather_rows(x, None, B, S, D)
    x1 = ops.gemm(attn, p.out_w, ops.LAYOUT_NT, bias=p.out_b, residual=x_cls)
    ln2, m2, r2 = ops.layernorm_fwd(x1, p.ln2_w, p.ln2_b, eps, save_stats=save)
    h = torch.empty((B, p.fc1_w.shape[0]), dtype=torch.float32, device=x.device) if save else None
    g = ops.gemm(ln2, p.fc1_w, ops.LAYOUT_NT, bias=p.fc1_b, aux=h, epilogue=ops.EPI_GELU)
    x2 = ops.gemm(g, p.fc2_w, ops.LAYOUT_NT, bias=p.fc2_b, residual=x1)   # [B, D] = final hidden state, CLS rows
    saved = (x, m1, r1, ln1, qkv, attn, lse, x1, m2, r2, ln2, h, g) if save else None
    return x2, saved


def last_layer_bwd_cls(dx2, p: LayerParams, saved, B: int, S: int, H: int, need: Dict[str, bool], alloc=None):
    """dx2 [B, D] is the gradient w.r.t. the CLS rows of the final hidden state; returns (dx [B*S, D], grads)."""
    x, m1, r1, ln1, qkv, attn, lse, x1, m2, r2, ln2, h, g = saved
    D = x.shape[1]
    gr: Dict[str, torch.Tensor] = {}
    linear_param_grads(dx2, g, bool(need.get("fc2_w")), bool(need.get("fc2_b")), gr, "fc2_w", "fc2_b", alloc)
    dh = ops.gemm(dx2, p.fc2_w, ops.LAYOUT_NN, aux=h, epilogue=ops.EPI_DGELU)
    linear_param_grads(dh, ln2, bool(need.get("fc1_w")), bool(need.get("fc1_b")), gr, "fc1_w", "fc1_b", alloc)
    dln2 = ops.gemm(dh, p.fc1_w, ops.LAYOUT_NN)
    dx1 = _ln_bwd(dln2, x1, p.ln2_w, m2, r2, dx2, bool(need.get("ln2_w") or need.get("ln2_b")), gr, "ln2_w", "ln2_b", alloc)
    linear_param_grads(dx1, attn, bool(need.get("out_w")), bool(need.get("out_b")), gr, "out_w", "out_b", alloc)
    dattn = ops.gemm(dx1, p.out_w, ops.LAYOUT_NN)
    dqkv = ops.attention_cls_bwd(qkv, attn, dattn, lse, B, S, H)          # [B*S, 3D]; d q only on the CLS rows
    linear_param_grads(dqkv, ln1, bool(need.get("qkv_w")), bool(need.get("qkv_b")), gr, "qkv_w", "qkv_b", alloc)
    dln1 = ops.gemm(dqkv, p.qkv_w, ops.LAYOUT_NN)
    del dqkv
    dres = ops.scatter_rows(dx1, None, B, S, D)                           # the skip connection carries dx1 on CLS rows only
    dx = _ln_bwd(dln1, x, p.ln1_w, m1, r1, dres, bool(need.get("ln1_w") or need.get("ln1_b")), gr, "ln1_w", "ln1_b", alloc)
    return dx, gr


# --------------------------------------------------------------------------------------------- vision tower

@dataclass
class VisionParams:
    class_embedding: torch.Tensor
    patch_w: torch.Tensor        # [D, C, p, p]
    pos: torch.Tensor            # [S, D]
    pre_w: torch.Tensor
    pre_b: torch.Tensor
    layers: List[LayerParams]
    post_w: torch.Tensor
    post_b: torch.Tensor
    proj_w: torch.Tensor         # visual_projection [P, D]

    HEAD = ("class_embedding", "patch_w", "pos", "pre_w", "pre_b")
    TAIL = ("post_w", "post_b", "proj_w")

    def tensors(self):
        out = [getattr(self, f) for f in self.HEAD]
        for l in self.layers:
            out += l.tensors()
        out += [getattr(self, f) for f in self.TAIL]
        return out

    @classmethod
    def from_tensors(cls, ts, n_layers):
        ts = list(ts)
        head = ts[:5]
        layers = [LayerParams(*ts[5 + 12 * i: 5 + 12 * (i + 1)]) for i in range(n_layers)]
        tail = ts[5 + 12 * n_layers:]
        return cls(*head, layers, *tail)

    def names(self):
        out = list(self.HEAD)
        for i in range(len(self.layers)):
            out += [f"layers.{i}.{f}" for f in LayerParams.FIELDS]
        return out + list(self.TAIL)


def vision_fwd(p: VisionParams, pixel_values: torch.Tensor, cfg, save: bool, hidden_out: Optional[list] = None):
    """get_image_features: [B,3,H,W] -> [B,P]  (hf:modeling_clip.py:202-218, :641-651, :744-751)."""
    v = cfg
    B = pixel_values.shape[0]
    S, D, H = v.seq_len, v.hidden_size, v.num_attention_heads
    cols = ops.im2col(pixel_values, v.patch_size)
    patch = ops.gemm(cols, p.patch_w.view(D, -1), ops.LAYOUT_NT)
    emb = ops.vision_assemble_fwd(patch, p.class_embedding, p.pos, B, S, D)
    del patch
    x, m0, r0 = ops.layernorm_fwd(emb, p.pre_w, p.pre_b, v.layer_norm_eps, save_stats=save)
    if hidden_out is not None:
        hidden_out.append(x)
    saved_layers = []
    prune = hidden_out is None and len(p.layers) > 0          # full hidden states are only materialised on request
    for li, lp in enumerate(p.layers):
        if prune and li == len(p.layers) - 1:
            cls_tok, sv = last_layer_fwd_cls(x, lp, B, S, H, v.layer_norm_eps, save)
            saved_layers.append(sv)
            break
        x, sv = layer_fwd(x, lp, B, S, H, False, v.layer_norm_eps, save)
        saved_layers.append(sv)
        if hidden_out is not None:
            hidden_out.append(x)
    if not prune:
        cls_tok = ops.gather_rows(x, None, B, S, D)
    pooled, mp, rp = ops.layernorm_fwd(cls_tok, p.post_w, p.post_b, v.layer_norm_eps, save_stats=save)
    out = ops.gemm(pooled, p.proj_w, ops.LAYOUT_NT)
    saved = (cols, emb, m0, r0, saved_layers, cls_tok, mp, rp, pooled, prune) if save else None
    return out, saved


def vision_bwd(p: VisionParams, saved, d_out: torch.Tensor, cfg, need: List[bool], on_ready=None, alloc=None):
    """Gradients for VisionParams.tensors() order (None where not needed).  `on_ready(dict name -> grad)` is called as
    soon as a group of gradients is final (the tail, then each layer from the top down, then the head): the
    data-parallel all-reduce of that group starts while the layers below are still being back-propagated.
    `alloc(name, shape)` may name the tensor a parameter gradient is to be written into (see _galloc)."""
    v = cfg
    cols, emb, m0, r0, saved_layers, cls_tok, mp, rp, pooled, pruned = saved
    B = cls_tok.shape[0]
    S, D, H = v.seq_len, v.hidden_size, v.num_attention_heads
    names = p.names()
    needd = dict(zip(names, need))
    grads: Dict[str, Optional[torch.Tensor]] = {n: None for n in names}
    dev = d_out.device
    if needd["proj_w"]:
        grads["proj_w"] = ops.gemm(d_out, pooled, ops.LAYOUT_TN, out=_galloc(alloc, "proj_w", tuple(p.proj_w.shape), dev))
    dpooled = ops.gemm(d_out, p.proj_w, ops.LAYOUT_NN)
    dcls = _ln_bwd(dpooled, cls_tok, p.post_w, mp, rp, None, bool(needd["post_w"] or needd["post_b"]), grads, "post_w", "post_b",
                   alloc)
    if on_ready is not None:
        on_ready({n: grads[n] for n in VisionParams.TAIL if grads[n] is not None})
    n_layers = len(p.layers)
    # stop as soon as nothing below still needs a gradient (e.g. only visual_projection trainable)
    lowest = None
    for i, n in enumerate(names):
        if needd[n] and n not in VisionParams.TAIL:
            li = -1 if n in VisionParams.HEAD else int(n.split(".")[1])
            lowest = li if lowest is None else min(lowest, li)
    if lowest is None:
        return [grads[n] for n in names]
    dx = None if pruned else ops.scatter_rows(dcls, None, B, S, D)
    for i in range(n_layers - 1, max(lowest, 0) - 1, -1):
        lneed = {f: needd[f"layers.{i}.{f}"] for f in LayerParams.FIELDS}
        lalloc = None if alloc is None else (lambda f, shape, i=i: alloc(f"layers.{i}.{f}", shape))
        if pruned and i == n_layers - 1:
            dx, gr = last_layer_bwd_cls(dcls, p.layers[i], saved_layers[i], B, S, H, lneed, lalloc)
        else:
            dx, gr = layer_bwd(dx, p.layers[i], saved_layers[i], B, S, H, False, lneed, lalloc)
        saved_layers[i] = None
        for f, t in gr.items():
            grads[f"layers.{i}.{f}"] = t
        if on_ready is not None:
            on_ready({f"layers.{i}.{f}": t for f, t in gr.items()})
    if lowest < 0:
        demb = _ln_bwd(dx, emb, p.pre_w, m0, r0, None, bool(needd["pre_w"] or needd["pre_b"]), grads, "pre_w", "pre_b", alloc)
        if needd["pos"] or needd["class_embedding"]:
            dpos = ops.colsum(demb.view(B, S * D), out=(_galloc(alloc, "pos", (S * D,), dev) if needd["pos"] else None))
            if needd["pos"]:
                grads["pos"] = dpos.view(S, D)
            if needd["class_embedding"]:
                ce = _galloc(alloc, "class_embedding", (D,), dev)
                ce.copy_(dpos[:D])
                grads["class_embedding"] = ce
        if needd["patch_w"]:
            dpatch = ops.vision_assemble_bwd(demb, B, S, D)
            pw = _galloc(alloc, "patch_w", (D, p.patch_w.numel() // D), dev)
            grads["patch_w"] = ops.gemm(dpatch, cols, ops.LAYOUT_TN, out=pw).view_as(p.patch_w)
        if on_ready is not None:
            on_ready({n: grads[n] for n in VisionParams.HEAD if grads[n] is not None})
    return [grads[n] for n in names]


# --------------------------------------------------------------------------------------------- frozen towers in bf16
# Opt-in mixed precision for towers that never receive gradients (the teacher's region encoder, the frozen text
# tower; BASELINE configs c3 / c5): GEMM inputs are bf16 (weights converted once, activations converted by the
# producing kernel), accumulation, residual stream, LayerNorm statistics, softmax and biases stay fp32.

def _w16(cache: dict, key: str, w: torch.Tensor) -> torch.Tensor:
    """bf16 copy of a GEMM weight, PERSISTENT: entry = [tensor, version of `w` it was made from, data pointer of `w`].
    When the optimizer (or load_state_dict) has written `w` since — the version counter is shared with the Parameter a
    detached view came from — the copy is refreshed IN PLACE, so the buffer a captured HIP graph reads stays the one
    that is kept current (and a capture started on a stale entry records the cast: every replay then re-reads the fp32
    masters, HipCLIPModel.invalidate_bf16_of_trainable)."""
    e = cache.get(key)
    src = w.detach().reshape(w.shape[0], -1)
    if e is None:
        e = [ops.cast_bf16(src.contiguous()), w._version, w.data_ptr()]
        cache[key] = e
    elif e[1] != w._version or e[2] != w.data_ptr():
        ops.cast_bf16(src.contiguous(), out=e[0])
        e[1], e[2] = w._version, w.data_ptr()
    return e[0]


def _layer_fwd_bf16(x, p: LayerParams, c: dict, pre: str, B: int, S: int, H: int, causal: bool, eps: float):
    ln1 = ops.layernorm_fwd_bf16(x, p.ln1_w, p.ln1_b, eps)
    qkv = ops.gemm_bf16(ln1, _w16(c, pre + "qkv", p.qkv_w), bias=p.qkv_b, out_bf16=True)
    attn = ops.attention_fwd_bf16(qkv, B, S, H, causal)       # bf16 q/k/v in, bf16 context out: no cast pass
    x1 = ops.gemm_bf16(attn, _w16(c, pre + "out", p.out_w), bias=p.out_b, residual=x)
    ln2 = ops.layernorm_fwd_bf16(x1, p.ln2_w, p.ln2_b, eps)
    g = ops.gemm_bf16(ln2, _w16(c, pre + "fc1", p.fc1_w), bias=p.fc1_b, gelu=True, out_bf16=True)
    return ops.gemm_bf16(g, _w16(c, pre + "fc2", p.fc2_w), bias=p.fc2_b, residual=x1)


def vision_fwd_bf16(p: VisionParams, pixel_values: torch.Tensor, cfg, cache: dict) -> torch.Tensor:
    """Frozen get_image_features with bf16 GEMM inputs; `cache` keeps the converted weights between calls."""
    v = cfg
    B = pixel_values.shape[0]
    S, D, H = v.seq_len, v.hidden_size, v.num_attention_heads
    cols = (ops.im2col_bf16(pixel_values, v.patch_size) if v.patch_size % 4 == 0        # one pass: gather + round
            else ops.cast_bf16(ops.im2col(pixel_values, v.patch_size)))
    patch = ops.gemm_bf16(cols, _w16(cache, "patch", p.patch_w), k=v.patch_dim)
    x, _, _ = ops.layernorm_fwd(ops.vision_assemble_fwd(patch, p.class_embedding, p.pos, B, S, D), p.pre_w, p.pre_b,
                                v.layer_norm_eps, save_stats=False)
    for li, lp in enumerate(p.layers[:-1]):
        x = _layer_fwd_bf16(x, lp, cache, f"v{li}.", B, S, H, False, v.layer_norm_eps)
    lp, pre = p.layers[-1], f"v{len(p.layers) - 1}."
    # last layer on the CLS rows only (see last_layer_fwd_cls)
    ln1 = ops.layernorm_fwd_bf16(x, lp.ln1_w, lp.ln1_b, v.layer_norm_eps)
    if S <= 512 and os.environ.get("DCLIP_BF16_ROW_ATTN", "1") != "0":
        # q | k | v written as bf16 (half the bytes of the 2304-wide projection's output), one-row kernel on them
        qkv = ops.gemm_bf16(ln1, _w16(cache, pre + "qkv", lp.qkv_w), bias=lp.qkv_b, out_bf16=True)
        attn16 = ops.attention_row_fwd_bf16(qkv, None, B, S, H)
    else:
        qkv = ops.gemm_bf16(ln1, _w16(cache, pre + "qkv", lp.qkv_w), bias=lp.qkv_b)
        attn16 = ops.cast_bf16(ops.attention_cls_fwd(qkv, B, S, H)[0])
    x1 = ops.gemm_bf16(attn16, _w16(cache, pre + "out", lp.out_w), bias=lp.out_b,
                       residual=ops.gather_rows(x, None, B, S, D))
    ln2 = ops.layernorm_fwd_bf16(x1, lp.ln2_w, lp.ln2_b, v.layer_norm_eps)
    g = ops.gemm_bf16(ln2, _w16(cache, pre + "fc1", lp.fc1_w), bias=lp.fc1_b, gelu=True, out_bf16=True)
    cls_tok = ops.gemm_bf16(g, _w16(cache, pre + "fc2", lp.fc2_w), bias=lp.fc2_b, residual=x1)
    pooled, _, _ = ops.layernorm_fwd(cls_tok, p.post_w, p.post_b, v.layer_norm_eps, save_stats=False)
    return ops.gemm(pooled, p.proj_w, ops.LAYOUT_NT)          # [B,D] x [P,D]: tiny, kept in exact fp32


# --------------------------------------------------------------------------------------------- bf16 TRAINING (vision)
# The student's vision tower with bf16 GEMM inputs in forward, dgrad AND wgrad (BASELINE configs c3 / c5 quote the step
# in bf16; opt-in `precision="bf16"` with gradients enabled).  fp32 master weights (bf16 copies W and W^T are rebuilt
# when the optimizer has stepped), fp32 accumulation, fp32 residual stream / LayerNorm statistics / softmax; the
# attention core runs on the bf16 MFMAs for sequences up to 64 tokens (attention_bf16.hip; the fp32 kernels above that).
# The weight-gradient product dW[out,in] = dY^T X comes from the token-major split-K kernel on the operands as the backward
# has them (for shapes it declines: the C = A W^T kernel on dY^T, X^T written by ops.transpose_bf16).  The patch embedding is
# a bf16 GEMM pair too where that kernel takes its shape (_patch_embed_bf16); the pooled LayerNorm and the projection stay
# fp32; every encoder layer is run at full size (the CLS-row pruning of the fp32 schedule would keep a full-size fp32 qkv
# projection).

def _w16t(cache: dict, key: str, w: torch.Tensor) -> torch.Tensor:
    """bf16 W^T [in, ld >= out] of an nn.Linear weight [out, in]: the `W` operand of the dgrad GEMM dX = dY (W^T)^T."""
    e = cache.get(key + ".T")
    src = w.detach().reshape(w.shape[0], -1)
    if e is None:
        e = [ops.transpose_bf16(src.contiguous()), w._version, w.data_ptr()]
        cache[key + ".T"] = e
    elif e[1] != w._version or e[2] != w.data_ptr():
        ops.transpose_bf16(src.contiguous(), out=e[0])
        e[1], e[2] = w._version, w.data_ptr()
    return e[0]


_TRAIN_WEIGHTS = (("qkv", "qkv_w"), ("out", "out_w"), ("fc1", "fc1_w"), ("fc2", "fc2_w"))


def refresh_train_weights(cache: dict, layers: List[LayerParams], prefix: str = "v") -> None:
    """Bring the bf16 copies W and W^T of every encoder-layer GEMM weight of a TRAINING tower up to date before its
    forward — in ONE launch (ops.mt_weights_bf16) when, as after every optimizer step, all of them are stale: the
    per-weight path is 48 casts + 48 transposes per step for ViT-B.  The record table lives on the device next to the
    persistent copies (uploaded once, so the launch is capturable in a HIP graph)."""
    stale = []
    for li, lp in enumerate(layers):
        for short, field in _TRAIN_WEIGHTS:
            w = getattr(lp, field)
            for key in (f"{prefix}{li}.{short}", f"{prefix}{li}.{short}.T"):
                e = cache.get(key)
                if e is None or e[1] != w._version or e[2] != w.data_ptr():
                    stale.append((key, w))
    if not stale:
        return
    tables = cache.setdefault("__mt_tables__", {})
    keys = tuple(k for k, _ in stale)
    tab = tables.get(keys)
    if tab is not None and all(cache[k][2] == w.data_ptr() for k, w in stale):
        ops.mt_weights_bf16(tab["dev"], tab["n"], tab["tiles"])
        for k, w in stale:
            cache[k][1] = w._version
        return
    # first time this set is stale: the per-weight kernels allocate / refresh the copies ...
    for key, w in stale:
        if key.endswith(".T"):
            _w16t(cache, key[:-2], w)
        else:
            _w16(cache, key, w)
    # ... and a table for exactly this set (all weights after an optimizer step; the trainable subset under a freeze rule)
    # is built for the following steps — not while a stream is capturing: the upload is a synchronous copy
    if torch.cuda.is_current_stream_capturing():
        return
    recs = {}
    for key, w in stale:
        base = key[:-2] if key.endswith(".T") else key
        r = recs.setdefault(base, [w, None, None])
        r[2 if key.endswith(".T") else 1] = cache[key][0]
    dev, n, tiles = ops.mt_weights_table([tuple(r) for r in recs.values()])
    tables[keys] = {"dev": dev, "n": n, "tiles": tiles}


def _attention_io16(M: int, S: int, D: int, I: int) -> bool:
    """Short sequences on the token-major backward schedule keep q/k/v, the attention output and their gradients in bf16
    between the GEMMs and the attention kernels: the qkv projection writes bf16, no cast launches either side of the
    attention.  DCLIP_BF16_ATTN_IO16=0: fp32 attention I/O (casts) instead."""
    return S <= 64 and _tokmajor_wgrads(M, D, I) and os.environ.get("DCLIP_BF16_ATTN_IO16", "1") != "0"


def _attention_mfma16() -> bool:
    """Which kernels serve the bf16-I/O attention of the training student: the bf16 MFMA pair (ops.attention_fwd_bf16_lse /
    attention_bwd_bf16: P and dS rounded to bf16 for the products they feed — the default) or, DCLIP_BF16_ATTN_MFMA=0, the
    fp32-arithmetic kernels with bf16 loads and stores (ops.attention_*_io16)."""
    return os.environ.get("DCLIP_BF16_ATTN_MFMA", "1") != "0"


def layer_fwd_bf16_train(x, p: LayerParams, c: dict, pre: str, B: int, S: int, H: int, causal: bool, eps: float):
    ln1, m1, r1 = ops.layernorm_fwd_bf16(x, p.ln1_w, p.ln1_b, eps, save_stats=True)
    if _attention_io16(x.shape[0], S, x.shape[1], p.fc1_w.shape[0]):
        qkv = ops.gemm_bf16(ln1, _w16(c, pre + "qkv", p.qkv_w), bias=p.qkv_b, out_bf16=True)
        attn16, lse = (ops.attention_fwd_bf16_lse if _attention_mfma16() else ops.attention_fwd_io16)(qkv, B, S, H, causal)
        attn = None
    else:
        qkv = ops.gemm_bf16(ln1, _w16(c, pre + "qkv", p.qkv_w), bias=p.qkv_b)        # fp32 out: the attention core is fp32
        attn, lse = ops.attention_fwd(qkv, B, S, H, causal)
        attn16 = ops.cast_bf16(attn)
    x1 = ops.gemm_bf16(attn16, _w16(c, pre + "out", p.out_w), bias=p.out_b, residual=x)
    ln2, m2, r2 = ops.layernorm_fwd_bf16(x1, p.ln2_w, p.ln2_b, eps, save_stats=True)
    g16, h16 = ops.gemm_bf16(ln2, _w16(c, pre + "fc1", p.fc1_w), bias=p.fc1_b, gelu=True, out_bf16=True, save_preact=True)
    x2 = ops.gemm_bf16(g16, _w16(c, pre + "fc2", p.fc2_w), bias=p.fc2_b, residual=x1)
    return x2, (x, m1, r1, ln1, qkv, attn, attn16, lse, x1, m2, r2, ln2, h16, g16)


_TOKMAJOR_PLAN: Dict[tuple, bool] = {}


def _tokmajor_wgrads(M: int, D: int, I: int) -> bool:
    """Can the four weight gradients of a layer take the token-major form (no transposes)?  One answer per layer so that
    the backward below has two straight schedules; decided once per (M, D, I) (four plan calls + an environment lookup
    per layer per backward otherwise).  DCLIP_BF16_WGRAD_TN=0, read at the first use, forces the transposing schedule."""
    key = (M, D, I)
    hit = _TOKMAJOR_PLAN.get(key)
    if hit is None:
        if os.environ.get("DCLIP_BF16_WGRAD_TN", "1") == "0":
            hit = False
        else:
            lib = _lib.load()
            hit = all(lib.dclip_gemm_bf16_wgrad_tokmajor_plan(m, n, M) > 0 for m, n in ((D, I), (I, D), (D, D), (3 * D, D)))
        _TOKMAJOR_PLAN[key] = hit
    return hit


_WGRAD_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


class _SideWgrads:
    """The four weight-gradient GEMMs of a layer on a SECOND stream beside the data-gradient chain (DCLIP_BF16_WGRAD_STREAM=0
    keeps them on the main stream).  A bf16 ping-pong GEMM takes a whole CU per workgroup, and the student's data-gradient
    GEMMs have 150 tiles for 256 CUs: the weight gradient of the same dY (independent of everything downstream) takes the
    idle CUs — same box, alternating: c3 43.83 -> 43.28 ms, the c2-shaped bf16 step under graph replay 16.78 -> 16.48 ms,
    results bit-identical.  (For the fp32 GEMMs, three workgroups per CU sharing the matrix pipe, the same idea measured
    SLOWER in round 2: there is no idle CU to take.)  Fork: the side stream waits for the main stream (dY is ready); join at the end of the layer.  Operands
    are kept alive until the join (the caching allocator would hand a freed block to the main stream while the side
    stream still reads it); scratch comes from a workspace lane of its own."""

    def __init__(self, dev):
        self.on = os.environ.get("DCLIP_BF16_WGRAD_STREAM", "1") != "0" and dev.type == "cuda"
        self.keep = []
        if self.on:
            self.main = torch.cuda.current_stream(dev)
            self.side = _WGRAD_STREAMS.setdefault(dev.index, torch.cuda.Stream(device=dev))

    def run(self, fn, *operands):
        if not self.on:
            return fn()
        self.keep.extend(operands)
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side), ops.workspace_lane(2):
            return fn()

    def join(self):
        if self.on:
            self.main.wait_stream(self.side)
            self.keep.clear()


def layer_bwd_bf16_tokmajor(dx2, p: LayerParams, c: dict, pre: str, saved, B: int, S: int, H: int, causal: bool,
                            need: Dict[str, bool], alloc=None, dx2_16=None, fc2_b=None, below_fc2_b=None, want_dx16=False):
    """layer_bwd_bf16 with the weight gradients read from the operands as they lie — dW = dY^T X on the token-major form of
    the ping-pong GEMM (ops.gemm_bf16_wgrad_tokmajor): the saved bf16 activations are used as they are, no transposed
    copies are written, and the bf16 copies of the fp32 gradients come out of the kernels that PRODUCE those gradients:
      dx2_16   bf16 copy of the incoming dx2 (LayerNorm1's backward of the layer above wrote it; None: cast here);
      fc2_b    this layer's fc2 bias gradient = column sums of dx2, already reduced by that same LayerNorm backward;
      below_fc2_b  where THIS layer's LayerNorm1 backward leaves the column sums of its dx (the layer below's fc2_b);
      want_dx16    ... and whether it writes the bf16 copy of dx for the layer below.
    Returns (dx, dx16 or None, grads)."""
    x, m1, r1, ln1, qkv, attn, attn16, lse, x1, m2, r2, ln2, h16, g16 = saved
    D = x.shape[1]
    I = g16.shape[1]
    dev = x.device
    gr: Dict[str, torch.Tensor] = {}
    wg = _SideWgrads(dev)
    # ---- fc2
    if dx2_16 is None:
        dx2_16 = ops.cast_bf16(dx2)
    if need.get("fc2_w"):
        o_ = _galloc(alloc, "fc2_w", (D, I), dev)
        gr["fc2_w"] = wg.run(lambda a_=dx2_16, b_=g16: ops.gemm_bf16_wgrad_tokmajor(a_, b_, out=o_), dx2_16, g16)
    if need.get("fc2_b"):
        gr["fc2_b"] = fc2_b if fc2_b is not None else ops.colsum(dx2, out=_galloc(alloc, "fc2_b", (D,), dev))
    dh16 = ops.gemm_bf16(dx2_16, _w16t(c, pre + "fc2", p.fc2_w), k=D, dgelu_of=h16, out_bf16=True)       # [M, I]
    del dx2_16
    # ---- fc1
    if need.get("fc1_w"):
        o1_ = _galloc(alloc, "fc1_w", (I, D), dev)
        gr["fc1_w"] = wg.run(lambda a_=dh16, b_=ln2: ops.gemm_bf16_wgrad_tokmajor(a_, b_, out=o1_), dh16, ln2)
    if need.get("fc1_b"):
        gr["fc1_b"] = ops.colsum_bf16(dh16, out=_galloc(alloc, "fc1_b", (I,), dev))
    dln2 = ops.gemm_bf16(dh16, _w16t(c, pre + "fc1", p.fc1_w), k=dh16.shape[1])                          # [M, D] fp32
    del dh16
    # LayerNorm2 backward: dx1 (+ the skip connection's dx2), its bf16 copy, and out_proj's bias gradient = colsum(dx1)
    want_ln2 = bool(need.get("ln2_w") or need.get("ln2_b"))
    out_b = _galloc(alloc, "out_b", (D,), dev) if need.get("out_b") else None
    dg = _galloc(alloc, "ln2_w", (D,), dev) if want_ln2 else None
    db = _galloc(alloc, "ln2_b", (D,), dev) if want_ln2 else None
    dx1, dg, db, dx1_16 = ops.layernorm_bwd(dln2, x1, p.ln2_w, m2, r2, dresidual=dx2, dgamma=dg, dbeta=db,
                                            need_param_grads=want_ln2, want_bf16=True, dx_colsum=out_b)
    if want_ln2:
        gr["ln2_w"], gr["ln2_b"] = dg, db
    if out_b is not None:
        gr["out_b"] = out_b
    # ---- out_proj
    if need.get("out_w"):
        o2_ = _galloc(alloc, "out_w", (D, D), dev)
        gr["out_w"] = wg.run(lambda a_=dx1_16, b_=attn16: ops.gemm_bf16_wgrad_tokmajor(a_, b_, out=o2_), dx1_16, attn16)
    if qkv.dtype == torch.bfloat16:          # bf16 I/O attention: dO arrives as bf16, dq / dk / dv leave as bf16
        dattn16 = ops.gemm_bf16(dx1_16, _w16t(c, pre + "out", p.out_w), k=D, out_bf16=True)
        del dx1_16
        dqkv16 = (ops.attention_bwd_bf16 if _attention_mfma16() else ops.attention_bwd_io16)(qkv, attn16, dattn16, lse, B, S, H,
                                                                                             causal)
        del dattn16
        if need.get("qkv_b"):
            gr["qkv_b"] = ops.colsum_bf16(dqkv16, out=_galloc(alloc, "qkv_b", (3 * D,), dev))
    else:
        dattn = ops.gemm_bf16(dx1_16, _w16t(c, pre + "out", p.out_w), k=D)                               # [M, D] fp32
        del dx1_16
        dqkv = ops.attention_bwd(qkv, attn, dattn, lse, B, S, H, causal)                                 # fp32 [M, 3D]
        dqkv16 = ops.cast_bf16(dqkv)
        if need.get("qkv_b"):
            gr["qkv_b"] = ops.colsum(dqkv, out=_galloc(alloc, "qkv_b", (3 * D,), dev))
        del dqkv
    # ---- qkv projection
    if need.get("qkv_w"):
        o3_ = _galloc(alloc, "qkv_w", (3 * D, D), dev)
        gr["qkv_w"] = wg.run(lambda a_=dqkv16, b_=ln1: ops.gemm_bf16_wgrad_tokmajor(a_, b_, out=o3_), dqkv16, ln1)
    dln1 = ops.gemm_bf16(dqkv16, _w16t(c, pre + "qkv", p.qkv_w), k=3 * D)
    del dqkv16
    want_ln1 = bool(need.get("ln1_w") or need.get("ln1_b"))
    dg = _galloc(alloc, "ln1_w", (D,), dev) if want_ln1 else None
    db = _galloc(alloc, "ln1_b", (D,), dev) if want_ln1 else None
    res = ops.layernorm_bwd(dln1, x, p.ln1_w, m1, r1, dresidual=dx1, dgamma=dg, dbeta=db, need_param_grads=want_ln1,
                            want_bf16=want_dx16, dx_colsum=below_fc2_b)
    dx, dg, db = res[0], res[1], res[2]
    if want_ln1:
        gr["ln1_w"], gr["ln1_b"] = dg, db
    wg.join()                                  # the weight gradients are final before the caller reports / reduces them
    return dx, (res[3] if want_dx16 else None), gr


def layer_bwd_bf16(dx2, p: LayerParams, c: dict, pre: str, saved, B: int, S: int, H: int, causal: bool, need: Dict[str, bool],
                   alloc=None, dx2_16=None, fc2_b=None, below_fc2_b=None, want_dx16=False):
    """Backward of layer_fwd_bf16_train -> (dx, dx16 or None, grads).  `alloc(field, shape)`: see _galloc — under data
    parallelism every parameter gradient (split-K weight gradients, bias column sums, LayerNorm dγ/dβ) is written straight
    into its bucket slice.  dx2_16 / fc2_b / below_fc2_b / want_dx16: see layer_bwd_bf16_tokmajor (the transposing
    schedule below ignores them, except that it fills below_fc2_b so the caller's bookkeeping holds)."""
    x, m1, r1, ln1, qkv, attn, attn16, lse, x1, m2, r2, ln2, h16, g16 = saved
    M = x.shape[0]
    D = x.shape[1]
    I = g16.shape[1]
    dev = x.device
    if _tokmajor_wgrads(M, D, I):
        return layer_bwd_bf16_tokmajor(dx2, p, c, pre, saved, B, S, H, causal, need, alloc, dx2_16, fc2_b, below_fc2_b,
                                       want_dx16)
    gr: Dict[str, torch.Tensor] = {}
    # ---- fc2
    dx2T, dx2_16 = ops.transpose_bf16(dx2, want_copy=True)
    if need.get("fc2_w"):
        gr["fc2_w"] = ops.gemm_bf16_wgrad(dx2T, ops.transpose_bf16(g16), M, out=_galloc(alloc, "fc2_w", (D, I), dev))
    if need.get("fc2_b"):
        gr["fc2_b"] = fc2_b if fc2_b is not None else ops.colsum(dx2, out=_galloc(alloc, "fc2_b", (D,), dev))
    dh16 = ops.gemm_bf16(dx2_16, _w16t(c, pre + "fc2", p.fc2_w), k=D, dgelu_of=h16, out_bf16=True)       # [M, I]
    del dx2T, dx2_16
    # ---- fc1
    dhT = ops.transpose_bf16(dh16)
    if need.get("fc1_w"):
        gr["fc1_w"] = ops.gemm_bf16_wgrad(dhT, ops.transpose_bf16(ln2), M, out=_galloc(alloc, "fc1_w", (I, D), dev))
    if need.get("fc1_b"):
        gr["fc1_b"] = ops.rowsum_bf16(dhT, M, out=_galloc(alloc, "fc1_b", (I,), dev))
    dln2 = ops.gemm_bf16(dh16, _w16t(c, pre + "fc1", p.fc1_w), k=dh16.shape[1])                          # [M, D] fp32
    del dh16, dhT
    dx1 = _ln_bwd(dln2, x1, p.ln2_w, m2, r2, dx2, bool(need.get("ln2_w") or need.get("ln2_b")), gr, "ln2_w", "ln2_b", alloc)
    # ---- out_proj
    dx1T, dx1_16 = ops.transpose_bf16(dx1, want_copy=True)
    if need.get("out_w"):
        gr["out_w"] = ops.gemm_bf16_wgrad(dx1T, ops.transpose_bf16(attn16), M, out=_galloc(alloc, "out_w", (D, D), dev))
    if need.get("out_b"):
        gr["out_b"] = ops.colsum(dx1, out=_galloc(alloc, "out_b", (D,), dev))
    dattn = ops.gemm_bf16(dx1_16, _w16t(c, pre + "out", p.out_w), k=D)                                   # [M, D] fp32
    del dx1T, dx1_16
    dqkv = ops.attention_bwd(qkv, attn, dattn, lse, B, S, H, causal)                                     # fp32 [M, 3D]
    # ---- qkv projection
    dqkvT, dqkv16 = ops.transpose_bf16(dqkv, want_copy=True)
    if need.get("qkv_w"):
        gr["qkv_w"] = ops.gemm_bf16_wgrad(dqkvT, ops.transpose_bf16(ln1), M, out=_galloc(alloc, "qkv_w", (3 * D, D), dev))
    if need.get("qkv_b"):
        gr["qkv_b"] = ops.colsum(dqkv, out=_galloc(alloc, "qkv_b", (3 * D,), dev))
    dln1 = ops.gemm_bf16(dqkv16, _w16t(c, pre + "qkv", p.qkv_w), k=3 * D)
    del dqkv, dqkvT, dqkv16
    dx = _ln_bwd(dln1, x, p.ln1_w, m1, r1, dx1, bool(need.get("ln1_w") or need.get("ln1_b")), gr, "ln1_w", "ln1_b", alloc)
    if below_fc2_b is not None:
        ops.colsum(dx, out=below_fc2_b)
    return dx, None, gr


_PATCH_BF16_PLAN: Dict[tuple, bool] = {}


def _patch_embed_bf16(v, rows: int) -> bool:
    """Does the patch embedding of the TRAINING tower (the convolution as a GEMM: [rows = images x patches][3 p p] x
    [D][3 p p]^T, and its weight gradient) run on the bf16 MFMAs like the encoder layers?  Yes when the one-pass bf16
    im2col and the token-major weight-gradient kernel both apply to the shape (`DCLIP_BF16_PATCH=0`: keep it fp32, as
    rounds 2-3 had it: 0.44 + 0.43 ms of fp32 GEMM per step at 256 images against 0.05 + 0.07)."""
    key = (v.hidden_size, v.patch_dim, v.patch_size, rows)
    hit = _PATCH_BF16_PLAN.get(key)
    if hit is None:
        hit = (os.environ.get("DCLIP_BF16_PATCH", "1") != "0" and v.patch_size % 4 == 0 and v.patch_dim % 64 == 0
               and _lib.load().dclip_gemm_bf16_wgrad_tokmajor_plan(v.hidden_size, v.patch_dim, rows) > 0)
        _PATCH_BF16_PLAN[key] = hit
    return hit


def vision_fwd_bf16_train(p: VisionParams, pixel_values: torch.Tensor, cfg, cache: dict):
    """get_image_features with gradients, bf16 GEMM inputs in the patch embedding and the encoder layers."""
    v = cfg
    B = pixel_values.shape[0]
    S, D, H = v.seq_len, v.hidden_size, v.num_attention_heads
    if _patch_embed_bf16(v, B * (S - 1)):
        cols = ops.im2col_bf16(pixel_values, v.patch_size)                    # one pass: gather + round; kept for the wgrad
        patch = ops.gemm_bf16(cols, _w16(cache, "vpatch", p.patch_w), k=v.patch_dim)
    else:
        cols = ops.im2col(pixel_values, v.patch_size)
        patch = ops.gemm(cols, p.patch_w.view(D, -1), ops.LAYOUT_NT)
    emb = ops.vision_assemble_fwd(patch, p.class_embedding, p.pos, B, S, D)
    del patch
    x, m0, r0 = ops.layernorm_fwd(emb, p.pre_w, p.pre_b, v.layer_norm_eps, save_stats=True)
    refresh_train_weights(cache, p.layers, "v")          # W and W^T of all layers, one launch per optimizer step
    saved_layers = []
    for li, lp in enumerate(p.layers):
        x, sv = layer_fwd_bf16_train(x, lp, cache, f"v{li}.", B, S, H, False, v.layer_norm_eps)
        saved_layers.append(sv)
    cls_tok = ops.gather_rows(x, None, B, S, D)
    pooled, mp, rp = ops.layernorm_fwd(cls_tok, p.post_w, p.post_b, v.layer_norm_eps, save_stats=True)
    out = ops.gemm(pooled, p.proj_w, ops.LAYOUT_NT)
    return out, (cols, emb, m0, r0, saved_layers, cls_tok, mp, rp, pooled)


def vision_bwd_bf16(p: VisionParams, saved, d_out: torch.Tensor, cfg, need: List[bool], cache: dict, on_ready=None, alloc=None):
    """Backward of vision_fwd_bf16_train; same contract as vision_bwd (`on_ready` per gradient group, `alloc` naming the
    tensor each parameter gradient is written into)."""
    v = cfg
    cols, emb, m0, r0, saved_layers, cls_tok, mp, rp, pooled = saved
    B = cls_tok.shape[0]
    S, D, H = v.seq_len, v.hidden_size, v.num_attention_heads
    names = p.names()
    needd = dict(zip(names, need))
    grads: Dict[str, Optional[torch.Tensor]] = {n: None for n in names}
    dev = d_out.device
    if needd["proj_w"]:
        grads["proj_w"] = ops.gemm(d_out, pooled, ops.LAYOUT_TN, out=_galloc(alloc, "proj_w", tuple(p.proj_w.shape), dev))
    dpooled = ops.gemm(d_out, p.proj_w, ops.LAYOUT_NN)
    dcls = _ln_bwd(dpooled, cls_tok, p.post_w, mp, rp, None, bool(needd["post_w"] or needd["post_b"]), grads, "post_w", "post_b",
                   alloc)
    if on_ready is not None:
        on_ready({n: grads[n] for n in VisionParams.TAIL if grads[n] is not None})
    lowest = None
    for n in names:
        if needd[n] and n not in VisionParams.TAIL:
            li = -1 if n in VisionParams.HEAD else int(n.split(".")[1])
            lowest = li if lowest is None else min(lowest, li)
    if lowest is None:
        return [grads[n] for n in names]
    dx = ops.scatter_rows(dcls, None, B, S, D)
    dx16, fc2_b = None, None
    bottom = max(lowest, 0)
    for i in range(len(p.layers) - 1, bottom - 1, -1):
        lneed = {f: needd[f"layers.{i}.{f}"] for f in LayerParams.FIELDS}
        lalloc = None if alloc is None else (lambda f, shape, i=i: alloc(f"layers.{i}.{f}", shape))
        # this layer's LayerNorm1 backward also leaves, for the layer below, the bf16 copy of dx and its column sums (= that
        # layer's fc2 bias gradient, written where `alloc` puts it)
        below = None
        if i > bottom and needd[f"layers.{i - 1}.fc2_b"]:
            below = _galloc(alloc, f"layers.{i - 1}.fc2_b", (D,), dev)
        dx, dx16, gr = layer_bwd_bf16(dx, p.layers[i], cache, f"v{i}.", saved_layers[i], B, S, H, False, lneed, lalloc,
                                      dx2_16=dx16, fc2_b=fc2_b, below_fc2_b=below, want_dx16=i > bottom)
        fc2_b = below
        saved_layers[i] = None
        for f, t in gr.items():
            grads[f"layers.{i}.{f}"] = t
        if on_ready is not None:
            on_ready({f"layers.{i}.{f}": t for f, t in gr.items()})
    if lowest < 0:
        demb = _ln_bwd(dx, emb, p.pre_w, m0, r0, None, bool(needd["pre_w"] or needd["pre_b"]), grads, "pre_w", "pre_b", alloc)
        if needd["pos"] or needd["class_embedding"]:
            dpos = ops.colsum(demb.view(B, S * D), out=(_galloc(alloc, "pos", (S * D,), dev) if needd["pos"] else None))
            if needd["pos"]:
                grads["pos"] = dpos.view(S, D)
            if needd["class_embedding"]:
                ce = _galloc(alloc, "class_embedding", (D,), dev)
                ce.copy_(dpos[:D])
                grads["class_embedding"] = ce
        if needd["patch_w"]:
            dpatch = ops.vision_assemble_bwd(demb, B, S, D)
            pw = _galloc(alloc, "patch_w", (D, p.patch_w.numel() // D), dev)
            if cols.dtype == torch.bfloat16:          # token-major split-K weight gradient from the saved bf16 columns
                got = ops.gemm_bf16_wgrad_tokmajor(ops.cast_bf16(dpatch), cols, out=pw)
                if got is None:
                    raise RuntimeError("patch-embedding weight gradient: the token-major bf16 kernel refused a shape its plan accepted")
                grads["patch_w"] = got.view_as(p.patch_w)
            else:
                grads["patch_w"] = ops.gemm(dpatch, cols, ops.LAYOUT_TN, out=pw).view_as(p.patch_w)
        if on_ready is not None:
            on_ready({n: grads[n] for n in VisionParams.HEAD if grads[n] is not None})
    return [grads[n] for n in names]


# --------------------------------------------------------------------------------------------- text tower

@dataclass
class TextParams:
    tok: torch.Tensor            # [vocab, D]
    pos: torch.Tensor            # [Tmax, D]
    layers: List[LayerParams]
    final_w: torch.Tensor
    final_b: torch.Tensor
    proj_w: torch.Tensor         # text_projection [P, D]

    HEAD = ("tok", "pos")
    TAIL = ("final_w", "final_b", "proj_w")

    def tensors(self):
        out = [self.tok, self.pos]
        for l in self.layers:
            out += l.tensors()
        return out + [self.final_w, self.final_b, self.proj_w]

    @classmethod
    def from_tensors(cls, ts, n_layers):
        ts = list(ts)
        layers = [LayerParams(*ts[2 + 12 * i: 2 + 12 * (i + 1)]) for i in range(n_layers)]
        return cls(ts[0], ts[1], layers, *ts[2 + 12 * n_layers:])

    def names(self):
        out = list(self.HEAD)
        for i in range(len(self.layers)):
            out += [f"layers.{i}.{f}" for f in LayerParams.FIELDS]
        return out + list(self.TAIL)


def text_encoder_fwd(p: TextParams, input_ids: torch.Tensor, cfg, save: bool, hidden_out: Optional[list] = None):
    """Embeddings + causal stack; returns the PRE-final-LN hidden states [B*T, D]."""
    t = cfg
    B, T = input_ids.shape
    D, H = t.hidden_size, t.num_attention_heads
    x = ops.text_embed_fwd(input_ids, p.tok, p.pos)
    if hidden_out is not None:
        hidden_out.append(x)
    saved_layers = []
    for lp in p.layers:
        x, sv = layer_fwd(x, lp, B, T, H, True, t.layer_norm_eps, save)
        saved_layers.append(sv)
        if hidden_out is not None:
            hidden_out.append(x)
    return x, saved_layers


def text_fwd_frozen(p: TextParams, input_ids: torch.Tensor, cfg):
    """get_text_features without gradients: like text_fwd, but the LAST layer is evaluated only where it is read —
    everything after its attention on the B first-EOS rows, its attention for that one query row per caption
    (keys 0..eos, causal).  Exact; 9/12 of that layer's GEMM work is never launched."""
    t = cfg
    B, T = input_ids.shape
    D, H = t.hidden_size, t.num_attention_heads
    eos = ops.first_eos(input_ids, t.eos_token_id)
    x = ops.text_embed_fwd(input_ids, p.tok, p.pos)
    for lp in p.layers[:-1]:
        x, _ = layer_fwd(x, lp, B, T, H, True, t.layer_norm_eps, False)
    lp = p.layers[-1]
    ln1, _, _ = ops.layernorm_fwd(x, lp.ln1_w, lp.ln1_b, t.layer_norm_eps, save_stats=False)
    qkv = ops.gemm(ln1, lp.qkv_w, ops.LAYOUT_NT, bias=lp.qkv_b)
    attn = ops.attention_row_fwd(qkv, eos, B, T, H)
    x1 = ops.gemm(attn, lp.out_w, ops.LAYOUT_NT, bias=lp.out_b, residual=ops.gather_rows(x, eos, B, T, D))
    ln2, _, _ = ops.layernorm_fwd(x1, lp.ln2_w, lp.ln2_b, t.layer_norm_eps, save_stats=False)
    g = ops.gemm(ln2, lp.fc1_w, ops.LAYOUT_NT, bias=lp.fc1_b, epilogue=ops.EPI_GELU)
    rows = ops.gemm(g, lp.fc2_w, ops.LAYOUT_NT, bias=lp.fc2_b, residual=x1)
    pooled, _, _ = ops.layernorm_fwd(rows, p.final_w, p.final_b, t.layer_norm_eps, save_stats=False)
    return ops.gemm(pooled, p.proj_w, ops.LAYOUT_NT)


def text_fwd(p: TextParams, input_ids: torch.Tensor, cfg, save: bool, hidden_out: Optional[list] = None):
    """get_text_features: ids [B,T] -> [B,P]  (hf:modeling_clip.py:541-586, :705-713).  LayerNorm is row-wise, so
    the first-EOS rows are gathered BEFORE final_layer_norm: only B rows are normalised and projected."""
    t = cfg
    B, T = input_ids.shape
    D = t.hidden_size
    x, saved_layers = text_encoder_fwd(p, input_ids, cfg, save, hidden_out)
    eos = ops.first_eos(input_ids, t.eos_token_id)
    rows = ops.gather_rows(x, eos, B, T, D)
    pooled, mp, rp = ops.layernorm_fwd(rows, p.final_w, p.final_b, t.layer_norm_eps, save_stats=save)
    out = ops.gemm(pooled, p.proj_w, ops.LAYOUT_NT)
    saved = (input_ids, saved_layers, eos, rows, mp, rp, pooled) if save else None
    return out, saved


def text_bwd(p: TextParams, saved, d_out: torch.Tensor, cfg, need: List[bool]):
    t = cfg
    input_ids, saved_layers, eos, rows, mp, rp, pooled = saved
    B, T = input_ids.shape
    D, H = t.hidden_size, t.num_attention_heads
    names = p.names()
    needd = dict(zip(names, need))
    grads: Dict[str, Optional[torch.Tensor]] = {n: None for n in names}
    if needd["proj_w"]:
        grads["proj_w"] = ops.gemm(d_out, pooled, ops.LAYOUT_TN)
    dpooled = ops.gemm(d_out, p.proj_w, ops.LAYOUT_NN)
    want = needd["final_w"] or needd["final_b"]
    drows, dg, db = ops.layernorm_bwd(dpooled, rows, p.final_w, mp, rp, need_param_grads=want)
    if want:
        grads["final_w"], grads["final_b"] = dg, db
    lowest = None
    for n in names:
        if needd[n] and n not in TextParams.TAIL:
            li = -1 if n in TextParams.HEAD else int(n.split(".")[1])
            lowest = li if lowest is None else min(lowest, li)
    if lowest is None:
        return [grads[n] for n in names]
    dx = ops.scatter_rows(drows, eos, B, T, D)
    for i in range(len(p.layers) - 1, max(lowest, 0) - 1, -1):
        lneed = {f: needd[f"layers.{i}.{f}"] for f in LayerParams.FIELDS}
        dx, gr = layer_bwd(dx, p.layers[i], saved_layers[i], B, T, H, True, lneed)
        saved_layers[i] = None
        for f, tn in gr.items():
            grads[f"layers.{i}.{f}"] = tn
    if lowest < 0:
        if needd["pos"]:
            dpos = torch.zeros_like(p.pos)
            dpos[:T] = ops.colsum(dx.view(B, T * D)).view(T, D)
            grads["pos"] = dpos
        if needd["tok"]:
            dtok = torch.zeros_like(p.tok)
            grads["tok"] = ops.text_embed_bwd(input_ids, dx, dtok)
    return [grads[n] for n in names]


def _text_stack_bf16(p: TextParams, input_ids: torch.Tensor, cfg, cache: dict, n_layers: int):
    t = cfg
    B, T = input_ids.shape
    x = ops.text_embed_fwd(input_ids, p.tok, p.pos)
    for li, lp in enumerate(p.layers[:n_layers]):
        x = _layer_fwd_bf16(x, lp, cache, f"t{li}.", B, T, t.num_attention_heads, True, t.layer_norm_eps)
    return x


def text_fwd_frozen_bf16(p: TextParams, input_ids: torch.Tensor, cfg, cache: dict) -> torch.Tensor:
    """text_fwd_frozen with bf16 GEMM inputs (see the bf16 note above vision_fwd_bf16)."""
    t = cfg
    B, T = input_ids.shape
    D, H = t.hidden_size, t.num_attention_heads
    eos = ops.first_eos(input_ids, t.eos_token_id)
    x = _text_stack_bf16(p, input_ids, cfg, cache, len(p.layers) - 1)
    lp, pre = p.layers[-1], f"t{len(p.layers) - 1}."
    ln1 = ops.layernorm_fwd_bf16(x, lp.ln1_w, lp.ln1_b, t.layer_norm_eps)
    if T <= 512 and os.environ.get("DCLIP_BF16_ROW_ATTN", "1") != "0":
        qkv = ops.gemm_bf16(ln1, _w16(cache, pre + "qkv", lp.qkv_w), bias=lp.qkv_b, out_bf16=True)
        attn16 = ops.attention_row_fwd_bf16(qkv, eos, B, T, H)
    else:
        qkv = ops.gemm_bf16(ln1, _w16(cache, pre + "qkv", lp.qkv_w), bias=lp.qkv_b)
        attn16 = ops.cast_bf16(ops.attention_row_fwd(qkv, eos, B, T, H))
    x1 = ops.gemm_bf16(attn16, _w16(cache, pre + "out", lp.out_w), bias=lp.out_b,
                       residual=ops.gather_rows(x, eos, B, T, D))
    ln2 = ops.layernorm_fwd_bf16(x1, lp.ln2_w, lp.ln2_b, t.layer_norm_eps)
    g = ops.gemm_bf16(ln2, _w16(cache, pre + "fc1", lp.fc1_w), bias=lp.fc1_b, gelu=True, out_bf16=True)
    rows = ops.gemm_bf16(g, _w16(cache, pre + "fc2", lp.fc2_w), bias=lp.fc2_b, residual=x1)
    pooled, _, _ = ops.layernorm_fwd(rows, p.final_w, p.final_b, t.layer_norm_eps, save_stats=False)
    return ops.gemm(pooled, p.proj_w, ops.LAYOUT_NT)


def text_token_level_bf16(p: TextParams, input_ids: torch.Tensor, cfg, cache: dict):
    """text_token_level with bf16 GEMM inputs; the token projection ([B*T,D] x [P,D]) runs in bf16 too."""
    t = cfg
    B, T = input_ids.shape
    x = _text_stack_bf16(p, input_ids, cfg, cache, len(p.layers))
    ln = ops.layernorm_fwd_bf16(x, p.final_w, p.final_b, t.layer_norm_eps)
    tokens = ops.gemm_bf16(ln, _w16(cache, "tproj", p.proj_w))
    eos = ops.first_eos(input_ids, t.eos_token_id)
    sentence = ops.gather_rows(tokens, eos, B, T, tokens.shape[1])
    return sentence, tokens, eos


def text_token_level(p: TextParams, input_ids: torch.Tensor, cfg):
    """Frozen teacher text pass giving BOTH outputs of one forward (the reference runs the tower twice per caption,
    training/patch_text_aggregation.py:534 and training/CLIP_image_distillation.py:607):
      sentence [B,P]   = text_projection(final_LN(h)[first EOS])                 (text_tokenizer.py:193,:216)
      tokens   [B*T,P] = text_projection(final_LN(h)) for every position          (text_tokenizer.py:202-207)
      eos [B] int32    = first-EOS index; word tokens of caption b are rows 1 .. eos[b]-1."""
    t = cfg
    B, T = input_ids.shape
    D = t.hidden_size
    x, _ = text_encoder_fwd(p, input_ids, cfg, save=False)
    ln, _, _ = ops.layernorm_fwd(x, p.final_w, p.final_b, t.layer_norm_eps, save_stats=False)
    tokens = ops.gemm(ln, p.proj_w, ops.LAYOUT_NT)
    eos = ops.first_eos(input_ids, t.eos_token_id)
    sentence = ops.gather_rows(tokens, eos, B, T, tokens.shape[1])
    return sentence, tokens, eos
