"""torch.autograd glue: one Function per tower / loss, each running the hand-written schedules in
engine.py on the HIP kernels.  Autograd sees a graph of ~5 nodes per step; everything inside a
node is an explicit launch sequence on the current HIP stream."""
from __future__ import annotations

from typing import List, Optional

import torch

from . import engine, ops


# Data-parallel hook: called from inside a tower's backward with [(Parameter, gradient), ...] as soon as those gradients
# are final, so their all-reduce overlaps the rest of the backward (dist.GradSync.on_grads_ready).  None = no hook.
_GRAD_READY_HOOK = None


def set_grad_ready_hook(fn):
    """`fn([(Parameter, gradient), ...]) -> bool`: True = the hook OWNS these gradients from here on (it will set
    `.grad` itself, e.g. to a slice of a reduced bucket) and the tower's backward returns None for them to autograd."""
    global _GRAD_READY_HOOK
    _GRAD_READY_HOOK = fn


# Where the hand-written backward writes a parameter's gradient: `fn(Parameter, shape) -> tensor or None`
# (dist.GradSync.grad_buffer: the parameter's slice of its persistent all-reduce bucket).  None = fresh tensors.
_GRAD_ALLOC = None


def set_grad_alloc(fn):
    global _GRAD_ALLOC
    _GRAD_ALLOC = fn


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


class VisionTowerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pixel_values, cfg, n_layers, *params):
        p = engine.VisionParams.from_tensors([_c(t.detach()) for t in params], n_layers)
        save = any(ctx.needs_input_grad[3:])
        out, saved = engine.vision_fwd(p, _c(pixel_values.detach()), cfg, save)
        ctx.p, ctx.saved, ctx.cfg = p, saved, cfg
        ctx.param_refs = params if save else None          # the nn.Parameters, for the data-parallel hook
        return out

    @staticmethod
    def backward(ctx, d_out):
        if ctx.saved is None:
            raise RuntimeError("VisionTowerFn.backward called twice, or forward ran without grad")
        on_ready = alloc = None
        owned = set()
        if _GRAD_READY_HOOK is not None or _GRAD_ALLOC is not None:
            by_name = dict(zip(ctx.p.names(), ctx.param_refs))
        if _GRAD_READY_HOOK is not None:
            hook = _GRAD_READY_HOOK

            def on_ready(named):
                if hook([(by_name[n], g) for n, g in named.items()]):
                    owned.update(named)
        if _GRAD_ALLOC is not None:
            galloc = _GRAD_ALLOC
            alloc = lambda name, shape: galloc(by_name[name], shape)      # noqa: E731
        grads = engine.vision_bwd(ctx.p, ctx.saved, _c(d_out), ctx.cfg, list(ctx.needs_input_grad[3:]), on_ready, alloc)
        ctx.saved = None
        if owned:           # handed to the data-parallel reducer: autograd neither accumulates nor clones them
            grads = [None if n in owned else g for n, g in zip(ctx.p.names(), grads)]
        return (None, None, None, *grads)


class VisionTowerBf16Fn(torch.autograd.Function):
    """VisionTowerFn with bf16 GEMM inputs in forward, dgrad and wgrad (engine.vision_fwd_bf16_train); `cache` holds the
    bf16 weight copies (HipCLIPModel._bf16_cache: rebuilt when a parameter version changes)."""

    @staticmethod
    def forward(ctx, pixel_values, cfg, n_layers, cache, *params):
        p = engine.VisionParams.from_tensors([_c(t.detach()) for t in params], n_layers)
        out, saved = engine.vision_fwd_bf16_train(p, _c(pixel_values.detach()), cfg, cache)
        ctx.p, ctx.saved, ctx.cfg, ctx.cache = p, saved, cfg, cache
        ctx.param_refs = params
        return out

    @staticmethod
    def backward(ctx, d_out):
        if ctx.saved is None:
            raise RuntimeError("VisionTowerBf16Fn.backward called twice")
        on_ready = alloc = None
        owned = set()
        if _GRAD_READY_HOOK is not None or _GRAD_ALLOC is not None:
            by_name = dict(zip(ctx.p.names(), ctx.param_refs))
        if _GRAD_READY_HOOK is not None:
            hook = _GRAD_READY_HOOK

            def on_ready(named):
                if hook([(by_name[n], g) for n, g in named.items()]):
                    owned.update(named)
        if _GRAD_ALLOC is not None:       # bf16 wgrads (split-K reduce), bias sums, LayerNorm dγ/dβ land in the bucket slices
            galloc = _GRAD_ALLOC
            alloc = lambda name, shape: galloc(by_name[name], shape)      # noqa: E731
        grads = engine.vision_bwd_bf16(ctx.p, ctx.saved, _c(d_out), ctx.cfg, list(ctx.needs_input_grad[4:]), ctx.cache,
                                       on_ready, alloc)
        ctx.saved = None
        if owned:
            grads = [None if n in owned else g for n, g in zip(ctx.p.names(), grads)]
        return (None, None, None, None, *grads)


class TextTowerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input_ids, cfg, n_layers, *params):
        p = engine.TextParams.from_tensors([_c(t.detach()) for t in params], n_layers)
        save = any(ctx.needs_input_grad[3:])
        if not save and len(p.layers) > 0:
            ctx.saved = None
            return engine.text_fwd_frozen(p, _c(input_ids), cfg)
        out, saved = engine.text_fwd(p, _c(input_ids), cfg, save)
        ctx.p, ctx.saved, ctx.cfg = p, saved, cfg
        return out

    @staticmethod
    def backward(ctx, d_out):
        if ctx.saved is None:
            raise RuntimeError("TextTowerFn.backward called twice, or forward ran without grad")
        grads = engine.text_bwd(ctx.p, ctx.saved, _c(d_out), ctx.cfg, list(ctx.needs_input_grad[3:]))
        ctx.saved = None
        return (None, None, None, *grads)


class CosineDistillationLossFn(torch.autograd.Function):
    """mean_b (1 - cos(student_b, teacher_b)) — training/CLIP_image_distillation.py:564-576.
    No gradient flows to the teacher (it is computed under no_grad in the reference step, :597-600)."""

    @staticmethod
    def forward(ctx, student, teacher):
        s, t = _c(student.detach().float()), _c(teacher.detach().float())
        loss_sum, cos = ops.cosine_loss_fwd(s, t)
        ctx.save_for_backward(s, t, cos)
        return loss_sum / s.shape[0]

    @staticmethod
    def backward(ctx, g):
        s, t, cos = ctx.saved_tensors
        # coef = upstream / B; upstream is a device scalar: multiply afterwards to avoid a host sync
        ds = ops.cosine_loss_bwd(s, t, cos, 1.0 / s.shape[0])
        return ds * g, None


class ContrastiveLossFn(torch.autograd.Function):
    """Symmetric InfoNCE over the in-batch (or all-gathered) similarity matrix with the constant temperature 0.05
    (training/CLIP_image_distillation.py:532-562).

    Single process: rows = all images, columns = all texts.  Data parallel (`group` given): every rank
    all-gathers the L2-normalised embeddings (global negatives), computes the row-LSE of its LOCAL images against
    all texts and of its LOCAL texts against all images, all-gathers those two LSE vectors, and returns
    its share  sum_local(lse - positive) / (2 * B_global)  of the loss; summing the returned value over ranks
    gives the single-process loss.  The backward needs no collective: with both LSE vectors known everywhere,
    each rank forms d/d(own rows) directly."""

    @staticmethod
    def forward(ctx, image_emb, text_emb, temperature, group):
        img, txt = _c(image_emb.detach().float()), _c(text_emb.detach().float())
        Bl = img.shape[0]
        inv_t = 1.0 / float(temperature)
        ihat, iinv = ops.normalize_rows_fwd(img)
        that, tinv = ops.normalize_rows_fwd(txt)
        if group is not None:
            import torch.distributed as dist
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            both = torch.stack([ihat, that])                               # [2, Bl, P]: one collective for both
            gathered = torch.empty((world * 2,) + tuple(both.shape[1:]), dtype=both.dtype, device=both.device)
            dist.all_gather_into_tensor(gathered, both, group=group)      # rank-major concatenation along dim 0
            gathered = gathered.view((world, 2) + tuple(both.shape[1:]))
            i_all = gathered[:, 0].reshape(world * Bl, -1).contiguous()
            t_all = gathered[:, 1].reshape(world * Bl, -1).contiguous()
        else:
            world, rank = 1, 0
            i_all, t_all = ihat, that
        Bg, off = world * Bl, rank * Bl
        lse_i, diag = ops.contrastive_lse(ihat, t_all, off, inv_t)       # image rows vs all texts
        lse_t, _ = ops.contrastive_lse(that, i_all, off, inv_t)          # text rows vs all images
        if group is not None:
            both = torch.stack([lse_i, lse_t])
            g2 = torch.empty((world * 2, Bl), dtype=both.dtype, device=both.device)
            dist.all_gather_into_tensor(g2, both, group=group)
            g2 = g2.view(world, 2, Bl)
            lse_i_all = g2[:, 0].reshape(-1).contiguous()
            lse_t_all = g2[:, 1].reshape(-1).contiguous()
        else:
            lse_i_all, lse_t_all = lse_i, lse_t
        coef = 1.0 / (2.0 * Bg)
        loss = ops.sub_reduce(lse_i, diag, coef)
        ops.sub_reduce(lse_t, diag, coef, out=loss, accumulate=True)
        ctx.save_for_backward(ihat, iinv, that, tinv, i_all, t_all, lse_i, lse_t, lse_i_all, lse_t_all)
        ctx.off, ctx.inv_t, ctx.Bg = off, inv_t, Bg
        return loss

    @staticmethod
    def backward(ctx, g):
        ihat, iinv, that, tinv, i_all, t_all, lse_i, lse_t, lse_i_all, lse_t_all = ctx.saved_tensors
        coef = ctx.inv_t / (2.0 * ctx.Bg)
        # d/d ihat_l: rows of Z use lse_i (own), columns use lse_t of every text
        d_ihat = ops.contrastive_grad(ihat, t_all, lse_i, lse_t_all, ctx.off, ctx.inv_t, coef)
        d_that = ops.contrastive_grad(that, i_all, lse_t, lse_i_all, ctx.off, ctx.inv_t, coef)
        d_img = ops.normalize_rows_bwd(d_ihat, ihat, iinv)
        d_txt = ops.normalize_rows_bwd(d_that, that, tinv)
        return d_img * g, d_txt * g, None, None


def cosine_distillation_loss(student: torch.Tensor, teacher: torch.Tensor) -> torch.Tensor:
    return CosineDistillationLossFn.apply(student, teacher)


def contrastive_loss(image_emb: torch.Tensor, text_emb: torch.Tensor, temperature: float = 0.05,
                     group=None) -> torch.Tensor:
    return ContrastiveLossFn.apply(image_emb, text_emb, temperature, group)
