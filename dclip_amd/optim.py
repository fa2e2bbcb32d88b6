"""`FusedAdamW`: torch.optim.AdamW semantics (the reference's optimizer, training/CLIP_image_distillation.py:680)
with the update and the Trainer's global-norm clipping (gradient_clip_val=0.5,
training/CLIP_image_distill_training.py:41) in the HIP library.  Subclasses torch.optim.Optimizer so that
LR schedulers, state_dict() and the Lightning-style checkpoint layout keep working."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 max_grad_norm: Optional[float] = None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.max_grad_norm = max_grad_norm
        self._partials = None
        self._coef = None
        self.last_grad_norm = None

    def _clip_coef(self, stream):
        """Device-side clip coefficient over every parameter that has a gradient (no host sync)."""
        lib = _lib.load()
        ps = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
        dev = ps[0].device
        counts = [lib.dclip_sumsq_blocks(p.grad.numel()) for p in ps]
        total = sum(counts)
        if self._partials is None or self._partials.numel() < total:
            self._partials = torch.empty(total, dtype=torch.float32, device=dev)
            self._coef = torch.empty(2, dtype=torch.float32, device=dev)
        o = 0
        for p, c in zip(ps, counts):
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            _lib.check(lib.dclip_sumsq_f32(g.data_ptr(), g.numel(), self._partials.data_ptr() + 4 * o, stream), "sumsq")
            o += c
        _lib.check(lib.dclip_clip_coef(self._partials.data_ptr(), total, float(self.max_grad_norm),
                                       self._coef.data_ptr(), self._coef.data_ptr() + 4, stream), "clip_coef")
        self.last_grad_norm = self._coef[1]
        return self._coef

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        coef = self._clip_coef(stream) if self.max_grad_norm else None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise ValueError("FusedAdamW: contiguous float32 CUDA parameters only")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                _lib.check(lib.dclip_adamw_f32(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                               st["exp_avg_sq"].data_ptr(), p.numel(), float(group["lr"]), float(b1),
                                               float(b2), float(group["eps"]), float(group["weight_decay"]),
                                               int(st["step"]), None if coef is None else coef.data_ptr(), stream),
                           "adamw")
        return loss
