"""`FusedAdamW`: torch.optim.AdamW semantics (the reference's optimizer, training/CLIP_image_distillation.py:680)
with the update and the Trainer's global-norm clipping (gradient_clip_val=0.5,
training/CLIP_image_distill_training.py:41) in the HIP library.  Subclasses torch.optim.Optimizer so that
LR schedulers, state_dict() and the Lightning-style checkpoint layout keep working.

Per step and per parameter group: ONE multi-tensor sum-of-squares launch, one clip-coefficient launch (the
coefficient stays on the device — no host sync) and ONE multi-tensor AdamW launch.  The table of tensor records is
rebuilt on the host each step (gradients are fresh tensors every backward) and uploaded with one small copy."""
from __future__ import annotations

import struct
from typing import Optional

import torch

from . import _lib


class FusedAdamW(torch.optim.Optimizer):
    _entry = "dclip_mt_adamw_f32"

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 max_grad_norm: Optional[float] = None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.max_grad_norm = max_grad_norm
        self._partials = None
        self._coef = None
        self._table = {}
        self.last_grad_norm = None

    def _records(self, group, gi):
        """(device table, ntensors, total_chunks, grads kept alive) for one parameter group."""
        lib = _lib.load()
        chunk = lib.dclip_mt_chunk_elems()
        assert lib.dclip_mt_record_bytes() == 48
        recs, keep, c0 = [], [], 0
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
            keep.append(g)
            n = p.numel()
            recs.append(struct.pack("<QQQQQii", p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                    st["exp_avg_sq"].data_ptr(), n, int(st["step"]), c0))
            c0 += (n + chunk - 1) // chunk
        if not recs:
            return None
        blob = b"".join(recs)
        # The upload is asynchronous (pinned -> device on the compute stream) and the host runs ahead of the GPU, so
        # the pinned staging buffer of step n must not be rewritten for step n+1 before its copy has executed (the
        # kernel of step n would read step n+1's bias-correction counters): a small ring of staging buffers, each
        # guarded by the event recorded behind its last copy.
        ring = self._table.get(gi)
        if ring is None or ring["dev"].numel() < len(blob):
            ring = {"dev": torch.empty(len(blob), dtype=torch.uint8, device=keep[0].device),
                    "host": [torch.empty(len(blob), dtype=torch.uint8).pin_memory() for _ in range(4)],
                    "done": [None] * 4, "next": 0}
            self._table[gi] = ring
        i = ring["next"]
        ring["next"] = (i + 1) % 4
        if ring["done"][i] is not None:
            ring["done"][i].synchronize()            # four steps ago: already finished unless the host is far ahead
        ring["host"][i][:len(blob)].copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
        ring["dev"][:len(blob)].copy_(ring["host"][i][:len(blob)], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ring["done"][i] = ev
        return ring["dev"], len(recs), c0, keep

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        tables = [(g, self._records(g, gi)) for gi, g in enumerate(self.param_groups)]
        tables = [(g, t) for g, t in tables if t is not None]
        if not tables:
            return loss
        coef = None
        if self.max_grad_norm:
            total = sum(t[2] for _, t in tables)
            dev = tables[0][1][0].device
            if self._partials is None or self._partials.numel() < total:
                self._partials = torch.empty(total, dtype=torch.float32, device=dev)
                self._coef = torch.empty(2, dtype=torch.float32, device=dev)
            o = 0
            for _, (tab, nt, nchunks, _keep) in tables:
                _lib.check(lib.dclip_mt_sumsq_f32(tab.data_ptr(), nt, nchunks, self._partials.data_ptr() + 4 * o, stream),
                           "mt_sumsq")
                o += nchunks
            _lib.check(lib.dclip_clip_coef(self._partials.data_ptr(), total, float(self.max_grad_norm),
                                           self._coef.data_ptr(), self._coef.data_ptr() + 4, stream), "clip_coef")
            self.last_grad_norm = self._coef[1]
            coef = self._coef
        for group, (tab, nt, nchunks, _keep) in tables:
            b1, b2 = group["betas"]
            _lib.check(getattr(lib, self._entry)(tab.data_ptr(), nt, nchunks, float(group["lr"]), float(b1), float(b2),
                                                 float(group["eps"]), float(group["weight_decay"]),
                                                 None if coef is None else coef.data_ptr(), stream), self._entry)
        # the kernel wrote the parameters through raw pointers: tell autograd / every cache keyed on tensor versions
        # (HipCLIPModel._bf16_cache) that they changed
        touched = [p for group, _t in tables for p in group["params"] if p.grad is not None]
        if touched:
            torch._C._increment_version(touched)
        return loss


class FusedAdam(FusedAdamW):
    """torch.optim.Adam semantics (weight decay, default 0, is L2 added to the gradient) on the same multi-tensor
    kernel — the teacher trainer's optimizer (training/train_contrastive_teacher.py:245-248)."""
    _entry = "dclip_mt_adam_f32"

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 max_grad_norm: Optional[float] = None):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
