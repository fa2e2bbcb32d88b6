"""Data parallelism for the distillation step: one process per GPU, `torch.distributed` over RCCL (backend
"nccl" on ROCm) on the node's xGMI mesh; `gloo` on CPU for tests.

The reference is single-GPU (training/CLIP_image_distill_training.py:38-39) — nothing here has a counterpart
there; the contract is only that the N-rank loss/gradients equal the single-process ones on the concatenated
batch (fixture F5):
  C1  global negatives: all-gather of the normalised embeddings + of two LSE vectors, inside
      functional.ContrastiveLossFn (forward only — its backward needs no collective).
  C2  gradient all-reduce of the trainable parameters, bucketed, launched from INSIDE the hand-written backward
      (grad-ready hook, one bucket per ViT layer) so RCCL's transfers overlap the layers still to be back-propagated.

Loss scaling: ContrastiveLossFn returns each rank's SHARE of the global loss (sum over ranks = the
single-process value), so its gradients must be SUMMED over ranks; the per-sample cosine losses are means
over the local batch, so theirs must be AVERAGED.  `local_loss_for_backward` pre-scales the terms so that one
SUM all-reduce of the gradients is exact.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """Rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run sets them)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return None
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC, what RCCL needs on this driver) must be in the environment BEFORE
        # the first HIP call of the process — callers that have already touched the GPU (torch.cuda.set_device) are too
        # late for a setdefault here.  dclip_amd/__init__.py and bench.py set it at import time.
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend)
    return dist.group.WORLD


def local_loss_for_backward(loss_image, loss_text, loss_contrastive_share, world: int):
    """What each rank back-propagates so that SUM-reducing gradients over ranks reproduces the single-process
    gradient of  mean_global(L_img) + mean_global(L_txt) + L_con."""
    return (loss_image + loss_text) / world + loss_contrastive_share


def global_loss_value(loss_image, loss_text, loss_contrastive_share, group) -> torch.Tensor:
    """The single-process loss value (for logging): all-reduce of the per-rank pieces."""
    world = dist.get_world_size(group)
    v = ((loss_image + loss_text) / world + loss_contrastive_share).detach().clone()
    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    return v


class GradSync:
    """Bucketed SUM all-reduce of the trainable parameters' gradients over the data-parallel group, overlapped with
    the backward pass.

    `on_grads_ready` is installed as functional's grad-ready hook: the vision tower's backward hands over each
    layer's gradients the moment they are final (top layer first).  They are packed into a flat fp32 bucket
    (~`bucket_mb`, one ViT-B layer = 28 MB) and the bucket's all-reduce is launched with async_op — RCCL runs it on
    its own stream over xGMI while the compute stream keeps back-propagating the layers below.  `finish()` (after
    `loss.backward()`) reduces whatever the hook never saw (parameters outside the hooked tower), waits for every
    bucket and points each `param.grad` at its slice of the reduced bucket (no copy back).  Without the hook
    installed, `reduce()` = finish() does everything after the backward (still bucketed and asynchronous among
    buckets).  One backward per `finish()`: with gradient accumulation call it after the LAST micro-batch only and
    without the overlap hook (the hook sees a micro-batch's gradients, not the accumulated ones)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group, bucket_mb: float = 25.0):
        self.group = group
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self._ids = {id(p) for p in self.params}
        self.bucket_elems = max(1, int(bucket_mb * (1 << 20) / 4))
        self._pending = []          # (param, grad) waiting for a bucket
        self._pending_elems = 0
        self._inflight = []         # (work, flat, [(param, offset, numel)])
        self._seen = set()
        self.reset_stats()

    # ---- bookkeeping for the bench line (bench.py "comm")
    def reset_stats(self):
        self._n_finish = 0
        self._n_buckets = 0
        self._n_bytes = 0
        self._wait_events = []      # (before, after) event pairs around finish()'s waits, on the compute stream

    def stats(self) -> dict:
        """Per-step averages since reset_stats(): buckets all-reduced, bytes all-reduced, and the EXPOSED all-reduce
        time (how long the compute stream sat in finish() waiting for RCCL after the backward had ended)."""
        n = max(1, self._n_finish)
        exposed = None
        if self._wait_events:
            torch.cuda.synchronize()
            exposed = sum(a.elapsed_time(b) for a, b in self._wait_events) / n
        return {"grad_buckets_per_step": self._n_buckets / n, "grad_allreduce_bytes_per_step": self._n_bytes / n,
                "grad_allreduce_exposed_ms_per_step": None if exposed is None else round(exposed, 4),
                "bucket_mb": round(self.bucket_elems * 4 / (1 << 20), 2)}

    # ---- called during backward
    def on_grads_ready(self, pairs):
        if self.group is None:
            return
        for p, g in pairs:
            if g is None or id(p) not in self._ids:
                continue
            if id(p) in self._seen:
                # a second delivery within one backward (a tower applied twice) would be dropped silently and
                # finish() would then overwrite the correctly accumulated p.grad with the first-only reduction
                raise RuntimeError("GradSync: a parameter's gradient was delivered twice in one backward; use "
                                   "reduce() after the backward (no overlap hook) for modules applied more than once")
            self._seen.add(id(p))
            self._pending.append((p, g))
            self._pending_elems += g.numel()
        if self._pending_elems >= self.bucket_elems:
            self._launch()

    def _launch(self):
        if not self._pending:
            return
        dev = self._pending[0][1].device
        flat = torch.empty(self._pending_elems, dtype=torch.float32, device=dev)
        layout, o = [], 0
        for p, g in self._pending:
            n = g.numel()
            layout.append((p, o, n))
            o += n
        # ONE multi-tensor copy packs the bucket (a copy per tensor is ~150 tiny launches per step)
        torch._foreach_copy_([flat[o:o + n] for _, o, n in layout], [g.reshape(-1) for _, g in self._pending])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((work, flat, layout))
        self._n_buckets += 1
        self._n_bytes += flat.numel() * 4
        self._pending, self._pending_elems = [], 0

    # ---- called after backward
    def finish(self):
        if self.group is None:
            return
        for p in reversed(self.params):                 # whatever no hook delivered (other towers, heads)
            if p.grad is not None and id(p) not in self._seen:
                self._seen.add(id(p))
                self._pending.append((p, p.grad))
                self._pending_elems += p.grad.numel()
                if self._pending_elems >= self.bucket_elems:
                    self._launch()
        self._launch()
        timed = bool(self._inflight) and self._inflight[0][1].is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for work, flat, layout in self._inflight:
            work.wait()
            for p, o, n in layout:                        # zero-copy: .grad becomes a view of the reduced bucket
                if p.grad is not None:
                    p.grad = flat[o:o + n].view_as(p)
        if timed:
            e1.record()
            self._wait_events.append((e0, e1))
        self._n_finish += 1
        self._inflight, self._seen = [], set()

    reduce = finish
