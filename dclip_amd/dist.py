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


def init_from_env(backend: Optional[str] = None, timeout_s: float = 180.0):
    """Rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run sets them).
    `timeout_s` bounds the rendezvous AND every collective (RCCL's watchdog aborts the process when one has been pending
    that long; gloo raises from `wait()`): a rank that dies or falls behind cannot park the others for the backend's
    default 10 / 30 minutes."""
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
        import datetime
        dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=float(timeout_s)))
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
    the backward pass, on PERSISTENT flat fp32 buckets.

    Layout: the parameters are walked in REVERSE registration order (the order a backward produces them: heads, the
    vision layers from the top down, the embeddings) and cut into buckets of ~`bucket_mb` (one ViT-B layer = 28 MB);
    every parameter owns a fixed slice of a bucket that is allocated once.
      * `grad_buffer` (installed as functional.set_grad_alloc) hands that slice to the hand-written backward, so the
        weight-gradient GEMMs / LayerNorm and bias reductions WRITE INTO THE BUCKET — no packing copy;
      * `on_grads_ready` (functional.set_grad_ready_hook) is called by the vision tower's backward the moment a group
        of gradients is final (top layer first); a gradient that was not written in place is copied into its slice; when
        the last parameter of a bucket has arrived the bucket's all-reduce is launched with async_op — RCCL moves it over
        xGMI on its own stream while the compute stream keeps back-propagating the layers below.  It returns True: the
        reducer owns these gradients (autograd is handed None for them — nothing is accumulated or cloned);
      * `finish()` (after `loss.backward()`) takes whatever no hook delivered from `.grad` (parameters outside the hooked
        tower), zero-fills slices of parameters that received no gradient, launches the remaining buckets, waits, and
        points EVERY parameter's `.grad` at its slice of the reduced bucket (a parameter that produced no gradient on
        this rank still receives the other ranks' sum, as under DDP: replicas cannot diverge).
    Without the hooks installed, `reduce()` = finish() does everything after the backward (still bucketed and
    asynchronous among buckets).

    Gradient accumulation (the reference trains with accumulate_grad_batches=4,
    training/CLIP_image_distill_training.py:42): run the first micro-batches WITHOUT the hooks (`hooks()` not entered;
    autograd accumulates into `.grad` as usual) and the LAST one inside `with sync.hooks():` — a hook delivery adds the
    `.grad` accumulated so far to the micro-batch's gradient in the bucket slice and clears it, so the all-reduce of the
    accumulated gradient still overlaps the last micro-batch's backward.  One `finish()` per optimizer step.
    The buckets are reused every step: the optimizer must have consumed `.grad` before the next hooked backward starts
    (same stream: it has), and `.grad` must be cleared (`zero_grad(set_to_none=True)`) before the next un-hooked one."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group, bucket_mb: float = 25.0, timing: bool = False):
        self.group = group
        self.timing = timing        # record HIP events around finish()'s waits (bench.py: "exposed" all-reduce time)
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self._ids = {id(p) for p in self.params}
        self.bucket_elems = max(1, int(bucket_mb * (1 << 20) / 4))
        # bucket plan: [(first param index, ...)] over the reversed parameter list; buffers allocated lazily (device of p)
        self._slot = {}             # id(param) -> (bucket index, offset, numel)
        self._bucket_params: List[List[torch.nn.Parameter]] = []
        self._bucket_size: List[int] = []
        cur, size = [], 0
        for p in reversed(self.params):
            n = (p.numel() + 3) // 4 * 4                       # 16-byte aligned slices (the GEMM writes C there)
            self._slot[id(p)] = (len(self._bucket_params), size, p.numel())
            cur.append(p)
            size += n
            if size >= self.bucket_elems:
                self._bucket_params.append(cur)
                self._bucket_size.append(size)
                cur, size = [], 0
        if cur:
            self._bucket_params.append(cur)
            self._bucket_size.append(size)
        self._flat: List[Optional[torch.Tensor]] = [None] * len(self._bucket_params)
        self._arrived = [0] * len(self._bucket_params)
        self._launched = [False] * len(self._bucket_params)
        self._work = []             # (bucket index, work)
        self._seen = set()
        self.reset_stats()

    # ---- bookkeeping for the bench line (bench.py "comm")
    def reset_stats(self):
        self._n_finish = 0
        self._n_buckets = 0
        self._n_bytes = 0
        self._n_inplace = 0
        self._n_copied = 0
        self._wait_events = []      # (before, after) event pairs around finish()'s waits (only with timing=True)
        self._wait_ms = 0.0         # pairs already read out (the list is drained every 64 steps: bounded)

    def _drain_wait_events(self):
        if self._wait_events:
            torch.cuda.synchronize()
            self._wait_ms += sum(a.elapsed_time(b) for a, b in self._wait_events)
            self._wait_events = []

    def stats(self) -> dict:
        """Per-step averages since reset_stats(): buckets all-reduced, bytes all-reduced, gradients written in place vs
        copied into their slice, and (timing=True) the EXPOSED all-reduce time — how long the compute stream sat in
        finish() waiting for RCCL after the backward had ended."""
        n = max(1, self._n_finish)
        exposed = None
        if self.timing and (self._wait_events or self._wait_ms):
            self._drain_wait_events()
            exposed = self._wait_ms / n
        return {"grad_buckets_per_step": self._n_buckets / n, "grad_allreduce_bytes_per_step": self._n_bytes / n,
                "grad_tensors_written_in_place_per_step": self._n_inplace / n,
                "grad_tensors_copied_per_step": self._n_copied / n,
                "grad_allreduce_exposed_ms_per_step": None if exposed is None else round(exposed, 4),
                "bucket_mb": round(self.bucket_elems * 4 / (1 << 20), 2)}

    def bucket_allreduce_ms(self, repeats: int = 3) -> List[float]:
        """Calibration leg (outside any timed region): every bucket all-reduced ALONE, synchronously, `repeats` times;
        mean milliseconds per bucket, in launch order.  With the bucket sizes this gives the bus rate the collective
        reaches when nothing overlaps it.  Collective call: every rank must make it.  The buckets' contents are
        multiplied by world**repeats — call it only between steps (the next backward overwrites them)."""
        out = []
        for b in range(len(self._bucket_params)):
            flat = self._flat[b]
            if flat is None:
                out.append(0.0)
                continue
            flat.zero_()
            if flat.is_cuda and self._host_staged():       # gloo rehearsal: what is timed is the staged round trip
                import time
                t0 = time.perf_counter()
                for _ in range(repeats):
                    host = self._host_mirror(b)
                    host.copy_(flat)
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                    flat.copy_(host)
                torch.cuda.synchronize()
                out.append((time.perf_counter() - t0) * 1e3 / repeats)
                continue
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)          # warm: connections, protocol choice
            if flat.is_cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(repeats):
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                e1.record()
                e1.synchronize()
                out.append(e0.elapsed_time(e1) / repeats)
            else:
                import time
                t0 = time.perf_counter()
                for _ in range(repeats):
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                out.append((time.perf_counter() - t0) * 1e3 / repeats)
        return out

    def bucket_bytes(self) -> List[int]:
        return [4 * n for n in self._bucket_size]

    # ---- buffers
    def _bucket(self, b: int, device) -> torch.Tensor:
        f = self._flat[b]
        if f is None:
            f = torch.empty(self._bucket_size[b], dtype=torch.float32, device=device)
            self._flat[b] = f
        return f

    def _view(self, p: torch.nn.Parameter) -> torch.Tensor:
        b, o, n = self._slot[id(p)]
        return self._bucket(b, p.device)[o:o + n]

    def grad_buffer(self, p: torch.nn.Parameter, shape) -> Optional[torch.Tensor]:
        """The tensor the backward should write p's gradient into (its bucket slice, viewed as `shape`), or None."""
        if self.group is None or id(p) not in self._ids:
            return None
        numel = 1
        for d in shape:
            numel *= int(d)
        if numel != p.numel():
            return None
        return self._view(p).view(tuple(shape))

    def hooks(self):
        """Context manager: install `on_grads_ready` / `grad_buffer` as the towers' data-parallel hooks for ONE backward
        (the last micro-batch of an accumulation group, or the only one), remove them on exit."""
        import contextlib
        from . import functional

        @contextlib.contextmanager
        def _cm():
            functional.set_grad_ready_hook(self.on_grads_ready)
            functional.set_grad_alloc(self.grad_buffer)
            try:
                yield self
            finally:
                functional.set_grad_ready_hook(None)
                functional.set_grad_alloc(None)
        return _cm()

    # ---- called during backward
    def _deliver(self, p, g, from_hook: bool = False):
        if id(p) in self._seen:
            # a second delivery within one backward (a tower applied twice) would overwrite the first in the bucket
            raise RuntimeError("GradSync: a parameter's gradient was delivered twice in one backward; use "
                               "reduce() after the backward (no overlap hooks) for modules applied more than once")
        self._seen.add(id(p))
        v = self._view(p)
        if g is None:
            v.zero_()
        elif g.data_ptr() == v.data_ptr():
            self._n_inplace += 1
        else:
            v.copy_(g.reshape(-1))
            self._n_copied += 1
        if from_hook and p.grad is not None:
            # earlier micro-batches of this accumulation group (run without the hooks) left their sum in .grad
            if p.grad.data_ptr() != v.data_ptr():
                v.add_(p.grad.reshape(-1))
            p.grad = None
        b = self._slot[id(p)][0]
        self._arrived[b] += 1
        if self._arrived[b] == len(self._bucket_params[b]):
            self._launch(b)

    def on_grads_ready(self, pairs) -> bool:
        if self.group is None:
            return False
        for p, g in pairs:
            if g is None or id(p) not in self._ids:
                continue
            self._deliver(p, g, from_hook=True)
        return True

    def _launch(self, b: int):
        if self._launched[b]:
            return
        self._launched[b] = True
        flat = self._flat[b]
        pad_lo = 0
        for p in self._bucket_params[b]:                      # alignment gaps between slices carry no data: keep them finite
            _, o, n = self._slot[id(p)]
            if o > pad_lo:
                flat[pad_lo:o].zero_()
            pad_lo = o + n
        if pad_lo < flat.numel():
            flat[pad_lo:].zero_()
        if flat.is_cuda and self._host_staged():
            # gloo has no device path of its own worth using: its CUDA all-reduce allocates pinned memory per collective and
            # was measured degrading from 8 ms to 12 s per 28 MB bucket with four processes on one card.  A device bucket is
            # staged through ONE persistent pinned mirror instead (rehearsals and tests only: the product backend is RCCL).
            host = self._host_mirror(b)
            host.copy_(flat, non_blocking=True)
            torch.cuda.current_stream(flat.device).synchronize()
            work = dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._work.append((b, work))
        self._n_buckets += 1
        self._n_bytes += flat.numel() * 4

    def _unstage(self, b: int):
        """gloo + device bucket: the reduced host mirror goes back to the bucket (ordered on the compute stream)."""
        flat = self._flat[b]
        if flat is not None and flat.is_cuda and self._host_staged():
            flat.copy_(self._host_mirror(b), non_blocking=True)

    def _host_staged(self) -> bool:
        hs = getattr(self, "_hs", None)
        if hs is None:
            hs = self._hs = dist.get_backend(self.group) == "gloo"
        return hs

    def _host_mirror(self, b: int) -> torch.Tensor:
        mirrors = self.__dict__.setdefault("_mirrors", {})
        m = mirrors.get(b)
        if m is None:
            m = mirrors[b] = torch.empty(self._bucket_size[b], dtype=torch.float32).pin_memory()
        return m

    # ---- called after backward
    def finish(self):
        if self.group is None:
            return
        for p in reversed(self.params):                 # whatever no hook delivered (other towers, heads)
            if id(p) not in self._seen:
                self._deliver(p, p.grad)                # None: the slice is zero-filled (the other ranks may have one)
        timed = self.timing and bool(self._work) and self._flat[self._work[0][0]].is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if os.environ.get("DCLIP_SYNC_TRACE"):          # debugging aid: how long the host sat in each bucket's wait()
            import sys
            import time
            t0 = time.perf_counter()
            waits = []
            for b, work in self._work:
                work.wait()
                self._unstage(b)
                waits.append((b, round(time.perf_counter() - t0, 3)))
            print(f"[GradSync rank {dist.get_rank(self.group)}] finish(): bucket -> seconds since entry {waits}",
                  file=sys.stderr, flush=True)
        else:
            for b, work in self._work:
                work.wait()
                self._unstage(b)
        if timed:
            e1.record()
            self._wait_events.append((e0, e1))
            if len(self._wait_events) >= 64:
                self._drain_wait_events()
        for p in self.params:                           # zero-copy: .grad is a view of the reduced bucket
            p.grad = self._view(p).view_as(p)
        self._n_finish += 1
        self._work, self._seen = [], set()
        self._arrived = [0] * len(self._bucket_params)
        self._launched = [False] * len(self._bucket_params)

    reduce = finish


def shard_batches(batches: Iterable, rank: int, world: int):
    """Rank `rank`'s share of a stream of per-GPU batches: batch i goes to rank i % world; a trailing group that does
    not cover every rank is dropped, so all ranks run the SAME number of steps (a rank with one step more would wait in
    its collectives for ever).  The single-process equivalent of one N-rank step is the concatenation of the N batches
    of a group."""
    group = []
    for b in batches:
        group.append(b)
        if len(group) == world:
            yield group[rank]
            group = []
