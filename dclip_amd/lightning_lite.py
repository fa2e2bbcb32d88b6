"""The few pieces of pytorch_lightning the reference leans on, restated (Lightning is not installed here and is
not on the hot path): a LightningModule-shaped base class, a `fit` loop with the same hook names, gradient
clipping / accumulation as configured in training/CLIP_image_distill_training.py:36-44, and a `.ckpt` writer /
reader with Lightning's dictionary layout (SURVEY.md §8b "Student checkpoint").

Host-side plumbing only — no arithmetic of the step lives here.
"""
from __future__ import annotations

import argparse
import os
from typing import Any, Dict, Iterable, Optional

import torch
import torch.nn as nn

LIGHTNING_VERSION_TAG = "2.0.0-dclip_amd"


class LightningLikeModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.hparams = argparse.Namespace()
        self._logged: Dict[str, float] = {}
        self.current_epoch = 0
        self.global_step = 0

    # -- LightningModule surface the reference uses
    def save_hyperparameters(self, hparams=None, ignore=None):
        if isinstance(hparams, argparse.Namespace):
            self.hparams = argparse.Namespace(**vars(hparams))
        elif isinstance(hparams, dict):
            self.hparams = argparse.Namespace(**hparams)

    @property
    def device(self) -> torch.device:
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    def log(self, name: str, value, prog_bar: bool = False, batch_size: Optional[int] = None, **kw):
        # keep the device tensor: reading it (.item()) would synchronise the stream every step
        self._logged[name] = value

    def logged(self, name: str) -> float:
        v = self._logged[name]
        return float(v.detach()) if isinstance(v, torch.Tensor) else float(v)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **init_kwargs):
        """`CLIPImageDistillation.load_from_checkpoint(path, map_location=, clip_model=, clip_preprocess=, strict=False)`
        as called by eval_scripts/flickr30k_eval.py:126-132."""
        # weights_only=True: nothing in the file is executed; argparse.Namespace (Lightning stores hparams as one)
        # is allow-listed explicitly.
        with torch.serialization.safe_globals([argparse.Namespace]):
            ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=True)
        hp = ckpt.get("hyper_parameters", {})
        hp = hp if isinstance(hp, argparse.Namespace) else argparse.Namespace(**dict(hp))
        model = cls(hp, **init_kwargs)
        model.load_state_dict(ckpt["state_dict"], strict=strict)
        return model


def save_checkpoint(path: str, module: LightningLikeModule, optimizer=None, scheduler=None, epoch: int = 0,
                    global_step: int = 0):
    """Lightning-layout dictionary: state_dict (prefixes `student.` / `teacher.cross_modal_attention.`),
    epoch, global_step, hyper_parameters, optimizer_states, lr_schedulers, callbacks, pytorch-lightning_version."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    ckpt = {
        "epoch": epoch,
        "global_step": global_step,
        "pytorch-lightning_version": LIGHTNING_VERSION_TAG,
        "state_dict": {k: v.detach().cpu() for k, v in module.state_dict().items()},
        "hyper_parameters": dict(vars(module.hparams)),
        "optimizer_states": [optimizer.state_dict()] if optimizer is not None else [],
        "lr_schedulers": [scheduler.state_dict()] if scheduler is not None else [],
        "callbacks": {},
    }
    tmp = path + ".tmp"
    torch.save(ckpt, tmp)
    os.replace(tmp, path)
    return path


def checkpoint_filename(epoch: int, train_loss: float) -> str:
    """ModelCheckpoint(filename="epoch-{epoch:02d}-{train_loss:.2f}") expands to e.g.
    `epoch-epoch=01-train_loss=3.01.ckpt` (training/CLIP_image_distill_training.py:27-34; the name
    eval_scripts/flickr30k_eval.py:113 expects)."""
    return f"epoch-epoch={epoch:02d}-train_loss={train_loss:.2f}.ckpt"


_END = object()


def _with_lookahead(iterable):
    """(index, item, next item or _END) with one item of look-ahead (loaders without __len__ work too)."""
    it = iter(iterable)
    try:
        prev = next(it)
    except StopIteration:
        return
    i = 0
    for cur in it:
        yield i, prev, cur
        prev = cur
        i += 1
    yield i, prev, _END


class Trainer:
    """`Trainer(max_epochs, accelerator="gpu", devices=N, precision=32, gradient_clip_val=0.5,
    accumulate_grad_batches=4, callbacks=[ModelCheckpoint(...)]).fit(model)` — the subset the launcher uses
    (training/CLIP_image_distill_training.py:36-45).

    `devices` > 1 (or an explicit `process_group`) trains data-parallel, one process per GPU over RCCL: the processes are
    started by `python -m torch.distributed.run --nproc-per-node N …` (the launcher script starts them itself when
    called bare, dclip_amd/CLIP_image_distill_training.py) and `fit` joins the rendezvous found in the environment.
    Every loader batch is a PER-GPU batch: batch i of the stream goes to rank i % N (`dist.shard_batches`), the loss uses
    global negatives (all-gathered embeddings), gradients are SUM-reduced in persistent buckets with the all-reduce
    launched from inside the LAST micro-batch's backward (`dist.GradSync`, accumulation keeps the overlap), rank 0 writes
    the checkpoints, and the logged / file-name `train_loss` is the global loss (`dist.global_loss_value`).  N ranks on
    batches b_0 … b_{N-1} perform exactly the update of one process on their concatenation."""

    def __init__(self, max_epochs: int = 1, gradient_clip_val: Optional[float] = 0.5, accumulate_grad_batches: int = 4,
                 checkpoint_dir: Optional[str] = None, save_top_k: int = 10, max_steps: Optional[int] = None,
                 use_hip_graph: bool = False, lr_interval: str = "epoch", devices: int = 1, process_group=None,
                 dist_backend: Optional[str] = None, bucket_mb: float = 25.0, **_ignored):
        """`use_hip_graph`: replay forward+backward from a captured HIP graph (dclip_amd/graph.py) — tensor batches of
        one fixed shape, single process; the update sequence and its results are those of the eager loop.
        `lr_interval`: "epoch" (default) advances the LR schedule once per epoch — what Lightning does with the
        reference's `return [optimizer], [scheduler]` (training/CLIP_image_distillation.py:679-682: a bare scheduler
        gets interval="epoch"), so with the default total_steps=1000 the LR decays by 0.1 % per epoch, as written.
        "step" advances it after every optimizer step (what the HF warm-up schedule was designed for)."""
        if lr_interval not in ("epoch", "step"):
            raise ValueError(f"lr_interval {lr_interval!r}")
        self.lr_interval = lr_interval
        self.use_hip_graph = use_hip_graph
        self.max_epochs = max_epochs
        self.clip = gradient_clip_val
        self.accum = max(1, accumulate_grad_batches)
        self.checkpoint_dir = checkpoint_dir
        self.save_top_k = save_top_k
        self.max_steps = max_steps
        self.saved = []          # (train_loss, path)
        if isinstance(devices, str):          # Lightning's "auto" / "-1": every GPU of the node; "2": two
            devices = int(devices) if devices.lstrip("-").isdigit() else -1
        if isinstance(devices, int):
            self.devices = devices if devices > 0 else max(1, torch.cuda.device_count())
        else:
            self.devices = len(devices)       # a list of device indices
        self.process_group = process_group
        self.dist_backend = dist_backend
        self.bucket_mb = bucket_mb
        self.grad_sync = None

    def _join_group(self):
        """The data-parallel group of this run: the one passed in, else (devices > 1) the rendezvous in the environment."""
        if self.process_group is not None or self.devices <= 1:
            return self.process_group
        import torch.distributed as tdist
        from . import dist as ddist
        if not tdist.is_initialized() and int(os.environ.get("WORLD_SIZE", "1")) != self.devices:
            raise RuntimeError(
                f"Trainer(devices={self.devices}) runs one process per GPU: start it with `python -m torch.distributed.run "
                f"--nnodes=1 --nproc-per-node {self.devices} --master-addr 127.0.0.1 <script> ...` "
                f"(found WORLD_SIZE={os.environ.get('WORLD_SIZE', 'unset')})")
        return ddist.init_from_env(self.dist_backend or os.environ.get("DCLIP_DIST_BACKEND"))

    def fit(self, model: LightningLikeModule, train_dataloaders: Optional[Iterable] = None,
            val_dataloaders: Optional[Iterable] = None):
        group = self._join_group()
        world, rank = 1, 0
        if group is not None:
            import torch.distributed as tdist
            from . import dist as ddist
            world, rank = tdist.get_world_size(group), tdist.get_rank(group)
            if self.use_hip_graph:
                raise RuntimeError("use_hip_graph captures a single-process step; data-parallel runs launch eagerly")
            model.process_group = group                   # global negatives + the per-rank loss share (dist.py)
        opts, scheds = model.configure_optimizers()
        opt, sched = opts[0], (scheds[0] if scheds else None)
        # the HIP optimizer clips the global norm itself (device-side coefficient, no host sync, one pass over the grads)
        fused_clip = bool(self.clip) and hasattr(opt, "max_grad_norm")
        if fused_clip:
            opt.max_grad_norm = self.clip
        sync = None
        if group is not None:
            sync = ddist.GradSync([p for g in opt.param_groups for p in g["params"]], group, bucket_mb=self.bucket_mb)
            self.grad_sync = sync
        train = train_dataloaders if train_dataloaders is not None else model.train_dataloader()
        val = val_dataloaders if val_dataloaders is not None else (
            model.val_dataloader() if hasattr(model, "val_dataloader") else None)
        step = 0
        graphed, acc, gparams = None, None, None
        prefetch = getattr(model, "prefetch_teacher", None)      # cross-step pipelining of a frozen teacher, where the module has it
        for epoch in range(self.max_epochs):
            model.current_epoch = epoch
            model.train()
            if graphed is None:
                opt.zero_grad(set_to_none=True)
            elif acc is not None:
                torch._foreach_zero_(acc)          # nothing of the previous epoch leaks into this epoch's first update
            last = None
            # a loader that already yields only this rank's share (DistributedSampler inside: `rank_sharded`) is taken as it
            # is; any other stream of per-GPU batches is dealt out round-robin
            if group is not None and getattr(train, "rank_sharded", False) and hasattr(getattr(train, "sampler", None), "set_epoch"):
                train.sampler.set_epoch(epoch)
            stream = train if (group is None or getattr(train, "rank_sharded", False)) else ddist.shard_batches(train, rank, world)
            for i, batch, upcoming in _with_lookahead(stream):
                is_last = upcoming is _END
                # Lightning steps on every `accum`-th batch AND on the last batch of the epoch (a trailing partial
                # group is not dropped); the divisor stays `accum` there as well
                boundary = (i + 1) % self.accum == 0 or is_last
                if self.use_hip_graph:
                    if graphed is None:
                        from .graph import GraphedStep
                        graphed = GraphedStep(model, batch)
                        gparams = [p for p in model.parameters() if p.requires_grad and p.grad is not None]
                        if self.accum > 1:
                            acc = [torch.zeros_like(p) for p in gparams]
                    loss = graphed.step(batch)
                    last = loss.detach().clone()
                    if acc is not None:        # the graph ASSIGNS this micro-batch's gradients; accumulate by hand
                        torch._foreach_add_(acc, [p.grad for p in gparams], alpha=1.0 / self.accum)
                        if boundary:
                            torch._foreach_copy_([p.grad for p in gparams], acc)
                            torch._foreach_zero_(acc)
                elif sync is not None:
                    loss = model.training_step(batch)          # this rank's SHARE of the global loss
                    if prefetch is not None and not is_last:
                        prefetch(upcoming)
                    if boundary:
                        with sync.hooks():                     # all-reduce launched from inside this backward
                            (loss / self.accum).backward()
                        sync.finish()
                    else:
                        (loss / self.accum).backward()         # plain accumulation into .grad
                    last = (loss.detach(), dict(getattr(model, "last_losses", {})))
                else:
                    loss = model.training_step(batch)
                    if prefetch is not None and not is_last:
                        # the NEXT batch's frozen meta-teacher starts now, on its own stream, and runs beside this batch's
                        # backward and the optimizer (CLIPImageDistillation.prefetch_teacher; tensor batches only)
                        prefetch(upcoming)
                    (loss / self.accum).backward()
                    last = loss.detach()
                if boundary:
                    if self.clip and not fused_clip:
                        torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], self.clip)
                    opt.step()
                    if sched is not None and self.lr_interval == "step":
                        sched.step()
                    if graphed is None:            # graph mode: gradients live in the graph's pool and are overwritten
                        opt.zero_grad(set_to_none=True)
                    step += 1
                    model.global_step = step
                if self.max_steps is not None and step >= self.max_steps:
                    break
            if sched is not None and self.lr_interval == "epoch":
                sched.step()
            if val is not None:
                model.eval()
                with torch.no_grad():
                    for batch in (val if (group is None or getattr(val, "rank_sharded", False))
                                  else ddist.shard_batches(val, rank, world)):
                        model.validation_step(batch)
            if group is not None and last is not None:
                # the value Lightning would log on one process: sum of the ranks' shares (one small all-reduce per epoch)
                share, parts = last
                if parts:
                    last = ddist.global_loss_value(parts["loss_image"], parts["loss_text"], parts["loss_contrastive"], group)
                else:
                    last = share.clone()
                    tdist.all_reduce(last, group=group)
                model.log("train_loss", last)
            if self.checkpoint_dir and last is not None:
                tl = float(last)
                path = os.path.join(self.checkpoint_dir, checkpoint_filename(epoch, tl))
                if rank == 0:
                    save_checkpoint(path, model, opt, sched, epoch, step)
                    self.saved.append((tl, path))
                    self.saved.sort()
                    for _, stale in self.saved[self.save_top_k:]:
                        if os.path.exists(stale):
                            os.remove(stale)
                    self.saved = self.saved[:self.save_top_k]
                if group is not None:
                    tdist.barrier(group=group)      # nobody races ahead of (or reads) a half-written checkpoint
            if self.max_steps is not None and step >= self.max_steps:
                break
        return model
