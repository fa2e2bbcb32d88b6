"""`CLIPImageDistillation` — the student distillation module, same name and call surface as the reference's
LightningModule (training/CLIP_image_distillation.py:439-763), with every dense operation on the HIP kernels.

Two batch formats are accepted by `training_step` / `validation_step`:
  * the reference's tuple `(images[B,3,224,224], captions: list[str], image_paths: list[str], weighted_boxes)`
    (:594) — needs `clip_preprocess` for tokenisation and real image files for the teacher's crops;
  * a tensor dict for synthetic / pre-tokenised data (BASELINE.json configs): `pixel_values`, `input_ids`, and
    either `teacher_image_emb` [B,P] or `regions` [B,R,3,H,W] (+ `region_counts` [B]) for the meta-teacher.

freeze_mode:
  "north_star"  text tower frozen, vision tower fully trainable (README.md:7; BASELINE.json north_star).
  "as_written"  what the reference's __init__ actually leaves trainable (:504-506 and nothing else): vision
                q/k/v/out projections + both projection heads + logit_scale + the whole text tower.
"""
from __future__ import annotations

import argparse
from typing import Dict, Optional

import torch

from . import functional
from .clip_model import HipCLIPModel, from_hf_state_dict
from .config import from_hf
from .lightning_lite import LightningLikeModule


def distill_losses(student_image: torch.Tensor, student_text: torch.Tensor, teacher_image: torch.Tensor,
                   teacher_text: torch.Tensor, temperature: float = 0.05, group=None) -> Dict[str, torch.Tensor]:
    """loss = L_img + L_txt + 1.0 * L_con  (training/CLIP_image_distillation.py:602, :618, :621-628)."""
    l_img = functional.cosine_distillation_loss(student_image, teacher_image)
    l_txt = functional.cosine_distillation_loss(student_text, teacher_text)
    l_con = functional.contrastive_loss(student_image, student_text, temperature, group)
    return {"loss": l_img + l_txt + 1.0 * l_con, "loss_image": l_img, "loss_text": l_txt, "loss_contrastive": l_con}


def bridge_weight(student_dim: int, teacher_dim: int, seed: int = 0) -> torch.Tensor:
    """The frozen teacher→student bridge of BASELINE config c5 (ViT-L/14 teacher, P=768 → ViT-B/32 student, P=512):
    a seeded Gaussian [student_dim, teacher_dim] matrix, std teacher_dim^-1/2, drawn on the CPU with an explicit
    generator so every rank / run / the oracle derive the same matrix.

    DECLARED RULE — the reference has none: its `cosine_distillation_loss` would raise on the width mismatch
    (training/CLIP_image_distillation.py:573; "FIX THIS IF GOING FROM VIT L TO VIT B",
    training/patch_text_aggregation.py:51).  A random projection keeps the cosine geometry of the teacher space up to
    Johnson–Lindenstrauss distortion, has no trainable state, and is stored in the student checkpoint under its own
    key (`teacher_bridge.weight`) so a run can be resumed / evaluated against the same targets."""
    gen = torch.Generator().manual_seed(1_000_003 + int(seed))
    return torch.randn((student_dim, teacher_dim), generator=gen, dtype=torch.float32) * float(teacher_dim) ** -0.5


class TeacherBridge(torch.nn.Module):
    """t [B, P_teacher] -> t @ W^T [B, P_student]; W is a frozen buffer (never a Parameter: AdamW must not see it)."""

    def __init__(self, student_dim: int, teacher_dim: int, seed: int = 0):
        super().__init__()
        self.register_buffer("weight", bridge_weight(student_dim, teacher_dim, seed))

    @torch.no_grad()
    def forward(self, t: torch.Tensor) -> torch.Tensor:
        from . import ops
        return ops.gemm(t.float().contiguous(), self.weight, ops.LAYOUT_NT)


def _as_hip_model(clip_model) -> HipCLIPModel:
    if isinstance(clip_model, HipCLIPModel):
        return clip_model
    if hasattr(clip_model, "config") and hasattr(clip_model, "state_dict"):       # an HF CLIPModel
        dev = next(clip_model.parameters()).device
        return from_hf_state_dict(from_hf(clip_model.config), clip_model.state_dict(), device=dev)
    raise TypeError("clip_model must be a dclip_amd HipCLIPModel or a transformers CLIPModel")


class CLIPImageDistillation(LightningLikeModule):
    def __init__(self, hparams, clip_model, clip_preprocess=None, teacher=None, freeze_mode: str = "north_star",
                 process_group=None, contrastive_teacher_path: Optional[str] = None, student_precision: str = "fp32"):
        """`student_precision`: "fp32" (default: the reference's `precision=32`,
        training/CLIP_image_distill_training.py:40, and the benched config c2) or "bf16" — the student's VISION tower
        multiplies in bf16 (forward, dgrad, wgrad) with fp32 master weights, fp32 accumulation and fp32
        LayerNorm / softmax / losses: what BASELINE configs c3 / c5 quote ("bf16 MFMA")."""
        super().__init__()
        if student_precision not in ("fp32", "bf16"):
            raise ValueError(f"student_precision {student_precision!r}")
        self.student_precision = student_precision
        # Run the FROZEN text tower's forward on a second HIP stream beside the vision tower's forward (they share nothing
        # until the loss): the two GEMM chains fill each other's tile-round tails and per-workgroup prologue / epilogue gaps.
        # None = on unless DCLIP_TEXT_STREAM=0 (eager loop 63.1 -> 62.2 ms at the benched size, capturable in a HIP graph).
        # Per-kernel timings (bench.py's GEMM events, rocprof) are only meaningful for kernels that run alone: bench.py sets
        # it False for its event-bracketed steps.
        self.overlap_frozen_text = None
        self._text_stream = None
        # Run the meta-teacher (tensor batches with `regions`) on a second HIP stream beside the student's image forward: the
        # two are independent until the loss, the bf16 student's GEMMs have 150 tiles for 256 CUs (one 128-KiB workgroup per
        # CU) and its LayerNorm / attention launches leave most of the chip idle — the teacher's 3600-tile GEMMs fill it.
        # Same box, alternating: c3 in bf16 43.14 -> 42.26 ms, c5 728.2 -> 726.1 ms, c3 with the fp32 student 90.59 -> 90.41 ms;
        # losses bit-identical.  None = on (DCLIP_TEACHER_STREAM=0 switches it off); not inside a HIP-graph capture.
        self.overlap_teacher = None
        self._teacher_stream = None
        self._prefetched = None          # (key, teacher image target, teacher sentence embedding) of prefetch_teacher()
        self.save_hyperparameters(hparams, ignore="clip_model")
        self.student = _as_hip_model(clip_model)
        self.preprocess = clip_preprocess
        self.process_group = process_group
        self.temperature = 0.05
        if teacher is None:
            # The reference's teacher loads its OWN CLIP instances and never updates them
            # (training/image_tokenizer.py:25, training/text_tokenizer.py:21).  Nothing is fetched by name here, so the
            # default teacher takes a frozen SNAPSHOT of the student's towers at construction: the distillation target
            # must not move with the student (the vision tower trains in both freeze modes).
            import copy
            from .patch_text_aggregation import PatchTextAggregation
            E = self.student.config.projection_dim               # 512 / 8 heads in the reference (:446-452)
            snapshot = copy.deepcopy(self.student)
            object.__setattr__(snapshot, "_bf16_w", None)
            for p in snapshot.parameters():
                p.requires_grad = False
            teacher = PatchTextAggregation(embed_dim=E, num_heads=max(1, E // 64), clip_model=snapshot,
                                           owns_clip=True, text_twin=self.student)
            teacher.to(next(self.student.parameters()).device)
        self.teacher = teacher
        if contrastive_teacher_path:
            # weights_only: a checkpoint is data, nothing in it is executed (training/CLIP_image_distillation.py:458-462)
            sd = torch.load(contrastive_teacher_path, map_location="cpu", weights_only=True)
            self.teacher.load_state_dict(sd, strict=False)
        self.teacher.eval()
        # config c5: a teacher wider than the student (L/14 → B/32) is bridged by the declared frozen projection
        t_dim, s_dim = int(self.teacher.embed_dim), int(self.student.config.projection_dim)
        self.teacher_bridge = None if t_dim == s_dim else TeacherBridge(
            s_dim, t_dim, int(getattr(self.hparams, "bridge_seed", 0)))
        self.set_freeze_mode(freeze_mode)

    # ------------------------------------------------------------------ freeze rules
    def set_freeze_mode(self, mode: str):
        if mode not in ("north_star", "as_written"):
            raise ValueError(f"freeze_mode {mode!r}")
        self.freeze_mode = mode
        for p in self.student.parameters():
            p.requires_grad = True
        if mode == "as_written":
            for name, param in self.student.vision_model.named_parameters():      # verbatim rule, :504-506
                if "proj" not in name:
                    param.requires_grad = False
        else:
            for p in self.student.text_model.parameters():
                p.requires_grad = False
            self.student.text_projection.weight.requires_grad = False
            self.student.logit_scale.requires_grad = False        # unused by the loss (temperature is 0.05)
        for p in self.teacher.parameters():
            p.requires_grad = False

    # ------------------------------------------------------------------ reference methods
    def forward(self, text=None, image=None):
        """:508-527 — `image` may be PIL (needs clip_preprocess) or a ready [B,3,H,W] tensor; `text` a str /
        list[str] (needs clip_preprocess) or an int64 id tensor."""
        if image is not None:
            if not isinstance(image, torch.Tensor):
                image = self.preprocess(images=image, return_tensors="pt")["pixel_values"]
            return self.student.get_image_features(pixel_values=image.to(self.device))
        if text is not None:
            if not isinstance(text, torch.Tensor):
                text = self.preprocess(text=text, return_tensors="pt", padding=True, truncation=True,
                                       max_length=77)["input_ids"]
            return self.student.get_text_features(input_ids=text.to(self.device))
        raise ValueError("Either text or image must be provided.")

    def compute_contrastive_loss(self, image_embeddings, text_embeddings, temperature=0.05):
        return functional.contrastive_loss(image_embeddings, text_embeddings, temperature, self.process_group)

    def cosine_distillation_loss(self, student_embeddings, teacher_embeddings):
        return functional.cosine_distillation_loss(student_embeddings, teacher_embeddings)

    def _tokenize(self, captions):
        if isinstance(captions, torch.Tensor):
            return captions
        if self.preprocess is None:
            raise RuntimeError("caption strings need clip_preprocess (an HF CLIPProcessor loaded from a local path)")
        return self.preprocess(text=captions, return_tensors="pt", padding=True, truncation=True)["input_ids"]

    @staticmethod
    def _batch_key(batch):
        r, t = batch["regions"], batch["input_ids"]
        return (r.data_ptr(), tuple(r.shape), r._version, t.data_ptr(), tuple(t.shape), t._version)

    def prefetch_teacher(self, batch) -> bool:
        """Start the meta-teacher for a tensor batch that will be passed to `training_step` LATER (software pipelining across
        steps: the teacher is frozen, its targets do not depend on the student's update).  Call it between
        `training_step(current)` and `loss.backward()` with the NEXT batch: the teacher's stream forks after the current
        forward, so the teacher's ~20 ms of work run beside the current backward, the optimizer and the next step's student
        forward — all of which leave CUs idle (150-tile GEMMs, LayerNorm / attention launches) — and `training_step(next)`
        joins it just before the loss.  Refused (False) when the teacher reads weights that are being trained (`_teacher_frozen`).
        Batches that carry `max_tokens` (token-padding size, known on the host) avoid the
        one host sync inside the teacher.  Returns False (nothing started) for batches without `regions`, on the CPU, inside a
        HIP-graph capture, or when second streams are switched off."""
        if not (isinstance(batch, dict) and "regions" in batch and "teacher_image_emb" not in batch):
            return False
        import os
        dev = self.device
        if dev.type != "cuda" or torch.cuda.is_current_stream_capturing() or os.environ.get("DCLIP_TEACHER_PREFETCH", "1") == "0":
            return False
        if self.overlap_teacher is False or os.environ.get("DCLIP_TEACHER_STREAM") == "0":
            return False
        if not self._teacher_frozen():
            return False
        from . import ops
        regions, tokens = batch["regions"].to(dev), batch["input_ids"].to(dev)
        main = torch.cuda.current_stream(dev)
        if self._teacher_stream is None:
            self._teacher_stream = torch.cuda.Stream(device=dev)
        self._teacher_stream.wait_stream(main)                       # fork behind what is queued now (the current forward)
        with torch.cuda.stream(self._teacher_stream), torch.no_grad(), ops.workspace_lane(3):
            target = self.teacher.compute_global_embedding_tensors(regions, tokens, batch.get("region_counts"),
                                                                   batch.get("max_tokens")).float()
            sentence = self.teacher.last_sentence_embedding
        self._prefetched = (self._batch_key(batch), target, sentence, regions, tokens)
        return True

    def _teacher_frozen(self) -> bool:
        """True when nothing the teacher reads is being trained: its cross-modal block and BOTH towers of its CLIP.  A
        teacher built on the student's own model (the towers shared) reads weights the optimizer is about to change — its
        targets for batch n+1 then depend on update n, and starting it early would read them before (or while) they are
        written."""
        clip = getattr(self.teacher, "_clip", None)
        towers = clip.parameters() if clip is not None else ()
        return not any(p.requires_grad for p in self.teacher.parameters()) and not any(p.requires_grad for p in towers)

    def _drop_prefetched(self) -> bool:
        """A prefetched target that is not this batch's (the loop skipped a batch) is released; always False."""
        self._prefetched = None
        return False

    def _teacher_beside_student(self, images) -> bool:
        import os
        on = self.overlap_teacher
        if on is None:
            env = os.environ.get("DCLIP_TEACHER_STREAM")
            on = env != "0"
        return bool(on) and images.is_cuda and not torch.cuda.is_current_stream_capturing()

    def _step(self, batch, log_name: str, bs_field: str):
        dev = self.device
        ran_teacher = True
        early_student_image = None
        grad_on = torch.is_grad_enabled()            # (the teacher branches below run under no_grad)
        self.teacher.last_sentence_embedding = None
        if isinstance(batch, dict) and "captions" in batch:
            # data.GpuCollate: decoded images already on the device, student preprocessing done there
            images = batch["pixel_values"].to(dev)
            host_tokens = self._tokenize(batch["captions"])
            tokens = host_tokens.to(dev)
            with torch.no_grad():
                teacher_image = self.teacher.compute_global_embedding_batch(
                    batch["image_paths"], host_tokens, batch["weighted_boxes"], batch.get("images_u8"),
                    batch.get("dims")).to(dev).float()
            teacher_text = None
        elif isinstance(batch, dict):
            images = batch["pixel_values"].to(dev)
            tokens = batch["input_ids"].to(dev)
            with torch.no_grad():
                if "teacher_image_emb" in batch:
                    teacher_image = batch["teacher_image_emb"].to(dev).float()
                    ran_teacher = False
                elif self._prefetched is not None and self._prefetched[0] == self._batch_key(batch):
                    # the teacher of THIS batch was started by prefetch_teacher() during the previous step: launch the student's
                    # forward, then join the teacher's stream
                    _key, teacher_image, sentence, _r, _t = self._prefetched
                    self._prefetched = None
                    main = torch.cuda.current_stream(dev)
                    with torch.set_grad_enabled(grad_on):
                        early_student_image = self.student.get_image_features(
                            pixel_values=images, precision=self.student_precision).float()
                    main.wait_stream(self._teacher_stream)                       # join
                    teacher_image.record_stream(main)
                    self.teacher.last_sentence_embedding = sentence
                    if sentence is not None:
                        sentence.record_stream(main)
                elif self._drop_prefetched() or self._teacher_beside_student(images):
                    # student image forward FIRST (asynchronous, main stream), then the teacher on its own stream: a host
                    # sync inside the teacher (token-padding size) then waits for the teacher's stream only
                    from . import ops
                    regions = batch["regions"].to(dev)
                    main = torch.cuda.current_stream(dev)
                    if self._teacher_stream is None:
                        self._teacher_stream = torch.cuda.Stream(device=dev)
                    self._teacher_stream.wait_stream(main)                       # fork: the batch is resident
                    with torch.set_grad_enabled(grad_on):
                        early_student_image = self.student.get_image_features(
                            pixel_values=images, precision=self.student_precision).float()
                    with torch.cuda.stream(self._teacher_stream), ops.workspace_lane(3):
                        teacher_image = self.teacher.compute_global_embedding_tensors(
                            regions, tokens, batch.get("region_counts"), batch.get("max_tokens")).float()
                    main.wait_stream(self._teacher_stream)                       # join
                    teacher_image.record_stream(main)
                    if self.teacher.last_sentence_embedding is not None:
                        self.teacher.last_sentence_embedding.record_stream(main)
                else:
                    teacher_image = self.teacher.compute_global_embedding_tensors(
                        batch["regions"].to(dev), tokens, batch.get("region_counts"), batch.get("max_tokens")).float()
                teacher_text = batch["teacher_text_emb"].to(dev).float() if "teacher_text_emb" in batch else None
        else:
            images, captions, image_paths, weighted_boxes_batch = batch
            images = images.to(dev, non_blocking=images.is_pinned())   # pinned by the DataLoader (pin_memory=True, :687)
            host_tokens = self._tokenize(captions)
            tokens = host_tokens.to(dev)
            with torch.no_grad():
                # the teacher gets the HOST ids: it sizes its token padding from them without a stream sync
                teacher_image = self.teacher.compute_global_embedding_batch(
                    image_paths, host_tokens, weighted_boxes_batch).to(dev).float()
            teacher_text = None
        if self.teacher_bridge is not None:
            if teacher_image.shape[1] != self.student.config.projection_dim:      # given targets may be pre-bridged
                teacher_image = self.teacher_bridge(teacher_image)
            if teacher_text is not None and teacher_text.shape[1] != self.student.config.projection_dim:
                teacher_text = self.teacher_bridge(teacher_text)
        shared_sentence = None
        if ran_teacher and self.teacher.shares_text_tower_with(self.student) \
                and getattr(self.teacher.text_tokenizer, "precision", "fp32") == "fp32":
            # The meta-teacher has just pushed these captions through the SAME frozen text tower to get its token
            # embeddings; the sentence embedding is row first-EOS of that pass.  One text forward serves the teacher's
            # tokens, the teacher's sentence target and the (frozen) student text features.
            shared_sentence = self.teacher.last_sentence_embedding
        text_frozen = not any(p.requires_grad for p in self.student.text_model.parameters()) \
            and not self.student.text_projection.weight.requires_grad
        text_precision = "bf16" if (self.student_precision == "bf16" and text_frozen) else "fp32"
        text_job = None
        text_beside = self.overlap_frozen_text
        if text_beside is None:
            import os
            text_beside = os.environ.get("DCLIP_TEXT_STREAM") != "0"
        if shared_sentence is None and text_frozen and text_beside and tokens.is_cuda:
            main = torch.cuda.current_stream(dev)
            if self._text_stream is None:
                self._text_stream = torch.cuda.Stream(device=dev)
            self._text_stream.wait_stream(main)                      # fork: the token ids are ready
            from . import ops
            with torch.cuda.stream(self._text_stream), torch.no_grad(), ops.workspace_lane(1):     # own scratch: concurrent
                text_job = self.student.get_text_features(input_ids=tokens, precision=text_precision).float()
        student_image = early_student_image if early_student_image is not None else \
            self.student.get_image_features(pixel_values=images, precision=self.student_precision).float()
        loss_image = self.cosine_distillation_loss(student_image, teacher_image)
        if shared_sentence is not None:
            student_text = shared_sentence.float()
        elif text_job is not None:
            main = torch.cuda.current_stream(dev)
            main.wait_stream(self._text_stream)                      # join
            text_job.record_stream(main)
            student_text = text_job
        elif text_frozen:
            with torch.no_grad():          # frozen text tower (of a bf16 student: bf16 GEMM inputs there as well)
                student_text = self.student.get_text_features(input_ids=tokens, precision=text_precision).float()
        else:
            student_text = self.student.get_text_features(input_ids=tokens).float()
        if teacher_text is None:
            with torch.no_grad():
                if self.teacher.shares_text_tower_with(self.student):
                    # frozen student text tower == teacher text tower: one forward serves both
                    teacher_text = student_text.detach()
                elif ran_teacher and self.teacher.last_sentence_embedding is not None:
                    # a separate teacher tower has just encoded these captions for its token embeddings: the sentence
                    # target (aggregate_text, :605-608) is row first-EOS of that very pass — no second teacher forward
                    teacher_text = self.teacher.last_sentence_embedding.float()
                else:
                    teacher_text = self.teacher.text_tokenizer.aggregate_text_ids(tokens).float()
                if self.teacher_bridge is not None:
                    teacher_text = self.teacher_bridge(teacher_text)
        loss_text = self.cosine_distillation_loss(student_text, teacher_text)
        contrastive = self.compute_contrastive_loss(student_image, student_text)
        if self.process_group is None:
            loss = loss_image + loss_text + 1.0 * contrastive
        else:
            # data parallel: return this rank's SHARE of the global loss (sum over ranks = the single-process value;
            # `contrastive` already is a share), so SUM-reducing gradients over ranks is exact (dist.py)
            from .dist import local_loss_for_backward
            import torch.distributed as tdist
            loss = local_loss_for_backward(loss_image, loss_text, contrastive, tdist.get_world_size(self.process_group))
        self.last_losses = {"loss_image": loss_image.detach(), "loss_text": loss_text.detach(),
                            "loss_contrastive": contrastive.detach()}
        self.log(log_name, loss.detach(), prog_bar=True, batch_size=getattr(self.hparams, bs_field, None))
        return loss

    def training_step(self, batch, batch_idx: int = 0):
        return self._step(batch, "train_loss", "train_batch_size")

    def validation_step(self, batch, batch_idx: int = 0):
        return self._step(batch, "val_loss", "eval_batch_size")

    def configure_optimizers(self):
        """AdamW over self.parameters() + linear warmup/decay (:679-682; HF get_linear_schedule_with_warmup)."""
        lr = getattr(self.hparams, "learning_rate", 2e-5)
        warm = getattr(self.hparams, "warmup_steps", 0)
        total = getattr(self.hparams, "total_steps", 1000)
        params = [p for p in self.parameters() if p.requires_grad]
        if params and params[0].is_cuda:
            from .optim import FusedAdamW
            optimizer = FusedAdamW(params, lr=lr)
        else:           # CPU: checkpoint handling / tests only; no kernels run there
            optimizer = torch.optim.AdamW(params, lr=lr)

        def lr_lambda(step: int):
            if step < warm:
                return float(step) / float(max(1, warm))
            return max(0.0, float(total - step) / float(max(1, total - warm)))

        scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda)
        return [optimizer], [scheduler]

    # ------------------------------------------------------------------ loaders (:685-693)
    def _dataset(self, json_file, cache_filename, decode_only=False):
        from .data import MultiModalDataset
        return MultiModalDataset(json_file, self.preprocess, cache_dir=getattr(self.hparams, "cache_dir", "./cache"),
                                 cache_filename=cache_filename, detector=getattr(self, "detector", None),
                                 decode_only=decode_only)

    def train_dataloader(self):
        """N3: the reference batches the TRAINING set with eval_batch_size (:687); kept.  Defaults are the reference's
        (`num_workers=0`, host preprocessing).  `hparams.gpu_preprocess=True` makes the host only DECODE (in
        `hparams.num_workers` worker processes) and runs resize / normalise / region crops on the GPU: at ~3900 img/s
        per GPU the step outruns a single-process PIL pipeline by an order of magnitude."""
        from torch.utils.data import DataLoader
        from .data import GpuBatches, MultiModalDataset, identity_collate
        cache = getattr(self.hparams, "train_cache_filename", "train_precache.pkl")
        workers = int(getattr(self.hparams, "num_workers", 0))
        decode_only = bool(getattr(self.hparams, "gpu_preprocess", False))
        ds = self._dataset(self.hparams.train_file, cache, decode_only=decode_only)
        # data parallel: every rank LOADS only its share (a DistributedSampler over the items, as Lightning installs one; the
        # Trainer does not shard such a loader again), so N ranks do not decode the epoch N times
        sampler = self._rank_sampler(ds, shuffle=True)
        kw = dict(batch_size=self.hparams.eval_batch_size, num_workers=workers, shuffle=sampler is None, sampler=sampler)
        if decode_only:
            loader = DataLoader(ds, collate_fn=identity_collate, persistent_workers=workers > 0, **kw)
            out = GpuBatches(loader, self.device, self.student.config.vision.image_size)
        else:
            out = DataLoader(ds, pin_memory=True, collate_fn=MultiModalDataset.custom_collate_fn, **kw)
        out.rank_sharded = sampler is not None
        return out

    def _rank_sampler(self, ds, shuffle: bool):
        if self.process_group is None:
            return None
        import torch.distributed as tdist
        from torch.utils.data.distributed import DistributedSampler
        return DistributedSampler(ds, num_replicas=tdist.get_world_size(self.process_group),
                                  rank=tdist.get_rank(self.process_group), shuffle=shuffle, drop_last=True)

    def val_dataloader(self):
        from torch.utils.data import DataLoader
        from .data import MultiModalDataset
        if not getattr(self.hparams, "val_file", None):
            return None
        ds = self._dataset(self.hparams.val_file, getattr(self.hparams, "val_cache_filename", "val_precache.pkl"))
        sampler = self._rank_sampler(ds, shuffle=False)
        out = DataLoader(ds, batch_size=self.hparams.eval_batch_size, num_workers=0, pin_memory=False, sampler=sampler,
                         collate_fn=MultiModalDataset.custom_collate_fn)
        out.rank_sharded = sampler is not None
        return out

    @staticmethod
    def add_model_specific_args(parent_parser: argparse.ArgumentParser) -> argparse.ArgumentParser:
        """:711-721 — same flags, same defaults, same return value (the parent parser)."""
        parser = parent_parser.add_argument_group("CLIPImageDistillation")
        parser.add_argument("--train_file", type=str, required=True, help="Path to the training JSON file.")
        parser.add_argument("--val_file", type=str, required=False, default=None, help="Path to the validation JSON file.")
        parser.add_argument("--train_batch_size", type=int, default=32, help="Training batch size.")
        parser.add_argument("--eval_batch_size", type=int, default=32, help="Evaluation batch size.")
        parser.add_argument("--learning_rate", type=float, default=2e-5, help="Learning rate.")
        parser.add_argument("--warmup_steps", type=int, default=0, help="Number of warmup steps.")
        parser.add_argument("--total_steps", type=int, default=1000, help="Total training steps.")
        return parent_parser
