"""`train_contrastive_teacher.main(args)` — the teacher trainer entry point, same arguments and the same checkpoint
files as training/train_contrastive_teacher.py, with the step on the HIP kernels.

What is kept (training/train_contrastive_teacher.py):
  * seed 42 (:99); everything frozen except parameters whose name contains cross_attn|attention|proj|fusion|final
    (:126-134) — on the teacher that is exactly the 12 tensors of `cross_modal_attention.*`;
  * Adam(lr=args.learning_rate) over the trainable set (:245-248); `--gradient_accumulation` is parsed and, as in
    the reference, not used (:435);
  * per batch: meta-teacher image embedding vs CLIP sentence embedding under the symmetric InfoNCE with
    temperature 0.05 (:251-261, :340-362);
  * `torch.save(teacher.state_dict(), f"{stem}_epoch{N}_val{loss:.4f}.pth")` every epoch, best → `output_path`,
    `output_path + ".interrupt.pth"` / `".error.pth"` on Ctrl-C / exception (:394-420).

Data parallel ("the teacher trainer shards identically"; the reference is single-GPU): under
`python -m torch.distributed.run --nproc-per-node N -m dclip_amd.train_contrastive_teacher …` (or with `process_group=`)
batch i goes to rank i % N, the InfoNCE uses global negatives, the 12 gradient tensors (8.4 MB) are SUM-reduced in one
bucket, rank 0 writes the checkpoints; N ranks perform the update of one process on the concatenated batches.

What differs: models are never fetched by name — `args.clip_path` names a LOCAL directory with HF CLIP weights (or
`teacher=` is passed in); batches may be the reference's `(images, captions, paths, boxes)` tuples (needs a
tokenizer) or tensor dicts `{regions, input_ids[, region_counts]}`; the DBM KNN cache (:19-95) feeds only the
out-of-scope KNN tokenizer and is not built.
"""
from __future__ import annotations

import argparse
import json
import os
import random
from typing import Iterable, Optional

import numpy as np
import torch

from . import functional
from .patch_text_aggregation import PatchTextAggregation


def seed_everything(seed: int = 42):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def compute_contrastive_loss(image_embeddings, text_embeddings, temperature=0.05, group=None):
    """:251-261 (textually the same function as CLIP_image_distillation.py:532-562)."""
    return functional.contrastive_loss(image_embeddings, text_embeddings, temperature, group)


class JsonPairDataset:
    """`[{image_path, captions|caption, boxes?}]` (json_creation/big_teacher_data.py:86-91) → the reference's batch
    tuples; boxes come from the JSON (or an empty list): there is no detector here."""

    def __init__(self, json_file: str, batch_size: int, shuffle: bool, seed: int = 42, rank: int = 0, world: int = 1):
        """`rank` / `world`: data parallel — every rank draws the SAME epoch order (same seed) and yields batch i only when
        i % world == rank; a trailing group that does not cover every rank is dropped (`rank_sharded` tells the trainer not
        to deal the stream out again)."""
        with open(json_file, "r", encoding="utf-8") as f:
            self.data = json.load(f)
        self.batch_size, self.shuffle = batch_size, shuffle
        self.rng = random.Random(seed)
        self.rank, self.world = rank, world
        self.rank_sharded = world > 1

    def __len__(self):
        n = (len(self.data) + self.batch_size - 1) // self.batch_size
        return n if self.world == 1 else n // self.world

    def __iter__(self):
        order = list(range(len(self.data)))
        if self.shuffle:
            self.rng.shuffle(order)
        nb = (len(order) + self.batch_size - 1) // self.batch_size
        full = nb - nb % self.world                     # batches that belong to complete rank groups
        for bi, i in enumerate(range(0, len(order), self.batch_size)):
            if self.world > 1 and (bi >= full or bi % self.world != self.rank):
                continue
            items = [self.data[j] for j in order[i:i + self.batch_size]]
            caps = []
            for it in items:
                c = it.get("captions", it.get("caption", ""))
                caps.append(c[0] if isinstance(c, list) and c else (c if isinstance(c, str) else ""))
            paths = [it.get("image_path", "") for it in items]
            boxes = [[(tuple(b[0]), float(b[1])) if isinstance(b[0], (list, tuple)) else (tuple(b[:4]), 1.0)
                      for b in it.get("boxes", [])] for it in items]
            yield None, caps, paths, boxes


def _embeddings(teacher: PatchTextAggregation, batch):
    if isinstance(batch, dict):
        dev = teacher.device
        ids = batch["input_ids"].to(dev)
        img = teacher.compute_global_embedding_tensors(batch["regions"].to(dev), ids, batch.get("region_counts"),
                                                       batch.get("max_tokens"))
        txt = teacher.text_tokenizer.aggregate_text_ids(ids)
    else:
        _, captions, image_paths, weighted_boxes_batch = batch
        img = teacher.compute_global_embedding_batch(image_paths, captions, weighted_boxes_batch)
        # the reference calls aggregate_text once per caption (:346): the sentence embeddings of these very captions
        # are row first-EOS of the token-level pass the call above has just made — no further text forward
        txt = teacher.last_sentence_embedding
    return img, txt


def build_teacher(args, device) -> PatchTextAggregation:
    from .clip_model import from_hf_state_dict
    from .config import from_hf
    path = getattr(args, "clip_path", None)
    if not path or not os.path.isdir(path):
        raise SystemExit("--clip_path must name a local directory with HF CLIP weights (nothing is downloaded by name)")
    from transformers import CLIPModel, CLIPTokenizer
    hf = CLIPModel.from_pretrained(path, local_files_only=True)
    clip = from_hf_state_dict(from_hf(hf.config), hf.state_dict(), device=device)
    tok = CLIPTokenizer.from_pretrained(path, local_files_only=True)
    return PatchTextAggregation(embed_dim=clip.config.projection_dim, num_heads=clip.config.projection_dim // 64,
                                similarity_threshold=0.85, projection_model_path="", faiss_index_path="",
                                embeddings_json_path="", clip_model=clip, tokenizer=tok).to(device)


def main(args, teacher: Optional[PatchTextAggregation] = None, train_batches: Optional[Iterable] = None,
         val_batches: Optional[Iterable] = None, process_group=None):
    seed_everything(42)
    from . import dist as ddist
    group = process_group
    if group is None and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        group = ddist.init_from_env(os.environ.get("DCLIP_DIST_BACKEND"))
    world, rank = 1, 0
    if group is not None:
        import torch.distributed as tdist
        world, rank = tdist.get_world_size(group), tdist.get_rank(group)
    if teacher is not None:
        device = teacher.device
    elif torch.cuda.is_available():
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    else:
        device = torch.device("cpu")
    if teacher is None:
        teacher = build_teacher(args, device)

    for param in teacher.parameters():
        param.requires_grad = False
    for name, param in teacher.named_parameters():
        if any(key in name for key in ["cross_attn", "attention", "proj", "fusion", "final"]):
            param.requires_grad = True
    trainable = [p for p in teacher.parameters() if p.requires_grad]
    if rank == 0:
        print(f"Training {sum(p.numel() for p in trainable):,} parameters out of "
              f"{sum(p.numel() for p in teacher.parameters()):,}")

    if train_batches is None:
        train_batches = JsonPairDataset(args.train_file, args.batch_size, shuffle=True, rank=rank, world=world)
    if val_batches is None and getattr(args, "val_file", None) and os.path.exists(args.val_file):
        val_batches = JsonPairDataset(args.val_file, args.batch_size, shuffle=False, rank=rank, world=world)

    if trainable and trainable[0].is_cuda:
        from .optim import FusedAdam
        optimizer = FusedAdam(trainable, lr=args.learning_rate)      # optim.Adam(trainable_params, lr) (:245-248)
    else:                                                            # CPU: argument / checkpoint plumbing only
        optimizer = torch.optim.Adam(trainable, lr=args.learning_rate)
    sync = ddist.GradSync(trainable, group) if group is not None else None      # 8.4 MB: one bucket

    def sharded(batches):
        if group is None or getattr(batches, "rank_sharded", False):
            return batches
        return ddist.shard_batches(batches, rank, world)

    def global_value(share_sum):
        """Sum over ranks of the per-rank loss shares accumulated on the device (one small all-reduce per epoch)."""
        if group is None or not isinstance(share_sum, torch.Tensor):
            return float(share_sum)
        v = share_sum.detach().clone()
        tdist.all_reduce(v, group=group)
        return float(v)

    def validate():
        teacher.eval()
        total, n = 0.0, 0
        with torch.no_grad():
            for batch in sharded(val_batches or []):
                img, txt = _embeddings(teacher, batch)
                v = compute_contrastive_loss(img, txt, group=group)
                total = total + (v.detach() if group is not None else float(v))
                n += 1
        return {"combined": global_value(total) / max(1, n)}

    out_dir = os.path.dirname(args.output_path)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    best_val_loss = float("inf")
    history = []
    try:
        for epoch in range(args.epochs):
            teacher.train()
            epoch_loss, n = 0.0, 0
            for batch_idx, batch in enumerate(sharded(train_batches)):
                optimizer.zero_grad()
                img_emb, txt_emb = _embeddings(teacher, batch)
                loss = compute_contrastive_loss(img_emb, txt_emb, group=group)     # N ranks: this rank's share
                loss.backward()
                if sync is not None:
                    sync.finish()
                optimizer.step()
                epoch_loss = epoch_loss + (loss.detach() if group is not None else float(loss.detach()))
                n += 1
            avg_loss = global_value(epoch_loss) / max(1, n)
            val_losses = validate()
            history.append((avg_loss, val_losses["combined"]))
            if rank == 0:
                print(f"Epoch {epoch + 1}/{args.epochs}  train {avg_loss:.4f}  val {val_losses['combined']:.4f}")
                epoch_save_path = f"{args.output_path.rsplit('.', 1)[0]}_epoch{epoch + 1}_val{val_losses['combined']:.4f}.pth"
                torch.save(teacher.state_dict(), epoch_save_path)
            if val_losses["combined"] < best_val_loss:
                best_val_loss = val_losses["combined"]
                if rank == 0:
                    torch.save(teacher.state_dict(), args.output_path)
            if group is not None:
                tdist.barrier(group=group)
    except KeyboardInterrupt:
        if rank == 0:
            torch.save(teacher.state_dict(), args.output_path + ".interrupt.pth")
    except Exception:
        if rank == 0:
            torch.save(teacher.state_dict(), args.output_path + ".error.pth")
        raise
    return {"best_val_loss": best_val_loss, "history": history}


def build_parser() -> argparse.ArgumentParser:
    """:430-441 — same flags and defaults (+ --clip_path, because nothing is fetched by name)."""
    parser = argparse.ArgumentParser(description="Train Contrastive-Aware Teacher with Gradient Accumulation")
    parser.add_argument("--train_file", type=str, required=True, help="Path to training JSON file")
    parser.add_argument("--val_file", type=str, default=None, help="Path to validation JSON file (optional)")
    parser.add_argument("--batch_size", type=int, default=64, help="Batch size per accumulation step")
    parser.add_argument("--gradient_accumulation", type=int, default=8, help="Number of gradient accumulation steps")
    parser.add_argument("--learning_rate", type=float, default=1e-5, help="Learning rate")
    parser.add_argument("--epochs", type=int, default=5, help="Number of epochs")
    parser.add_argument("--output_path", type=str, default="./teacher_contrastive/contrastive_teacher_ViT-16.pth",
                        help="Path to save the trained teacher model")
    parser.add_argument("--clip_path", type=str, default=None, help="LOCAL directory with HF CLIP weights + tokenizer")
    parser.add_argument("--devices", type=int, default=1, help="GPUs of this node (one process per GPU, RCCL)")
    return parser


if __name__ == "__main__":
    import sys
    _args = build_parser().parse_args()
    if _args.devices > 1 and "WORLD_SIZE" not in os.environ:
        from .CLIP_image_distill_training import launch_ranks
        raise SystemExit(launch_ranks(_args.devices, "dclip_amd.train_contrastive_teacher", sys.argv[1:]))
    main(_args)
