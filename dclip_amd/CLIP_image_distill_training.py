"""Student launcher — `main(args)` as in training/CLIP_image_distill_training.py:20-45, with the Lightning pieces it
uses restated in lightning_lite.py.  Models are loaded from a LOCAL path (`--clip_path`), never by hub name."""
from __future__ import annotations

import argparse
import os

import torch

from .CLIP_image_distillation import CLIPImageDistillation
from .lightning_lite import Trainer


def main(args, clip_model=None, clip_preprocess=None, train_batches=None, val_batches=None, **module_kwargs):
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if clip_model is None:
        if not getattr(args, "clip_path", None) or not os.path.isdir(args.clip_path):
            raise SystemExit("--clip_path must name a local directory with HF CLIP weights (nothing is downloaded)")
        from transformers import CLIPModel, CLIPProcessor
        clip_model = CLIPModel.from_pretrained(args.clip_path, local_files_only=True).to(device)
        clip_preprocess = CLIPProcessor.from_pretrained(args.clip_path, local_files_only=True)
    model = CLIPImageDistillation(args, clip_model, clip_preprocess, **module_kwargs).to(device)
    trainer = Trainer(max_epochs=args.phase1_epochs, accelerator="gpu", devices=1, precision=32,
                      gradient_clip_val=0.5, accumulate_grad_batches=4, checkpoint_dir=args.checkpoint_dir,
                      save_top_k=10, max_steps=getattr(args, "max_steps", None))
    trainer.fit(model, train_batches, val_batches)
    return model, trainer


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser()
    parser = CLIPImageDistillation.add_model_specific_args(parser)
    parser.add_argument("--checkpoint_dir", type=str, default="./checkpoints")          # :50
    parser.add_argument("--phase1_epochs", type=int, default=2)                          # :51
    parser.add_argument("--clip_path", type=str, default=None)
    return parser


if __name__ == "__main__":
    main(build_parser().parse_args())
