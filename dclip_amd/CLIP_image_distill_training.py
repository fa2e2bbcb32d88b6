"""Student launcher — `main(args)` as in training/CLIP_image_distill_training.py:20-45, with the Lightning pieces it
uses restated in lightning_lite.py.  Models are loaded from a LOCAL path (`--clip_path`), never by hub name.

`--devices N` (the reference hard-codes `devices=1`, :39) trains data-parallel on N GPUs of one node: run as a script
it starts its N ranks itself (`python -m torch.distributed.run --nproc-per-node N` as a CHILD process, before anything
has touched the GPU); under torch.distributed.run it joins the rendezvous it is given."""
from __future__ import annotations

import argparse
import os

import torch

from .CLIP_image_distillation import CLIPImageDistillation
from .lightning_lite import Trainer


def main(args, clip_model=None, clip_preprocess=None, train_batches=None, val_batches=None, **module_kwargs):
    devices = int(getattr(args, "devices", 1) or 1)
    if devices > 1 and torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
        device = torch.device("cuda", torch.cuda.current_device())
    else:
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if clip_model is None:
        if not getattr(args, "clip_path", None) or not os.path.isdir(args.clip_path):
            raise SystemExit("--clip_path must name a local directory with HF CLIP weights (nothing is downloaded)")
        from transformers import CLIPModel, CLIPProcessor
        clip_model = CLIPModel.from_pretrained(args.clip_path, local_files_only=True).to(device)
        clip_preprocess = CLIPProcessor.from_pretrained(args.clip_path, local_files_only=True)
    model = CLIPImageDistillation(args, clip_model, clip_preprocess, **module_kwargs).to(device)
    trainer = Trainer(max_epochs=args.phase1_epochs, accelerator="gpu", devices=devices, precision=32,
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
    parser.add_argument("--devices", type=int, default=1, help="GPUs of this node (one process per GPU, RCCL)")
    return parser


def launch_ranks(devices: int, module: str, argv) -> int:
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node devices -m module argv...` as a child process; returns
    its exit code.  Called before this process has touched the GPU; never an exec."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // devices)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={devices}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", module] + list(argv)
    return subprocess.run(cmd, env=env).returncode


if __name__ == "__main__":
    import sys
    _args = build_parser().parse_args()
    if _args.devices > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(_args.devices, "dclip_amd.CLIP_image_distill_training", sys.argv[1:]))
    main(_args)
