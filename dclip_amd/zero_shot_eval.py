"""Zero-shot classification on the HIP kernels: the script-level flows of eval_scripts/test_zero_shot_ImageNet.py and
eval_scripts/CIFAR_zeroshot.py (checkpoint consumers).

    evaluate_zero_shot(model_name, model, processor, dataloader, classnames, dataset_name=None)
        same call as the reference's function (test_zero_shot_ImageNet.py:37, CIFAR_zeroshot.py:47): prompts
        "a photo of a {name}" (", a type of object" appended for CIFAR, CIFAR_zeroshot.py:52-55) tokenised by
        `processor(text=prompts, return_tensors="pt", padding=True)`; "base" encodes with `model`, anything else with
        `model.student`; images come from the loader in [0,1] and get CLIP's mean / std (:69-71); returns
        {"top1", "top5"}.
    ImageFolderDataset(root, size=224)   torchvision's ImageFolder + Resize(size) + CenterCrop(size) + ToTensor()
        (:141-151; torchvision is not needed: PIL does the bilinear shortest-edge resize it would call)
    main(argv)   --data_root --classnames --clip_path --checkpoint --dataset {imagenet,cifar10,cifar100} --batch_size
                 --max_images --results  (the reference hard-codes all of them, downloads CIFAR and fetches models by name)

What differs: batches instead of batch_size=1; the true class's rank is counted on MFMA similarity tiles
(`eval.zero_shot_ranks`: `100.0 *` and `topk(5)` only decide whether fewer than 1 / 5 classes score higher); models,
tokenizer and data come from LOCAL paths.  Image decoding and the torchvision transform stay on the host.
"""
from __future__ import annotations

import argparse
import os
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from . import eval as E

IMAGE_EXT = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


class ImageFolderDataset:
    """`datasets.ImageFolder(root, transform=Compose([Resize(size), CenterCrop(size), ToTensor()]))`: one sub-directory per
    class (sorted names -> indices 0..C-1), images sorted inside; items are ([3,size,size] float in [0,1], label)."""

    def __init__(self, root: str, size: int = 224, max_images: Optional[int] = None):
        self.root, self.size = root, size
        self.classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
        if not self.classes:
            raise FileNotFoundError(f"no class folders under {root}")
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples = []
        for c in self.classes:
            for dirpath, _dirs, files in sorted(os.walk(os.path.join(root, c))):
                for f in sorted(files):
                    if f.lower().endswith(IMAGE_EXT):
                        self.samples.append((os.path.join(dirpath, f), self.class_to_idx[c]))
        if max_images:
            self.samples = self.samples[:max_images]

    def __len__(self):
        return len(self.samples)

    def transform(self, img) -> torch.Tensor:
        from PIL import Image
        img = img.convert("RGB")
        w, h = img.size
        s = self.size
        if w <= h:                                        # torchvision Resize(int): shortest edge -> s, long edge int(s*long/short)
            nw, nh = s, int(s * h / w)
        else:
            nw, nh = int(s * w / h), s
        if (nw, nh) != (w, h):
            img = img.resize((nw, nh), Image.BILINEAR)
        left, top = int(round((nw - s) / 2.0)), int(round((nh - s) / 2.0))          # CenterCrop
        img = img.crop((left, top, left + s, top + s))
        arr = np.asarray(img, dtype=np.float32) / 255.0                              # ToTensor
        return torch.from_numpy(arr).permute(2, 0, 1).contiguous()

    def __getitem__(self, i):
        from PIL import Image
        path, label = self.samples[i]
        with Image.open(path) as im:
            return self.transform(im), label


def batches(dataset, batch_size: int):
    """DataLoader(dataset, batch_size, shuffle=False) without worker processes: (images [b,3,S,S], labels [b])."""
    for i in range(0, len(dataset), batch_size):
        items = [dataset[j] for j in range(i, min(i + batch_size, len(dataset)))]
        yield torch.stack([x for x, _ in items]), torch.tensor([y for _, y in items], dtype=torch.int64)


def make_prompts(classnames: Sequence[str], dataset_name: Optional[str] = None) -> List[str]:
    if dataset_name and "cifar" in dataset_name.lower():
        return [f"a photo of a {name}, a type of object" for name in classnames]      # CIFAR_zeroshot.py:52-53
    return [f"a photo of a {name}" for name in classnames]                            # test_zero_shot_ImageNet.py:42


@torch.no_grad()
def evaluate_zero_shot(model_name: str, model, processor: Callable, dataloader, classnames: Sequence[str],
                       dataset_name: Optional[str] = None):
    """Evaluate a single model for zero-shot classification (reference signature; see the module docstring)."""
    print(f"\nEvaluating {model_name}" + (f" on {dataset_name}" if dataset_name else "") + "...")
    clip = model if model_name == "base" or not hasattr(model, "student") else model.student
    dev = next(clip.parameters()).device
    tok = processor(text=make_prompts(classnames, dataset_name), return_tensors="pt", padding=True)
    ids = tok["input_ids"] if isinstance(tok, dict) or hasattr(tok, "keys") else tok
    text_features = clip.get_text_features(input_ids=ids.to(dev))
    mean = torch.tensor(E.CLIP_MEAN, device=dev).view(1, 3, 1, 1)
    std = torch.tensor(E.CLIP_STD, device=dev).view(1, 3, 1, 1)
    correct_top1 = correct_top5 = total = 0
    for images, labels in dataloader:
        images = (images.to(dev).float() - mean) / std
        ranks = E.zero_shot_ranks(clip.get_image_features(pixel_values=images), text_features, labels)
        correct_top1 += int((ranks == 0).sum())
        correct_top5 += int((ranks < 5).sum())
        total += len(labels)
    top1 = correct_top1 / total if total > 0 else 0
    top5 = correct_top5 / total if total > 0 else 0
    print(f"{model_name} Top-1: {top1:.4f}, Top-5: {top5:.4f}")
    return {"top1": top1, "top5": top5}


def main(argv=None, clip_model=None, processor=None):
    ap = argparse.ArgumentParser(description="Zero-shot evaluation (ImageNet / CIFAR folder layouts) on dclip_amd")
    ap.add_argument("--data_root", required=True, help="ImageFolder layout: one sub-directory per class")
    ap.add_argument("--classnames", default=None, help="text file, one class name per line (default: the folder names)")
    ap.add_argument("--dataset", default="imagenet", choices=["imagenet", "cifar10", "cifar100"])
    ap.add_argument("--clip_path", default=None, help="LOCAL directory with HF CLIP weights + processor")
    ap.add_argument("--checkpoint", default=None, help="student .ckpt (Lightning layout); evaluates 'custom' as well")
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--max_images", type=int, default=0)
    ap.add_argument("--results", default=None, help="results text file (default <dataset>_zero_shot_results.txt)")
    args = ap.parse_args(argv)
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    from .CLIP_image_distillation import CLIPImageDistillation, _as_hip_model
    if clip_model is None:
        if not args.clip_path or not os.path.isdir(args.clip_path):
            raise SystemExit("--clip_path must name a local directory with HF CLIP weights (nothing is downloaded by name)")
        from transformers import CLIPModel, CLIPProcessor
        clip_model = CLIPModel.from_pretrained(args.clip_path, local_files_only=True).to(device).eval()
        processor = CLIPProcessor.from_pretrained(args.clip_path, local_files_only=True)
    base = _as_hip_model(clip_model).to(device)
    ds = ImageFolderDataset(args.data_root, base.config.vision.image_size, args.max_images or None)
    if args.classnames:
        with open(args.classnames, "r") as f:
            classnames = [line.strip() for line in f.readlines()]
    else:
        classnames = list(ds.classes)
    print(f"Dataset contains {len(ds)} images with {len(ds.classes)} classes; {len(classnames)} class names")
    results = {"base": evaluate_zero_shot("base", base, processor, batches(ds, args.batch_size), classnames, args.dataset)}
    if args.checkpoint:
        import copy
        custom = CLIPImageDistillation.load_from_checkpoint(args.checkpoint, map_location=device, clip_model=copy.deepcopy(base),
                                                            clip_preprocess=processor, strict=False).to(device).eval()
        results["custom"] = evaluate_zero_shot("custom", custom, processor, batches(ds, args.batch_size), classnames,
                                               args.dataset)
    out = args.results or f"{args.dataset}_zero_shot_results.txt"
    with open(out, "w") as f:
        f.write(f"Zero-Shot {args.dataset} Results\n")
        for name, r in results.items():
            f.write(f"{name} Top-1: {r['top1']:.4f}\n{name} Top-5: {r['top5']:.4f}\n\n")
        if "custom" in results and results["base"]["top1"] > 0:
            rel = (results["custom"]["top1"] - results["base"]["top1"]) / results["base"]["top1"] * 100
            f.write(f"Relative Top-1 improvement: {rel:+.2f}%\n")
    return results


if __name__ == "__main__":
    main()
