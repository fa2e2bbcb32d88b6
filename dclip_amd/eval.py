"""Checkpoint consumers on the HIP kernels: text<->image retrieval (Recall@1/5/10 + MAP) and zero-shot classification.

Mirrors eval_scripts/flickr30k_eval.py:16-88 (`calculate_retrieval_metrics`, same return structure) and
eval_scripts/test_zero_shot_ImageNet.py:37-125 (`evaluate_zero_shot`: prompts "a photo of a {name}", CLIP
mean/std normalisation, top-1 / top-5).  The reference builds the full caption x image matrix chunk by chunk and
argsorts each row and column; here the rank of a ground truth is counted directly on MFMA similarity tiles
(dclip_rank_count), so nothing of size [captions, images] is materialised.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Sequence

import torch

from . import ops

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # eval_scripts/test_zero_shot_ImageNet.py:69-70
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _normalised(x: torch.Tensor) -> torch.Tensor:
    xhat, _ = ops.normalize_rows_fwd(x.detach().float().contiguous())
    return xhat


def retrieval_ranks(image_embeddings: torch.Tensor, caption_embeddings: torch.Tensor, caption_to_image: torch.Tensor):
    """t2i rank of every caption's image among all images, and i2t rank of every caption among all captions for its
    own image (0 = best).  caption_to_image [Nc] int32 indexes rows of image_embeddings."""
    img = _normalised(image_embeddings)
    cap = _normalised(caption_embeddings)
    gt = caption_to_image.to(img.device).to(torch.int32).contiguous()
    s_gt = ops.rowdot_gather(cap, img, gt)                       # <caption_c, image(c)>
    t2i = ops.rank_count(cap, img, s_gt, gt)                     # images scoring higher for caption c
    own = img[gt.long()].contiguous()                            # image(c) as the query row
    i2t = ops.rank_count(own, cap, s_gt, None)                   # captions scoring higher for image(c); self = c
    return t2i, i2t


def calculate_retrieval_metrics(image_embeddings, caption_embeddings, image_ids: Sequence, caption_image_ids: Sequence
                                ) -> Dict[str, Dict[str, float]]:
    """Same metrics dict as the reference function; takes the embeddings instead of the similarity matrix."""
    index = {iid: i for i, iid in enumerate(image_ids)}
    gt = torch.tensor([index[c] for c in caption_image_ids], dtype=torch.int32)
    t2i, i2t = retrieval_ranks(image_embeddings, caption_embeddings, gt)
    t2i_ranks = t2i.cpu().tolist()
    per_cap = i2t.cpu().tolist()
    best = defaultdict(lambda: 1 << 30)
    for c, iid in enumerate(caption_image_ids):                 # best rank among an image's ground-truth captions (:62)
        best[iid] = min(best[iid], per_cap[c])
    i2t_ranks = [best[iid] for iid in image_ids if iid in best]

    def recall_at_k(ranks, k):
        return len([r for r in ranks if r < k]) / len(ranks)

    def mean_ap(ranks):
        return float(sum(1.0 / (r + 1) for r in ranks) / len(ranks))

    return {d: {"R@1": recall_at_k(r, 1), "R@5": recall_at_k(r, 5), "R@10": recall_at_k(r, 10), "MAP": mean_ap(r)}
            for d, r in (("t2i", t2i_ranks), ("i2t", i2t_ranks))}


@torch.no_grad()
def evaluate_retrieval(model, pixel_batches, id_batches, image_ids, caption_image_ids):
    """`model` is a CLIPImageDistillation (forward(image=)/forward(text=)) or a HipCLIPModel."""
    enc_i = (lambda x: model(image=x)) if hasattr(model, "student") else (lambda x: model.get_image_features(pixel_values=x))
    enc_t = (lambda x: model(text=x)) if hasattr(model, "student") else (lambda x: model.get_text_features(input_ids=x))
    img = torch.cat([enc_i(b) for b in pixel_batches])
    cap = torch.cat([enc_t(b) for b in id_batches])
    return calculate_retrieval_metrics(img, cap, image_ids, caption_image_ids)


def zero_shot_ranks(image_features: torch.Tensor, class_text_features: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """Rank of the true class for every image (0 = top-1 correct); `100.0 *` of the reference does not change ranks."""
    img = _normalised(image_features)
    txt = _normalised(class_text_features)
    lab = labels.to(img.device).to(torch.int32).contiguous()
    return ops.rank_count(img, txt, ops.rowdot_gather(img, txt, lab), lab)


@torch.no_grad()
def evaluate_zero_shot(clip_model, image_batches, label_batches, class_input_ids, normalize_images: bool = True):
    """top-1 / top-5 accuracy; images in [0,1] are normalised with CLIP mean/std as the reference does (:69-71)."""
    dev = next(clip_model.parameters()).device
    text = clip_model.get_text_features(input_ids=class_input_ids.to(dev))
    mean = torch.tensor(CLIP_MEAN, device=dev).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, device=dev).view(1, 3, 1, 1)
    top1 = top5 = total = 0
    for images, labels in zip(image_batches, label_batches):
        images = images.to(dev)
        if normalize_images:
            images = (images - mean) / std
        ranks = zero_shot_ranks(clip_model.get_image_features(pixel_values=images), text, labels)
        top1 += int((ranks == 0).sum())
        top5 += int((ranks < 5).sum())
        total += len(labels)
    return {"top1": top1 / max(total, 1), "top5": top5 / max(total, 1)}
