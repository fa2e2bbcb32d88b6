"""Data formats on the input side of the distillation step: the dataset the reference's
training loop iterates, its detection-box caches, and the student image preprocessing.

    MultiModalDataset(json_file, clip_preprocess, cache_dir, use_batch_cache, cache_filename)
        -> (pixel_values[3,S,S] f32, caption str, image_path str, weighted_boxes [((x1,y1,x2,y2) int, conf float)])
                                                            (training/CLIP_image_distillation.py:78-405)
    MultiModalDataset.custom_collate_fn(batch) -> (pixel_values[B,3,S,S], captions, image_paths, boxes)   (:407-434)
    load_or_compute_yolo(image_path, clip_preprocess, cache_dir)    per-image `<basename>.pkl` cache      (:43-75)

On-disk layouts kept (so caches written by the reference keep working):
  * `<cache_dir>/<cache_filename>`: ONE pickle of `{image_path: [((x1,y1,x2,y2), conf), ...]}` (protocol 4, :299);
  * files of 1 GiB and more are converted once to a `dbm` database `<stem>_mmap.db` whose values are the pickled
    per-image lists, plus `<stem>_keys.pkl` = the key list (:131-262);
  * `<cache_dir>/<basename(image_path)>.pkl`: one list per image (:45).
A box cache holds nothing but dict / list / tuple / str / int / float, so every file is read with an unpickler that
resolves NO globals: a file that names a class or function is rejected instead of executed.

What is not here: the detector.  The reference runs YOLO (a downloaded checkpoint) on a cache miss (:60-62,
:268-289); this package takes `detector=` (any callable `image_path -> boxes`) and raises on a miss without one.

Host code only (image decoding, files); the arithmetic of preprocessing is available on the GPU as well
(`GpuCollate`, kernels in csrc/crop_resize.hip) and both agree bit for bit with HF's CLIPImageProcessor.
"""
from __future__ import annotations

import dbm
import io
import json
import os
import pickle
import random
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

BIG_CACHE_BYTES = 1024 * 1024 * 1024        # :126 — larger pickles are served from dbm


# ----------------------------------------------------------------------------------------------- box caches

class _PlainUnpickler(pickle.Unpickler):
    """Containers, strings and numbers need no global lookup; anything that does is not a box cache."""

    def find_class(self, module, name):
        raise pickle.UnpicklingError(f"box cache names {module}.{name}: refusing to resolve globals")


def plain_loads(data: bytes):
    return _PlainUnpickler(io.BytesIO(data)).load()


def plain_load(path: str):
    with open(path, "rb") as f:
        return _PlainUnpickler(f).load()


def atomic_pickle_dump(obj, path: str, protocol: int = 4):
    """tmp file + rename (:64-73, :296-306)."""
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        pickle.dump(obj, f, protocol=protocol)
    os.replace(tmp, path)


def _norm_boxes(boxes) -> list:
    return [((int(b[0][0]), int(b[0][1]), int(b[0][2]), int(b[0][3])), float(b[1])) for b in boxes]


class DiskCache:
    """Read-mostly mapping over the dbm layout (:147-174): keys utf-8 paths, values pickled box lists."""

    def __init__(self, db, keys: Sequence[str]):
        self.db = db
        self.keys_list = list(keys)
        self.extra = {}                       # misses filled in during the run stay in memory (:360)

    def __getitem__(self, key):
        if key in self.extra:
            return self.extra[key]
        if isinstance(key, str) and key.encode("utf-8") in self.db:
            return plain_loads(self.db[key.encode("utf-8")])
        return None

    def __setitem__(self, key, value):
        self.extra[key] = value

    def __contains__(self, key):
        return key in self.extra or (isinstance(key, str) and key.encode("utf-8") in self.db)

    def get(self, key, default=None):
        v = self[key] if key in self else None
        return default if v is None else v

    def __len__(self):
        return len(self.keys_list) + len(self.extra)

    def __bool__(self):
        return len(self) > 0


def open_box_cache(cache_dir: str, cache_filename: str, big_bytes: int = BIG_CACHE_BYTES):
    """Strategy 1 of the reference (:113-262): dict from the pickle, or the dbm proxy for big caches (converted
    once).  Returns `{}` when nothing usable exists."""
    stem = os.path.splitext(cache_filename)[0]
    precache = os.path.join(cache_dir, cache_filename)
    mmap_file = os.path.join(cache_dir, f"{stem}_mmap.db")
    keys_file = os.path.join(cache_dir, f"{stem}_keys.pkl")
    if not os.path.exists(precache):
        return {}
    try:
        if os.path.getsize(precache) < big_bytes:
            cache = plain_load(precache)
            if not isinstance(cache, dict):
                raise pickle.UnpicklingError("box cache is not a dict")
            return cache
        try:
            db = dbm.open(mmap_file, "r")
        except dbm.error:
            db = None
        if db is None:
            cache = plain_load(precache)
            db = dbm.open(mmap_file, "c")
            for k, v in cache.items():
                db[k.encode("utf-8")] = pickle.dumps(v)
            atomic_pickle_dump(list(cache.keys()), keys_file)
            del cache
            db.close()
            db = dbm.open(mmap_file, "r")
        keys = plain_load(keys_file) if os.path.exists(keys_file) else [k.decode("utf-8") for k in db.keys()]
        return DiskCache(db, keys)
    except (pickle.UnpicklingError, EOFError, ImportError, MemoryError, AttributeError, IndexError) as e:
        print(f"Warning: Cache file error ({e}), regenerating...")
        return {}


def load_or_compute_yolo(image_path: str, clip_preprocess=None, cache_dir: str = "./cache",
                         detector: Optional[Callable] = None):
    """:43-75 — the per-image cache; `detector(image_path)` stands where the reference constructs YOLO."""
    os.makedirs(cache_dir, exist_ok=True)
    cache_file = os.path.join(cache_dir, os.path.basename(image_path) + ".pkl")
    try:
        if os.path.exists(cache_file):
            return plain_load(cache_file)
    except (pickle.UnpicklingError, EOFError, ImportError):
        print(f"Warning: Corrupt cache file detected for {os.path.basename(image_path)}, regenerating...")
        if os.path.exists(cache_file):
            os.remove(cache_file)
    if detector is None:
        raise RuntimeError(f"no cached boxes for {image_path!r} and no detector= was given (the YOLO detector is "
                           "outside the distillation step; precompute the box cache)")
    boxes = _norm_boxes(detector(image_path))
    atomic_pickle_dump(boxes, cache_file, protocol=pickle.DEFAULT_PROTOCOL)
    return boxes


# ----------------------------------------------------------------------------------------------- preprocessing

def shortest_edge_size(h: int, w: int, size: int):
    """(new_h, new_w) of HF `get_resize_output_image_size(..., default_to_square=False)`."""
    short, long = (w, h) if w <= h else (h, w)
    new_long = int(size * long / short)
    return (new_long, size) if w <= h else (size, new_long)


class ClipImagePreprocess:
    """The image half of HF `CLIPProcessor` with its default CLIP settings, needing no files: RGB -> shortest edge
    to `size` (PIL BICUBIC) -> centre crop -> float32(float64(v)/255-scale) -> (x - mean) / std -> CHW.
    Call it the way the reference calls its processor (`preprocess(images=img, text="", return_tensors="pt")`,
    :349).  `tokenizer` (an HF CLIPTokenizer from a LOCAL path) is needed only for non-empty `text`."""

    def __init__(self, size: int = 224, mean=CLIP_MEAN, std=CLIP_STD, tokenizer=None):
        self.size, self.mean, self.std, self.tokenizer = size, tuple(mean), tuple(std), tokenizer

    def decode(self, image) -> np.ndarray:
        """PIL image / HWC uint8 array -> HWC uint8 RGB array."""
        if hasattr(image, "convert"):
            image = image.convert("RGB")
        arr = np.array(image, dtype=np.uint8)          # a writable copy (torch.from_numpy wants one)
        if arr.ndim != 3 or arr.shape[2] != 3:
            raise ValueError(f"expected an RGB image, got array of shape {arr.shape}")
        return arr

    def image(self, image) -> torch.Tensor:
        from PIL import Image
        arr = self.decode(image)
        h, w = arr.shape[:2]
        nh, nw = shortest_edge_size(h, w, self.size)
        res = np.asarray(Image.fromarray(arr).resize((nw, nh), resample=Image.BICUBIC), dtype=np.uint8)
        top, left = (nh - self.size) // 2, (nw - self.size) // 2
        win = res[top:top + self.size, left:left + self.size]
        x = (win.astype(np.float64) * (1 / 255)).astype(np.float32)
        x = (x - np.array(self.mean, dtype=np.float32)) / np.array(self.std, dtype=np.float32)
        return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))

    def __call__(self, images=None, text=None, return_tensors: str = "pt", **kw):
        out = {}
        if images is not None:
            many = isinstance(images, (list, tuple))
            out["pixel_values"] = torch.stack([self.image(im) for im in (images if many else [images])])
        if text is not None and text != "" and text != [""]:
            if self.tokenizer is None:
                raise RuntimeError("caption strings need a CLIPTokenizer loaded from a local path (tokenizer=)")
            out["input_ids"] = self.tokenizer(text, return_tensors="pt", **kw)["input_ids"]
        return out


class GpuCollate:
    """Collate for `MultiModalDataset(..., decode_only=True)`: pads the decoded uint8 images into one batch, uploads
    it once and runs the student preprocessing on the GPU (bit-exact with `ClipImagePreprocess`).  The batch keeps
    the uploaded `images_u8` / `dims`, so the teacher's region crops (`CLIPPatchTokenizer.crop_boxes_gpu`) can be
    cut from the same copy instead of decoding every file a second time."""

    def __init__(self, device, size: int = 224, mean=CLIP_MEAN, std=CLIP_STD):
        self.device, self.size, self.mean, self.std = torch.device(device), size, mean, std

    def __call__(self, batch):
        from . import ops
        arrs, captions, paths, boxes = zip(*batch)
        B = len(arrs)
        hmax, wmax = max(a.shape[0] for a in arrs), max(a.shape[1] for a in arrs)
        host = torch.zeros((B, hmax, wmax, 3), dtype=torch.uint8)
        for b, a in enumerate(arrs):
            host[b, :a.shape[0], :a.shape[1]] = torch.from_numpy(np.ascontiguousarray(a))
        dims = torch.tensor([a.shape[:2] for a in arrs], dtype=torch.int32)
        images_u8, dims_d = host.to(self.device), dims.to(self.device)
        pixel_values = ops.clip_preprocess(images_u8, dims_d, self.size, self.mean, self.std)
        return {"pixel_values": pixel_values, "captions": list(captions), "image_paths": list(paths),
                "weighted_boxes": list(boxes), "images_u8": images_u8, "dims": dims_d}


def identity_collate(batch):
    """For DataLoader workers of a decode_only dataset: hand the list of samples through unchanged (workers must
    not touch the GPU; `GpuBatches` collates in the main process)."""
    return batch


class GpuBatches:
    """Iterates a DataLoader of decoded samples (decode_only dataset + identity_collate, any number of worker
    processes doing the JPEG decoding) and turns every list into a device batch with `GpuCollate` in the main
    process: the host only decodes, resize / normalise / crops run on the GPU."""

    def __init__(self, loader, device, size: int = 224):
        self.loader, self.collate = loader, GpuCollate(device, size)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for samples in self.loader:
            yield self.collate(samples)


# ----------------------------------------------------------------------------------------------- the dataset

class MultiModalDataset(torch.utils.data.Dataset):
    def __init__(self, json_file, clip_preprocess=None, cache_dir="./cache", use_batch_cache=True,
                 cache_filename="train_precache.pkl", detector: Optional[Callable] = None, decode_only: bool = False,
                 seed: Optional[int] = None):
        try:
            with open(json_file, "r", encoding="utf-8") as f:
                self.data = json.load(f)
        except Exception as e:                                     # :89-95
            print(f"Error loading dataset JSON: {e}")
            self.data = []
        self.preprocess = clip_preprocess if clip_preprocess is not None else ClipImagePreprocess()
        self.cache_dir = cache_dir
        os.makedirs(cache_dir, exist_ok=True)
        self.use_batch_cache = use_batch_cache
        self.cache_filename = cache_filename
        self.detector = detector
        self.decode_only = decode_only
        self.rng = random.Random(seed) if seed is not None else random
        self.cached_yolo = {}
        if use_batch_cache:
            self.cached_yolo = open_box_cache(cache_dir, cache_filename)
            if not self.cached_yolo and detector is not None:      # strategy 2 (:264-306): build and save the cache
                for item in self.data:
                    if "image_path" in item:
                        self.cached_yolo[item["image_path"]] = _norm_boxes(detector(item["image_path"]))
                try:
                    atomic_pickle_dump(self.cached_yolo, os.path.join(cache_dir, cache_filename))
                except Exception as e:
                    print(f"Error saving cache: {e}")

    def __len__(self):
        return len(self.data)

    def _boxes(self, image_path: str):
        if not self.use_batch_cache:
            return load_or_compute_yolo(image_path, self.preprocess, self.cache_dir, self.detector)
        boxes = self.cached_yolo.get(image_path, None)
        if boxes is None:                                          # :357-360: in-memory update only
            boxes = load_or_compute_yolo(image_path, self.preprocess, self.cache_dir, self.detector)
            self.cached_yolo[image_path] = boxes
        return boxes

    def __getitem__(self, idx):
        from PIL import Image
        max_retries, current = 3, idx
        for retry in range(max_retries):
            try:
                item = self.data[current]
                image_path = item.get("image_path", "")
                if "captions" in item:                             # :331-335: one caption drawn at random
                    captions = item.get("captions", [])
                    caption = self.rng.choice(captions) if captions else ""
                else:
                    caption = item.get("caption", "")
                with Image.open(image_path) as img:
                    image = img.convert("RGB").copy()
                if self.decode_only:
                    pixel_values = np.array(image, dtype=np.uint8)
                else:
                    pixel_values = self.preprocess(images=image, text="", return_tensors="pt")["pixel_values"].squeeze(0)
                return pixel_values, caption, image_path, self._boxes(image_path)
            except RuntimeError:
                raise                                              # a missing detector is a setup error, not bad data
            except Exception as e:
                if retry < max_retries - 1:
                    current = (current + 1) % len(self.data)
                    print(f"Retrying with index {current} after error: {e}")
                else:
                    print(f"Failed after {max_retries} attempts. Last error: {e}")
        size = getattr(self.preprocess, "size", 224)
        size = size if isinstance(size, int) else 224
        blank = np.zeros((size, size, 3), dtype=np.uint8) if self.decode_only else torch.zeros(3, size, size)
        return blank, "", "", []                                   # :401-405

    @staticmethod
    def custom_collate_fn(batch):
        pixel_values, captions, image_paths, weighted_boxes_batch = zip(*batch)
        return torch.stack(pixel_values), list(captions), list(image_paths), list(weighted_boxes_batch)
