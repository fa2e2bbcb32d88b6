"""hipGraph replay of the distillation step's forward + backward.

At the reference's default batch size (`--train_batch_size 32`, training/CLIP_image_distillation.py:716) the step is
a chain of ~600 short kernels and the host cannot launch them as fast as the MI355X retires them.  `GraphedStep`
records the whole `training_step(batch)` + `loss.backward()` chain ONCE into a HIP graph (torch.cuda.CUDAGraph —
the kernels of libdclip_hip.so are launched on torch's current stream, so stream capture sees every one of them)
and replays it per step on new data copied into the captured input buffers.  The optimizer stays outside the graph
(its bias-correction factors and learning rate are host scalars that change every step).

Constraints (checked): single process (collectives inside the backward are not captured here), tensor ("dict")
batches of a fixed shape, no host synchronisation inside the step.  Parameter gradients live in the graph's memory
pool: read them after `step()`, do not set them to None.
"""
from __future__ import annotations

from typing import Dict

import torch


class GraphedStep:
    def __init__(self, module, example_batch: Dict[str, torch.Tensor], warmup: int = 2):
        if getattr(module, "process_group", None) is not None:
            raise RuntimeError("GraphedStep captures a single-process step (process_group must be None)")
        if not isinstance(example_batch, dict):
            raise TypeError("GraphedStep needs a tensor batch (dict), not the (images, captions, paths, boxes) tuple")
        self.module = module
        dev = module.device
        self.static = {k: (v.to(dev).clone() if isinstance(v, torch.Tensor) else v) for k, v in example_batch.items()}
        params = [p for p in module.parameters() if p.requires_grad]
        # eager warm-up on a side stream: sizes every workspace / allocator pool before capture
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                for p in params:
                    p.grad = None
                module.training_step(self.static).backward()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        for p in params:
            p.grad = None
        # a bf16 student multiplies by bf16 COPIES of its fp32 master weights: the conversion must be inside the graph,
        # or every replay would use the copies made during the warm-up above (the optimizer runs outside the graph)
        for m in module.modules():
            if hasattr(m, "invalidate_bf16_of_trainable"):
                m.invalidate_bf16_of_trainable()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = module.training_step(self.static)
            self.loss.backward()
        self.losses = dict(getattr(module, "last_losses", {}))

    def load(self, batch: Dict[str, torch.Tensor]):
        for k, v in batch.items():
            if isinstance(v, torch.Tensor):
                dst = self.static[k]
                if tuple(v.shape) != tuple(dst.shape) or v.dtype != dst.dtype:
                    raise ValueError(f"GraphedStep: batch[{k!r}] is {tuple(v.shape)}/{v.dtype}, captured "
                                     f"{tuple(dst.shape)}/{dst.dtype} (a HIP graph replays fixed shapes)")
                # pageable host memory: a non-blocking copy is staged by the runtime and is NOT ordered with the graph
                # launch that follows (seen as a replay on stale inputs, 2e-4 drift of a trainer); only device or
                # pinned sources may go asynchronously
                dst.copy_(v, non_blocking=bool(v.is_cuda or v.is_pinned()))

    def step(self, batch: Dict[str, torch.Tensor] = None) -> torch.Tensor:
        """Forward + backward on `batch` (None: the data already in the captured buffers).  Returns the loss tensor
        (captured buffer, overwritten by the next replay); parameter .grad fields hold this step's gradients."""
        if batch is not None:
            self.load(batch)
        self.graph.replay()
        return self.loss

    __call__ = step
