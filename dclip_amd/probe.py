"""Re-derivable probe vectors: fixtures store <grad, probe(key)> instead of whole gradients."""
import zlib

import torch


def probe_vector(key: str, numel: int) -> torch.Tensor:
    gen = torch.Generator().manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    return torch.randn(numel, generator=gen, dtype=torch.float32)
