"""Multi-GPU layer: one process per GPU, scene replicated, samples sliced by global frame index, one sum-reduction
of the fp32 accumulation buffers over RCCL (torch.distributed backend "nccl" on ROCm) or gloo on CPU.

The reference is single-GPU; the only contract kept is the seed rule (main.comp:251): rank r renders global frames
r, r+N, r+2N, ... so the union over ranks is exactly the sample set of a one-GPU run of N*k frames.
"""
from __future__ import annotations

import numpy as np


def frame_slice(rank: int, world: int):
    """(first, stride) for rt_set_frame_slice."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return rank, world


def tile_slice(rank: int, world: int):
    """(first, stride) for rt_set_tile_slice: rank r renders the 8x8 tiles r, r + world, ... (numbered row by row)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return rank, world


def tile_mask(width: int, height: int, first: int, stride: int) -> np.ndarray:
    """Boolean (height, width) image of the pixels a tile slice owns."""
    tx = (width + 7) // 8
    yy, xx = np.mgrid[0:height, 0:width]
    tile = (yy // 8) * tx + xx // 8
    return (tile >= first) & ((tile - first) % stride == 0)


def frames_of_rank(rank: int, world: int, n_local: int):
    first, stride = frame_slice(rank, world)
    return [first + k * stride for k in range(n_local)]


class DeviceBufferView:
    """Zero-copy torch view of a raw device allocation (the renderer's fp32 RGBA sum buffer)."""

    def __init__(self, ptr: int, n_bytes: int):
        self.__cuda_array_interface__ = {"shape": (n_bytes // 4,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def accumulator_tensor(renderer, device):
    import torch
    ptr, n_bytes = renderer.accum_device_buffer()
    return torch.as_tensor(DeviceBufferView(ptr, n_bytes), device=device)


def reduce_accumulators(t, dst: int | None = None):
    """Sum-reduces the per-rank accumulation buffers in place: all-reduce, or reduce to `dst`."""
    import torch.distributed as dist
    if dst is None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    else:
        dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)
    return t


def mean_from_sum(sum_rgba: np.ndarray) -> np.ndarray:
    """HDR mean image from an RGBA sum buffer whose alpha holds the sample count."""
    s = np.asarray(sum_rgba, dtype=np.float32)
    n = np.maximum(s[..., 3:4], 1.0)
    return s[..., :3] / n
