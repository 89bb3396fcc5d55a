"""Slice-level data parallelism: independent cine slices sharded over ranks, one all-gather for assembly.

The reference reconstructs one slice (HDF5 example, batch 1) per forward with no cross-slice state
(models/varnet.py:143-151), so ranks need no collective on the data path.  Backend "nccl" is RCCL on
ROCm; the same code runs on gloo for the CPU tests.
"""
from typing import List

import torch
import torch.distributed as dist


# bench.py --force-dist: run the collective even in a world of one rank (hardware rehearsal of the RCCL path on a 1-GPU box)
FORCE_COLLECTIVE = False


def slice_indices(n_slices: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership: slice i belongs to rank i % world."""
    return list(range(rank, n_slices, world))


def padded_count(n_slices: int, world: int) -> int:
    """Slices per rank after padding to equal shards (all_gather needs equal sizes)."""
    return (n_slices + world - 1) // world


def assemble_volume(local: torch.Tensor, n_slices: int) -> torch.Tensor:
    """local: (padded_count, ...) outputs of this rank's slices in ownership order (rows past the
    rank's real share are ignored).  Returns (n_slices, ...) in slice order on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not FORCE_COLLECTIVE):
        return local[:n_slices]
    world = dist.get_world_size()
    per = padded_count(n_slices, world)
    assert local.shape[0] == per, (local.shape, per)
    gathered = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, local.contiguous())
    # gathered[r * per + j] is slice r + j * world
    out = gathered.view(world, per, *local.shape[1:]).transpose(0, 1).reshape(world * per, *local.shape[1:])
    return out[:n_slices]
