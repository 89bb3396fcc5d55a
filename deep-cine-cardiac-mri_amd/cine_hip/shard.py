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


class GradientAllReduce:
    """Data-parallel TRAINING over the same ranks: every rank runs the training step of its own slices (batch 1, like the reference's
    Lightning loop, pl_modules/varnet_module.py:97-113) and the parameter gradients are averaged with ONE all-reduce per step.  The models
    are small (4.3 MB XF-VarNet ... 24.7 MB XPDNet) against a 40-80 ms step, so a single flat bucket after ``loss.backward()`` is the right
    shape for xGMI's per-link rings: one latency, no per-layer hooks to overlap.  Aliased parameters (the cascades share their networks)
    appear once.  Usage:  sync = GradientAllReduce(model);  loss.backward();  sync();  optimiser.step()."""

    def __init__(self, module: torch.nn.Module, group=None, broadcast: bool = True):
        seen, self.params = set(), []
        for p in module.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p)); self.params.append(p)
        self.group = group
        # Replicas must start from identical weights (torch DDP does the same in its constructor): ONE flat broadcast from the
        # group's first rank, so differently seeded ranks cannot diverge silently.  broadcast=False only checks (a checksum
        # all-reduce) and raises on a mismatch.
        if self._active() and self.params:
            flat = torch.cat([p.detach().reshape(-1) for p in self.params])
            if broadcast:
                dist.broadcast(flat, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
                with torch.no_grad():
                    # into the parameters THEMSELVES (not p.data): the copy bumps p._version, which is what the packed-weight caches
                    # (UnetWeights.pointers, the MWCNN / CRNN packs) and the stale-weight training key are keyed on -- a forward that
                    # ran before this constructor must not leave pre-broadcast packed weights behind
                    torch._foreach_copy_(list(self.params),
                                         [c.view_as(p) for c, p in zip(flat.split([p.numel() for p in self.params]), self.params)])
            else:
                ck = torch.stack([flat.double().sum(), flat.double().abs().sum()])
                lo, hi = ck.clone(), ck.clone()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
                dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
                if not torch.equal(lo, hi):
                    raise RuntimeError("GradientAllReduce: the ranks hold different parameter values (seed them identically or pass broadcast=True)")

    def _active(self) -> bool:
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(self.group) > 1 or FORCE_COLLECTIVE

    def __call__(self) -> None:
        if not self._active():
            return
        world = dist.get_world_size(self.group)
        # A parameter without a gradient (a sub-network this step did not use, e.g. the sensitivity network when sens_maps are
        # passed in) contributes zeros.  The reduced gradient is written on EVERY rank where ANY rank had one (what DDP's
        # find_unused_parameters does): the `have` bitmap rides at the end of the same buffer (a SUM of 0 / 1 flags), so ranks that
        # disagree about which parameters are unused still take the same optimiser step and the replicas cannot drift apart.
        have = [p.grad is not None for p in self.params]
        dev, dt = self.params[0].device, self.params[0].dtype
        nparam = len(self.params)
        complete = all(have)
        if complete:          # the usual step: the flags are a cached device tensor of ones -- no host-to-device copy, and no answer to wait for
            if getattr(self, "_ones", None) is None or self._ones.device != dev:
                self._ones = torch.ones(nparam, dtype=dt, device=dev)
            flags = self._ones
        else:
            flags = torch.tensor([float(h) for h in have], dtype=dt, device=dev)
        flat = torch.cat([p.grad.reshape(-1) if h else torch.zeros(p.numel(), dtype=p.dtype, device=p.device)
                          for p, h in zip(self.params, have)] + [flags])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)       # one gather kernel, one collective, one scatter
        # Only a rank that LACKS a gradient has to learn whether another rank had it (a blocking read-back, and the one thing in here that
        # cannot be captured into a hipGraph); a rank with every gradient writes every parameter anyway and keeps enqueueing ahead.
        any_have = [True] * nparam if complete else (flat[-nparam:] > 0).tolist()
        flat = flat[:-nparam].mul_(1.0 / world)
        pieces = flat.split([p.numel() for p in self.params])
        dst, src = [], []
        for c, p, h, ah in zip(pieces, self.params, have, any_have):
            if h:
                dst.append(p.grad); src.append(c.view_as(p.grad))
            elif ah:
                p.grad = c.view_as(p).clone()                 # unused here, used on another rank: take the average like everybody else
        if dst:
            torch._foreach_copy_(dst, src)
