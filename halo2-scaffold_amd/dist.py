"""Sliced multi-GPU MSM: one process per GPU, contiguous slice of scalars and bases per rank, the
96-byte partial points combined by all-gather + fold.

This is the partition halo2_proofs::arithmetic::best_multiexp applies per CPU thread (contiguous
chunks, partial results folded by addition) lifted to GPUs.  RCCL has no user-defined reduction, so
the "all-reduce of EC points" is an all-gather of world x 96 B over xGMI followed by world-1 point
additions on every rank (latency-bound: < 1 KB payload).
`fold` ((world, k, 12) -> (k, 12)) defaults to the device kernel (h2mi_g1_fold_groups); tests inject a
CPU fold to exercise the collective plumbing over gloo without a GPU.
"""
import numpy as np


def slice_bounds(n: int, rank: int, world: int):
    """contiguous slice [lo, hi) of rank; sizes differ by at most one when world does not divide n."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def device_fold(allp: np.ndarray) -> np.ndarray:
    """(world, k, 12) partial Jacobian points -> (k, 12): one device launch folds all k MSMs."""
    from ._lib import check, lib

    allp = np.ascontiguousarray(allp, dtype=np.uint64)
    world, k = allp.shape[0], allp.shape[1]
    out = np.zeros((k, 12), dtype=np.uint64)
    check(lib.h2mi_g1_fold_groups(allp.ctypes.data, world, k, out.ctypes.data), "fold")
    return out


class PartialPointCombiner:
    """all-gather (k,12)-limb partial Jacobian points from every rank and fold them per slot."""

    def __init__(self, fold=device_fold, device=None, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.fold = fold
        self.group = group
        self.device = device
        self.world = dist.get_world_size(group)
        self._gathered = None

    def __call__(self, partial: np.ndarray) -> np.ndarray:
        import torch

        partial = np.ascontiguousarray(partial, dtype=np.uint64).reshape(-1, 12)
        k = len(partial)
        # uint64 has no NCCL dtype: ship the limbs as int64 bit patterns
        t = torch.from_numpy(partial.view(np.int64).copy())
        if self.device is not None:
            t = t.to(self.device)
        if self._gathered is None or self._gathered[0].shape != t.shape or self._gathered[0].device != t.device:
            self._gathered = [torch.empty_like(t) for _ in range(self.world)]  # reused: one step = one tiny all-gather
        self.dist.all_gather(self._gathered, t, group=self.group)
        allp = torch.stack(self._gathered).cpu().numpy().view(np.uint64)  # (world, k, 12): one copy back, not `world`
        return self.fold(allp)
