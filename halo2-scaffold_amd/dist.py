"""Sliced multi-GPU MSM: one process per GPU, contiguous slice of scalars and bases per rank, the
96-byte partial points combined by all-gather + fold.

This is the partition halo2_proofs::arithmetic::best_multiexp applies per CPU thread (contiguous
chunks, partial results folded by addition) lifted to GPUs.  RCCL has no user-defined reduction, so
the "all-reduce of EC points" is an all-gather of world x 96 B over xGMI followed by world-1 point
additions on every rank (latency-bound: < 1 KB payload).

A prover needs the commitments of a phase before it can draw the next challenge (the joins of
create_proof: theta, beta/gamma, y, x, SHPLONK's v and u), so the combine happens at EVERY join, not once
per proof: `PhaseCombiner.combine(first_slot, count)`.

  device path (backend nccl = RCCL): the partial points are written by the library straight into a torch CUDA
      tensor; the all-gather is issued under the library's own stream (torch.cuda.ExternalStream around
      h2mi_library_stream), so it is ordered after the MSM reductions of the phase and before the fold
      (h2mi_g1_fold_groups_dev, also on the library stream) without any host synchronisation; nothing leaves HBM.
  host path (backend gloo: CPU rehearsal and the world-size-2 tests): device -> host copy of the phase's points,
      CPU all-gather, host -> device, the same device fold — or an injected CPU fold when there is no GPU at all.
"""
import ctypes as C

import numpy as np


def slice_bounds(n: int, rank: int, world: int):
    """contiguous slice [lo, hi) of rank; sizes differ by at most one when world does not divide n."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def device_fold(allp: np.ndarray) -> np.ndarray:
    """(world, k, 12) partial Jacobian points on the host -> (k, 12): one device launch folds all k MSMs."""
    from ._lib import check, lib

    allp = np.ascontiguousarray(allp, dtype=np.uint64)
    world, k = allp.shape[0], allp.shape[1]
    out = np.zeros((k, 12), dtype=np.uint64)
    check(lib.h2mi_g1_fold_groups(allp.ctypes.data, world, k, out.ctypes.data), "fold")
    return out


class PartialPointCombiner:
    """host-array form: all-gather (k,12)-limb partial Jacobian points from every rank and fold them per slot."""

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
        # uint64 has no NCCL dtype: ship the limbs as int64 bit patterns
        t = torch.from_numpy(partial.view(np.int64).copy())
        if self.device is not None:
            t = t.to(self.device)
        if self._gathered is None or self._gathered[0].shape != t.shape or self._gathered[0].device != t.device:
            self._gathered = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(self._gathered, t, group=self.group)
        allp = torch.stack(self._gathered).cpu().numpy().view(np.uint64)  # (world, k, 12)
        return self.fold(allp)


class PhaseCombiner:
    """Per-phase combine of the partial MSM results of a sliced multi-GPU prover.

    `partial_ptr` is where this rank's MSMs write their 96-byte results (slot i at partial_ptr + 96 i);
    `combine(first, count)` makes slots [first, first + count) of `combined_ptr` hold the sums over all ranks.
    """

    def __init__(self, slots: int, backend: str, torch_device=None, group=None, host_arrays=None):
        """host_arrays = (partial, combined, fold): numpy (slots, 12) arrays standing in for the two device buffers
        and a CPU fold ((world, count, 12) -> (count, 12)) — lets the world-size-2 CPU test drive the same
        phase / slot arithmetic and collective calls without a GPU (never used by the product)."""
        import torch
        import torch.distributed as dist

        from ._lib import check, lib
        from .device import DevBuf

        import os

        self.lib, self.check, self.dist, self.torch = lib, check, dist, torch
        self.group = group
        self.world = dist.get_world_size(group)
        self.slots = slots
        # H2MI_COMBINE=host|rccl (A/B for the first run on a real multi-GPU node; SURVEY.md 8e: "a host-side gather ... must
        # be benchmarked against RCCL"): rccl = the all-gather is stream-ordered on the library's stream and nothing leaves
        # HBM (default with backend nccl); host = the phase's 96-byte points are copied to the host (which synchronises),
        # gathered there and uploaded again — the collective still runs over the process group's backend
        mode = os.environ.get("H2MI_COMBINE", "rccl" if backend == "nccl" else "host")
        if mode not in ("host", "rccl") or (mode == "rccl" and backend != "nccl"):
            raise ValueError(f"H2MI_COMBINE={mode!r} with backend {backend!r}: use host or rccl (rccl needs the nccl backend)")
        self.mode = mode
        self.coll_device = torch_device if backend == "nccl" else None  # where collective tensors must live
        self.on_device = mode == "rccl"
        self.combines = 0
        self.host_arrays = host_arrays
        if host_arrays is not None:
            self.on_device = False
            self.partial_ptr = None
            return
        self.combined = DevBuf(96 * slots)
        if self.on_device:
            sp = C.c_void_p()
            check(lib.h2mi_library_stream(C.byref(sp)), "library_stream")
            self.stream = torch.cuda.ExternalStream(sp.value, device=torch_device)
            # torch owns the two exchange buffers (RCCL wants registered torch storage); the library writes and
            # reads them through their raw pointers
            self.partial_t = torch.zeros(12 * slots, dtype=torch.int64, device=torch_device)
            self.gathered_t = torch.zeros(12 * slots * self.world, dtype=torch.int64, device=torch_device)
            torch.cuda.synchronize(torch_device)
            self.partial_ptr = self.partial_t.data_ptr()
            # First contact (N > 1 has never run on hardware here): one all-gather of a single point on the library's stream, exactly
            # as combine() issues it.  Every rank reports whether it worked through an ordinary all-reduce; if ANY rank failed, ALL
            # ranks take the host-synchronised form — the decision is collective, so the ranks cannot end up in different modes.
            # (If the process group itself is unusable the all-reduce raises, as it would have anyway.)
            ok = 1
            try:
                with torch.cuda.stream(self.stream):
                    dist.all_gather_into_tensor(self.gathered_t[: 12 * self.world], self.partial_t[:12], group=group)
                self.stream.synchronize()
            except Exception as e:  # noqa: BLE001 - whatever the collective raises is the information
                ok = 0
                self.probe_error = repr(e)
            flag = torch.tensor([ok], dtype=torch.int32, device=torch_device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            torch.cuda.synchronize(torch_device)
            if int(flag.item()) == 0:
                import sys

                print(f"[h2mi] stream-ordered RCCL all-gather failed on some rank ({getattr(self, 'probe_error', 'not on this rank')}): "
                      "every rank falls back to H2MI_COMBINE=host", file=sys.stderr, flush=True)
                self.mode, self.on_device = "host", False
                del self.partial_t, self.gathered_t
                self._partial = DevBuf(96 * slots)
                self._gathered = DevBuf(96 * slots * self.world)
                self.partial_ptr = self._partial.ptr
        else:
            self._partial = DevBuf(96 * slots)
            self._gathered = DevBuf(96 * slots * self.world)
            self.partial_ptr = self._partial.ptr

    def combine(self, first: int, count: int) -> None:
        """all-gather + fold of slots [first, first + count); call after h2mi_join()."""
        if count == 0:
            return
        lib, check = self.lib, self.check
        self.combines += 1
        if self.on_device:
            torch = self.torch
            with torch.cuda.stream(self.stream):  # ordered on the library's stream: after the join, before the fold
                self.dist.all_gather_into_tensor(self.gathered_t[: 12 * count * self.world], self.partial_t[12 * first : 12 * (first + count)],
                                                 group=self.group)
            check(lib.h2mi_g1_fold_groups_dev(self.gathered_t.data_ptr(), self.world, count, self.combined.ptr + 96 * first, None), "fold")
            return
        torch = self.torch
        part = np.empty((count, 12), dtype=np.int64)
        if self.host_arrays is not None:
            part[:] = self.host_arrays[0][first : first + count].view(np.int64)
        else:
            check(lib.h2mi_memcpy_d2h(part.ctypes.data, self.partial_ptr + 96 * first, 96 * count), "d2h")
        mine = torch.from_numpy(part.reshape(-1))
        if self.coll_device is not None:  # RCCL collectives take device tensors: host-synchronised round trip through them
            mine = mine.to(self.coll_device)
        gathered = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(gathered, mine, group=self.group)
        allp = np.ascontiguousarray(torch.stack(gathered).cpu().numpy())
        if self.host_arrays is not None:
            _, combined, fold = self.host_arrays
            combined[first : first + count] = fold(allp.view(np.uint64).reshape(self.world, count, 12))
            return
        check(lib.h2mi_memcpy_h2d(self._gathered.ptr, allp.ctypes.data, allp.nbytes), "h2d")
        check(lib.h2mi_g1_fold_groups_dev(self._gathered.ptr, self.world, count, self.combined.ptr + 96 * first, None), "fold")

    def result(self) -> np.ndarray:
        """(slots, 12) combined Jacobian points (synchronises)."""
        if self.host_arrays is not None:
            return self.host_arrays[1]
        return self.combined.to_numpy(shape=(self.slots, 12))

    def release(self):
        if self.host_arrays is not None:
            return
        self.combined.free()
        if not self.on_device:
            self._partial.free()
            self._gathered.free()
