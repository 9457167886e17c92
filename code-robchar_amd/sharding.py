"""Multi-GPU sharding of the (controller x perturbation) sample space - one process per GPU.

The reference has no multi-process path (its only parallel construct, a `multiprocessing.Pool` per
controller at mcsim.py:451-455, is never enabled).  Every (controller, draw) evaluation is independent
(noise_model.py:98-109 keeps no cross-sample state), so the sample space is partitioned by CONTROLLER:
rank r owns a contiguous block of controllers with all K draws of each, which keeps the per-controller
reductions (RIM, std, min, Q, sort) rank-local.  The single exchange step is an all-gather of the
per-rank fidelity slabs (RCCL over xGMI when the backend is "nccl"; gloo on CPU for the tests), optionally
of the per-controller metric rows only.

`controller_partition` / `padded_rows` are what `MCDataSim`, `bench.py` and the C ABI's multi-device entry share.
`ShardedMC` is the one-level building block for callers that drive single sigma levels themselves (full fidelity
slabs on every rank); `MCDataSim` has its own device-resident pipeline on the same partition (metric rows gathered,
slabs only for a `.mc` cache).  `compute` is injected: callers pass `backend.mc_fidelity` (HIP); the CPU tests pass
the oracle so that the partitioning, padding and reassembly logic runs under gloo without a GPU.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np


def controller_partition(num_controllers: int, world_size: int):
    """Contiguous, balanced split of controller indices: the first (C mod G) ranks get one extra.
    Returns a list of (start, stop) per rank."""
    base, extra = divmod(int(num_controllers), int(world_size))
    bounds, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < extra else 0)
        bounds.append((start, start + n))
        start += n
    return bounds


def padded_rows(num_controllers: int, world_size: int) -> int:
    """Rows per rank after padding to equal shard size (all_gather_into_tensor needs equal shapes)."""
    return -(-int(num_controllers) // int(world_size))


@dataclass
class ShardedLevel:
    """All-gathered result of one sigma level on every rank."""
    fid: object            # (C, K) fidelity tensor (torch or numpy), fully reassembled
    local_rows: tuple      # (start, stop) of this rank's controllers


class ShardedMC:
    """Runs one sigma level of the MC path sharded over the ranks of a torch.distributed group.

    Each rank evaluates its controller block and the slabs are all-gathered so that every rank holds the
    full (C, K) tensor (the layout `get_algo_fid_dist` produces, mcsim.py:423-456).
    """

    def __init__(self, compute: Callable, group=None, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.compute = compute
        self.device = device

    def local_slice(self, num_controllers: int):
        return controller_partition(num_controllers, self.world)[self.rank]

    def run_level(self, controllers, local_draws, nspin: int, inspin: int, outspin: int,
                  num_controllers: Optional[int] = None, **kw) -> ShardedLevel:
        """controllers: the FULL (C, N+1) array (every rank has the controller file);
        local_draws: (C_local, K, N, 3) draws of this rank's block only."""
        import torch
        C = int(num_controllers if num_controllers is not None else len(controllers))
        lo, hi = self.local_slice(C)
        rows = padded_rows(C, self.world)
        ctrl_local = controllers[lo:hi]
        K = int(local_draws.shape[1])
        fid_local = self.compute(ctrl_local, local_draws, nspin, inspin, outspin, **kw)
        if not isinstance(fid_local, torch.Tensor):
            fid_local = torch.from_numpy(np.ascontiguousarray(fid_local))
        if self.world == 1:
            return ShardedLevel(fid_local, (lo, hi))
        # RCCL ("nccl") gathers device tensors over xGMI; gloo (CPU tests, rehearsals) needs host tensors
        via_host = fid_local.is_cuda and self.dist.get_backend(self.group) != "nccl"
        if via_host:
            fid_local = fid_local.cpu()
        shard = torch.full((rows, K), float("nan"), dtype=torch.float64, device=fid_local.device)
        shard[: hi - lo] = fid_local
        gathered = torch.empty((self.world * rows, K), dtype=torch.float64, device=fid_local.device)
        self.dist.all_gather_into_tensor(gathered, shard, group=self.group)
        # drop the padding rows of the short ranks
        bounds = controller_partition(C, self.world)
        if all(b[1] - b[0] == rows for b in bounds):
            full = gathered
        else:
            full = torch.cat([gathered[r * rows: r * rows + (b[1] - b[0])] for r, b in enumerate(bounds)])
        return ShardedLevel(full, (lo, hi))
