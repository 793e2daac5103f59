"""Multi-GPU forms of the path (one process per GPU, torch.distributed: backend "nccl" = RCCL over xGMI).

* jobs (BASELINE config 5): independent content x style jobs, one per GPU, no data-path collective;
  `aggregate_throughput` is all that touches the network (after the timed region).
* levels (BASELINE config 4): loss = sum over pyramid levels of loss_l(D^l x), so a rank that owns a
  subset of the levels produces a partial pixel gradient and partial loss rows; ONE all-reduce(sum) of
  the (3,H0,W0) gradient plus the 4*levels+1 loss scalars per closure completes them, and the
  optimiser update is replicated deterministically on every rank (no broadcast).
  xGMI is point-to-point: for the 18.9 MB (L=2) / 75.5 MB (L=3) gradient RCCL's direct
  reduce-scatter + all-gather over the mesh costs ~0.1-0.3 ms against a 7-30 ms closure.
  The work split is 75/19/5/1 %, so this form tops out at 1.33x; it exists for memory (each rank
  holds only its levels' activations) and as the exchange step spatial sharding will reuse."""
from __future__ import annotations

from typing import Sequence

import torch


def deal_levels(levels: Sequence[int], world: int) -> Sequence[Sequence[int]]:
    """Levels dealt to `world` ranks, largest first onto the least-loaded rank (ties: the lowest rank).  The work of
    level l is 4**-l of level 0's.  world 2, levels 0-2: [[0], [1, 2]] = 76/24 %, where l % world gave 81/19."""
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for l in sorted(levels):
        r = min(range(world), key=lambda i: (load[i], i))
        out[r].append(l)
        load[r] += 4.0 ** -l
    return out


def owned_levels(levels_num: int, rank: int, world: int) -> Sequence[int]:
    """Levels of rank `rank` under level sharding (level 0, 75 % of the work, is alone on rank 0 whenever world >= 2)."""
    return deal_levels(range(levels_num), world)[rank]


def level_mask(levels_num: int, rank: int, world: int) -> int:
    m = 0
    for l in owned_levels(levels_num, rank, world):
        m |= 1 << l
    return m


def allreduce_closure(grad: torch.Tensor, losses: torch.Tensor, dist_mod=None, group=None) -> None:
    """In place: sums the partial pixel gradient and the partial loss rows over the ranks.  `losses`
    holds 4 floats per level (zeros for levels this rank does not own) + the partial grand total."""
    if dist_mod is None:
        import torch.distributed as dist_mod
    dist_mod.all_reduce(grad, op=dist_mod.ReduceOp.SUM, group=group)
    dist_mod.all_reduce(losses, op=dist_mod.ReduceOp.SUM, group=group)
    reform_total(losses)


def reform_total(losses: torch.Tensor) -> None:
    """Every level row has exactly one non-zero contributor, so the summed rows are exact; the grand total is re-formed
    from them in level order - the association the unsharded closure (and the reference, :179-185) uses.  L-BFGS'
    accept test `f_new < f` flips on a one-ulp difference, so this keeps sharded and unsharded runs on the same branch."""
    levels = (losses.numel() - 1) // 4
    total = losses[0].clone()
    for l in range(1, levels):
        total = total + losses[4 * l]
    losses[-1] = total


def aggregate_throughput(done: int, seconds: float, dist_mod=None, device=None):
    """(total closures over all ranks, slowest rank's time)."""
    if dist_mod is None or not dist_mod.is_initialized():
        return done, seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    d = torch.tensor([float(done)], dtype=torch.float64, device=device)
    dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
    dist_mod.all_reduce(d, op=dist_mod.ReduceOp.SUM)
    return int(round(d.item())), float(t.item())


# ---- spatial sharding of the top pyramid level (SURVEY 8(e) partition B, halo recompute) -------------------------------
STRIPE_HALO = 96        # rows: >= the receptive-field radius of relu5_1 (78), a multiple of 16


class StripePlan:
    """Rows of an H0-row image that rank `rank` of `world` owns, and the extended stripe it evaluates: the owned rows
    plus STRIPE_HALO rows on each interior side.  Boundaries between stripes are multiples of 16 (pooling alignment)."""

    def __init__(self, H0: int, world: int, rank: int, halo: int = STRIPE_HALO):
        if halo % 16:
            raise ValueError("the halo must be a multiple of 16 rows")
        units = H0 // 16
        if world > units:
            raise ValueError("more ranks than 16-row units")
        lo = (units * rank // world) * 16
        hi = H0 if rank == world - 1 else (units * (rank + 1) // world) * 16     # the bottom stripe takes a ragged remainder
        self.H0 = H0
        self.own = (lo, hi)                                     # owned rows in image coordinates
        self.ext = (max(0, lo - halo), min(H0, hi + halo))      # rows of the stripe image
        self.row0 = lo - self.ext[0]                            # owned rows in stripe coordinates
        self.rows = hi - lo

    @property
    def ext_rows(self) -> int:
        return self.ext[1] - self.ext[0]

    def cut(self, img):
        """rows of a (1,3,H0,W) tensor that form the stripe image (a contiguous copy)"""
        return img[:, :, self.ext[0]:self.ext[1], :].contiguous()

    def add_into(self, full, stripe_grad):
        """overlap-add of a stripe's gradient into the (1,3,H0,W) gradient"""
        full[:, :, self.ext[0]:self.ext[1], :] += stripe_grad
        return full
