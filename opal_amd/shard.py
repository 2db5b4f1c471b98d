"""Multi-GPU plumbing of the search path: the database shards by sequence, every rank
searches its own shard, and the per-sequence results are collected on one rank with a single gather.

One process per GPU; `torch.distributed` backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU
tests. Nothing here computes alignments: it only decides which sequences a rank owns and moves
int32 result vectors.
"""
import numpy as np

from . import synth


def shard_bounds(total, world):
    """Contiguous, near-equal split of `total` sequences over `world` ranks -> int64[world+1]."""
    return (np.arange(world + 1, dtype=np.int64) * total) // world


def synthetic_shard(seqs_per_rank, rank, seed=42):
    """Rank `rank`'s shard of the synthetic database `seed` (weak scaling: every rank owns
    `seqs_per_rank` sequences; shard r is sequences [r*n, (r+1)*n) of the one global database)."""
    return synth.database(seqs_per_rank, seed=seed, first=rank * seqs_per_rank)


def gather_results(local, world, rank, dst=0):
    """ONE collective per result vector: every rank's `local` (1-D int32 tensor, equal length on all
    ranks) to rank `dst`. Returns the list of per-rank tensors on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return [local]
    out = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, out, dst=dst)
    return out


def global_order(gathered):
    """Concatenate per-rank result vectors into database order (shards are contiguous)."""
    import torch
    return torch.cat(list(gathered))
