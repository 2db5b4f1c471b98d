"""world_size-2 test of the multi-rank path on CPU (gloo): each rank owns a shard of the same synthetic
database, computes its shard's scores (with the CPU oracle standing in for the device search, which needs
a GPU), and rank 0 receives every score with one gather, in database order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib
from opal_amd import shard, synth
from opal_amd.score_matrix import blosum50

N_PER_RANK = 60
QLEN = 40


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res, offs, lens = shard.synthetic_shard(N_PER_RANK, rank)
    db = [res[offs[i]:offs[i + 1]] for i in range(N_PER_RANK)]
    q = synth.query(QLEN, seed=7)
    rc, s, eq, et = oracle_lib.Oracle().search(q, db, 11, 1, blosum50()[1], 0, oracle_lib.MODE_SW)
    assert rc == 0
    gathered = shard.gather_results(torch.from_numpy(s.copy()), world, rank)
    dist.barrier()
    if rank == 0:
        np.save(out_path, shard.global_order(gathered).numpy())
    else:
        assert gathered is None
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path, oracle):
    world = 2
    out = str(tmp_path / "scores.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    # single-process answer on the union of the shards
    res, offs, lens = synth.database(world * N_PER_RANK, seed=42)
    db = [res[offs[i]:offs[i + 1]] for i in range(world * N_PER_RANK)]
    rc, want, _, _ = oracle.search(synth.query(QLEN, seed=7), db, 11, 1, blosum50()[1], 0, oracle_lib.MODE_SW)
    np.testing.assert_array_equal(got, want)


def test_shard_bounds_cover_everything():
    for total, world in ((10, 3), (100000, 8), (7, 8), (0, 2)):
        b = shard.shard_bounds(total, world)
        assert b[0] == 0 and b[-1] == total and (np.diff(b) >= 0).all() and np.diff(b).max() - np.diff(b).min() <= 1
    # shards of the synthetic database are slices of the one global database
    a = shard.synthetic_shard(50, 1)
    g = synth.database(100, seed=42)
    assert (a[2] == g[2][50:]).all() and (a[0] == g[0][g[1][50]:]).all()
