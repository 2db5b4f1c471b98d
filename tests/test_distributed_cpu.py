"""world_size-2 tests of the multi-rank plumbing on CPU (gloo): the residue-balanced deal, the padded
single gather and the permutation back to database order. The device search needs a GPU, so here the CPU
oracle stands in for it (the same ranks running the real device search are covered by
tests/test_gpu_fullsize.py::test_two_rank_bench_on_one_gpu)."""
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


def _worker_balanced(rank, world, port, out_path, total):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    all_lens = synth.lengths(42, 0, total)
    owned_lists = shard.deal(all_lens, world)
    owned = owned_lists[rank]
    npad = max(len(o) for o in owned_lists)
    res, offs, lens = synth.database_subset(owned, 42, lens=all_lens[owned])
    db = [res[offs[i]:offs[i + 1]] for i in range(len(owned))]
    rc, s, eq, et = oracle_lib.Oracle().search(synth.query(QLEN, seed=7), db, 11, 1, blosum50()[1], 0, oracle_lib.MODE_SW)
    assert rc == 0
    block = torch.zeros(npad, dtype=torch.int32)
    block[: len(owned)] = torch.from_numpy(s.copy())
    gathered = shard.gather_results(block, world, rank)
    dist.barrier()
    if rank == 0:
        np.save(out_path, shard.scatter_to_global(gathered, owned_lists, total).numpy())
    dist.destroy_process_group()


def test_two_rank_balanced_deal_and_gather(tmp_path, oracle):
    """The bench path's sharding: non-contiguous, residue-balanced shards, padded gather, permutation back."""
    world, total = 2, 200
    out = str(tmp_path / "scores_balanced.npy")
    mp.spawn(_worker_balanced, args=(world, _free_port(), out, total), nprocs=world, join=True)
    got = np.load(out)
    res, offs, lens = synth.database(total, seed=42)
    db = [res[offs[i]:offs[i + 1]] for i in range(total)]
    rc, want, _, _ = oracle.search(synth.query(QLEN, seed=7), db, 11, 1, blosum50()[1], 0, oracle_lib.MODE_SW)
    np.testing.assert_array_equal(got, want)


def _worker_plan(rank, world, port, out_path, total):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    all_lens = synth.lengths(42, 0, total)
    owned_lists = shard.deal(all_lens, world)
    owned = owned_lists[rank]
    plan = shard.GatherPlan(owned_lists, 3, "cpu", root=rank == 0)
    res, offs, lens = synth.database_subset(owned, 42, lens=all_lens[owned])
    db = [res[offs[i]:offs[i + 1]] for i in range(len(owned))]
    rc, s, eq, et = oracle_lib.Oracle().search(synth.query(QLEN, seed=7), db, 11, 1, blosum50()[1], 1, oracle_lib.MODE_SW)
    assert rc == 0
    block = torch.full((3 * plan.npad,), -99, dtype=torch.int32)
    for k, v in enumerate((s, eq, et)):
        block[k * plan.npad:k * plan.npad + len(owned)] = torch.from_numpy(v.copy())
    for _ in range(2):  # the buffers of a plan are reused from step to step
        glob = plan.gather(block, rank)
    dist.barrier()
    if rank == 0:
        np.save(out_path, glob.numpy())
    else:
        assert glob is None
    dist.destroy_process_group()


def test_two_rank_gather_plan_one_index_operation(tmp_path, oracle):
    """bench.py's exchange as it runs with N > 1: every rank sends ONE padded {score, endQ, endT} block, the root puts all
    three vectors back into database order with one index_select."""
    world, total = 2, 300
    out = str(tmp_path / "plan.npy")
    mp.spawn(_worker_plan, args=(world, _free_port(), out, total), nprocs=world, join=True)
    got = np.load(out)
    res, offs, lens = synth.database(total, seed=42)
    db = [res[offs[i]:offs[i + 1]] for i in range(total)]
    rc, s, eq, et = oracle.search(synth.query(QLEN, seed=7), db, 11, 1, blosum50()[1], 1, oracle_lib.MODE_SW)
    np.testing.assert_array_equal(got[0], s)
    np.testing.assert_array_equal(got[1], eq)
    np.testing.assert_array_equal(got[2], et)


def test_bench_launcher_kills_the_other_ranks_when_one_dies():
    """bench.py --gpus 2 with rank 1 failing before the rendezvous and rank 0 stuck (both hooks act before torch or the GPU
    are touched, so this runs anywhere): the parent must notice, kill rank 0 and return non-zero within seconds, with the
    ranks' stderr tails in its own."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_FAIL_RANK="1", BENCH_HANG_RANK="0")
    env.pop("RANK", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert time.time() - t0 < 20
    assert "rank 1 exited with code 7" in p.stderr and "failing on request" in p.stderr
    # ... and the deadline: every rank stuck
    env = dict(os.environ, BENCH_HANG_RANK="0,1", BENCH_DEADLINE_S="2")
    env.pop("RANK", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "deadline" in p.stderr and time.time() - t0 < 20


def test_sprot_lengths_have_the_published_shape():
    """The "sprot" length distribution (bench.py --workload sprot): Swiss-Prot-like body, a real tail."""
    lens = synth.lengths(42, 0, synth.SPROT_SEQS, "sprot")
    assert len(lens) == 541762 and lens.max() == synth.SPROT_MAX_LEN and lens.min() >= 30
    assert 340 < lens.mean() < 370 and 150 < (lens > 5000).sum() < 400 and (lens > 20000).sum() >= 10
    # a slice equals the same slice of the whole (counter-based generator), and stride-separated residues do not overlap
    assert (synth.lengths(42, 1000, 500, "sprot") == lens[1000:1500]).all()
    ids = np.argsort(lens)[-2:]
    r, o, l = synth.database_subset(ids, 42, lens=lens[ids], stride=synth.SPROT_STRIDE)
    assert len(r) == int(lens[ids].sum())


def test_deal_balances_residues_and_partitions():
    lens = synth.lengths(42, 0, 50000)
    for world in (1, 2, 4, 8):
        owned = shard.deal(lens, world)
        allidx = np.sort(np.concatenate(owned))
        assert (allidx == np.arange(len(lens))).all()           # a partition
        loads = np.array([lens[o].astype(np.int64).sum() for o in owned])
        assert loads.max() - loads.min() <= 64 * 5000             # within one group of the longest sequences
        assert loads.max() / loads.mean() < 1.01
        for o in owned:                                           # every shard longest-first
            assert (np.diff(lens[o].astype(np.int64)) <= 0).all()
    # deterministic
    a, b = shard.deal(lens, 4), shard.deal(lens, 4)
    assert all((x == y).all() for x, y in zip(a, b))


def test_database_subset_matches_contiguous_generation():
    r, o, l = synth.database(400, 42, first=1000)
    ids = np.random.RandomState(0).permutation(400)[:120] + 1000
    r2, o2, l2 = synth.database_subset(ids, 42)
    for k, g in enumerate(ids):
        i = g - 1000
        assert l2[k] == l[i] and (r2[o2[k]:o2[k + 1]] == r[o[i]:o[i + 1]]).all()
    rt, ot = synth.residues_device(ids, l2, 42, "cpu")
    assert (rt.numpy() == r2).all() and (ot == o2).all()


def test_shard_bounds_cover_everything():
    for total, world in ((10, 3), (100000, 8), (7, 8), (0, 2)):
        b = shard.shard_bounds(total, world)
        assert b[0] == 0 and b[-1] == total and (np.diff(b) >= 0).all() and np.diff(b).max() - np.diff(b).min() <= 1
    # shards of the synthetic database are slices of the one global database
    a = shard.synthetic_shard(50, 1)
    g = synth.database(100, seed=42)
    assert (a[2] == g[2][50:]).all() and (a[0] == g[0][g[1][50]:]).all()
