#!/usr/bin/env python3
"""bench.py -- GCUPS of the database-search hot path on MI355X (driver contract in the task prompt).

A "step" is one pass of the hot path over one batch: one query against the whole synthetic protein
database shard of this rank, database already packed and resident in HBM, results written to device
buffers in database order, then (N > 1 only) ONE gather of the per-sequence results to rank 0 -- the path's only exchange
(issued by the host while the GPU is already on the next step's search: opalx_search_device_async / opalx_db_wait).

Workloads (BASELINE.json `configs`):
  c2 (default)  300-aa synthetic query x 100 000 synthetic sequences PER GPU, SW score-only, gaps 11/1  -> "weak"
  c3            the c2 database, --mode NW|HW|OV|SW with end locations
  c4            P18080 (513 aa) x 5 000 000 synthetic sequences in total, sharded over the GPUs        -> "strong"
  c5            2000-aa synthetic query on the c4 database (16 -> 32 bit re-run ladder)                 -> "strong"
  c5x3          BASELINE.json configs[4] as written -- "forcing 16/32-bit overflow re-run": BLOSUM50 cannot push a 2000-aa
                alignment past 16 bits, so the forced re-run comes from a x3-scaled matrix and PLANTED copies of
                the query (--planted, default 1 % of the sequences; exact copies and copies with 20 substitutions) among
                100 000 synthetic sequences: the 16-bit rung lists them, the 32-bit rung re-runs them in row chunks  -> "strong"
  sprot         the shape the reference was published on (reference aligner_comparison.md:6-34): 541 762 sequences with a
                Swiss-Prot-like length tail (up to 35 213 residues), gaps 3/1, SW score-only, the published queries
                O74807 / P19930 / Q3ZAI3 / P18080 (--query-fasta picks the timed one, default P18080)  -> "strong"
With N > 1 the global database is dealt to the ranks in 64-sequence groups balancing the residue count
(opal_amd/shard.py); rank 0 puts the gathered results back into database order with one index operation.
With N = 1 the rank owns the database in its own order: the library writes every result straight to its database index.

`--gpus N` with no RANK in the environment starts the N ranks itself (fresh child processes, before this
process touches torch or the GPU), watches them (first failure or the deadline kills the rest) and relays rank 0's
JSON line; under torchrun (RANK set) it runs as one rank.

GCUPS = queryLength * sum(dbSeqLengths) / 1e9 / seconds (reference src/opal_aligner.cpp:203-207;
cells of overflow re-runs are not counted).  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# VALU issue bound of the plain-16-bit kernel (profiles/r01_ubench_valu_*.txt, r02_ubench_bank.txt): v_add_u16 /
# v_sub_u16 clamp / v_max_i16 issue one wave-instruction per 2 clocks per SIMD; a Smith-Waterman cell costs 9 of
# them (6 for the skewed NW/HW/OV kernels), a wave-instruction covers 64 lanes = 64 cells.
SIMDS, CLOCK_HZ, CLK_PER_INSTR, LANES = 1024, 2.4e9, 2.0, 64.0
PUBLISHED_S = {"O74807": 5.2, "P19930": 6.9, "Q3ZAI3": 10.8, "P18080": 14.7}  # reference aligner_comparison.md:32 (OpAl AVX2, i7-4770K, 1 thread)


def valu_peak_gcups(instr_per_cell, clock_hz=CLOCK_HZ):
    return SIMDS * clock_hz / (instr_per_cell * CLK_PER_INSTR / LANES) / 1e9


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5", "c5x3", "sprot"])
    ap.add_argument("--seqs", type=int, default=None, help="database sequences per GPU (weak scaling)")
    ap.add_argument("--total-seqs", type=int, default=None, help="database sequences in total (strong scaling)")
    ap.add_argument("--qlen", type=int, default=None)
    ap.add_argument("--gap-open", type=int, default=None)
    ap.add_argument("--gap-ext", type=int, default=1)
    ap.add_argument("--mode", default="SW")
    ap.add_argument("--search-type", type=int, default=None)
    ap.add_argument("--in-flight", type=int, default=2, choices=[1, 2], help="N = 1: searches queued on the handle (2: step i is enqueued behind step i-1, then the host waits for step i-1)")
    ap.add_argument("--settle-ms", type=float, default=40.0, help="untimed steps for this many ms BEFORE the W warm-up steps: the GPU's clock "
                    "governor raises the shader clock with sustained activity (2.07 GHz after an idle gap, 2.38 after ~20 ms of back-to-back "
                    "launches: profiles/r05_clock_c2.txt); 0 = none")
    ap.add_argument("--planted", type=float, default=0.01, help="c5x3: fraction of the sequences replaced by copies of the query")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="default c2 run only: skip the strong-scaling legs `also.c4` / `also.c5` (BASELINE.json "
                    "configs[3] / [4]: P18080 and a 2000-aa query x 5 000 000 sequences in total, sharded over the N GPUs)")
    ap.add_argument("--query-fasta", default=None, help="use the first record of this FASTA file as the query")
    args = ap.parse_args(argv)
    wl = args.workload
    data = os.path.join(ROOT, "tests", "golden", "data")
    if wl in ("c4", "c5") and args.total_seqs is None and args.seqs is None:
        args.total_seqs = 5000000
    if wl == "sprot" and args.total_seqs is None and args.seqs is None:
        args.total_seqs = 541762
    if wl in ("c4", "sprot") and args.query_fasta is None and args.qlen is None:
        args.query_fasta = os.path.join(data, "query_P18080.fasta")
    if args.gap_open is None:
        args.gap_open = 3 if wl == "sprot" else 11
    if wl == "c5x3" and args.total_seqs is None and args.seqs is None:
        args.total_seqs = 100000
    if wl in ("c5", "c5x3") and args.qlen is None:
        args.qlen = 2000
    if args.qlen is None:
        args.qlen = 300
    if args.search_type is None:
        args.search_type = 1 if wl == "c3" else 0
    if args.seqs is None and args.total_seqs is None:
        args.seqs = 100000
    big = (args.total_seqs or 0) >= 1000000
    if args.steps is None:
        args.steps = 5 if big else 100
    if args.warmup is None:
        args.warmup = 1 if big else 10
    args.dist = "sprot" if wl == "sprot" else "c2"
    return args


def _tail(path, limit=3000):
    try:
        with open(path, "rb") as fh:
            fh.seek(0, os.SEEK_END)
            size = fh.tell()
            fh.seek(max(0, size - limit))
            return fh.read().decode(errors="replace")
    except OSError:
        return ""


def spawn_ranks(n):
    """Start the N ranks as fresh children of this process (which has not imported torch nor touched the GPU), one
    per GPU, rendezvous on 127.0.0.1, and WATCH them: every child is polled; the first non-zero exit, or the deadline
    (BENCH_DEADLINE_S, default 1500 s), kills the others within a fraction of a second -- a rank that died at init would
    otherwise leave the rest waiting in the rendezvous or in a collective until torch's own watchdog fires, minutes later.
    Every rank's stdout/stderr is kept; on failure their tails are printed. Rank 0's stdout (the JSON line) is relayed."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    deadline = time.time() + float(os.environ.get("BENCH_DEADLINE_S", "1500"))
    tmp = tempfile.mkdtemp(prefix="bench_ranks_")
    procs, files = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = open(os.path.join(tmp, f"rank{r}.out"), "wb")
        err = open(os.path.join(tmp, f"rank{r}.err"), "wb")
        files.append((out, err))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out, stderr=err))
    why = None
    while why is None:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            why = "rank %d exited with code %d" % bad[0]
        elif all(rc == 0 for rc in rcs):
            break
        elif time.time() > deadline:
            why = "deadline (BENCH_DEADLINE_S) passed"
        else:
            time.sleep(0.1)
    if why is not None:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    for out, err in files:
        out.close(); err.close()
    out0 = _tail(os.path.join(tmp, "rank0.out"), 1 << 20)
    if why is not None:
        sys.stderr.write(f"bench.py: {why}; exit codes {[p.returncode for p in procs]}\n")
        for r in range(n):
            sys.stderr.write(f"--- rank {r} stderr (tail)\n{_tail(os.path.join(tmp, f'rank{r}.err'))}\n")
            if r:
                sys.stderr.write(f"--- rank {r} stdout (tail)\n{_tail(os.path.join(tmp, f'rank{r}.out'), 1000)}\n")
        sys.stdout.write(out0)
        return 1
    sys.stderr.write(_tail(os.path.join(tmp, "rank0.err")))
    sys.stdout.write(out0)
    return 0


def cpu_quota_cores():
    """CPU cores this container may use on average (cgroup v2 cpu.max / v1 cfs quota), None = unlimited."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
            quota = int(fh.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
            period = int(fh.read())
        return None if quota <= 0 else quota / period
    except (OSError, ValueError):
        return None


class RefCpu:
    """The reference's own AVX2 path (oracle/_ref/libopal_ref.so, built from /root/reference in the build container)
    behind a small pthread harness (oracle/ref_bench.c). Checker / baseline only."""

    def __init__(self):
        import oracle_lib
        bench_so = os.path.join(oracle_lib.ORACLE_DIR, "_ref", "libref_bench.so")
        self.ok = oracle_lib.have_ref() and os.path.exists(bench_so)
        if self.ok:
            C.CDLL(oracle_lib.ref_lib_path(), mode=C.RTLD_GLOBAL)
            self.lib = C.CDLL(bench_so)
            self.lib.ref_bench_run.restype = C.c_double

    def run(self, query, res, offs, n, matrix, gap_open, gap_ext, mode_code, threads):
        """seconds for the first n sequences on `threads` threads"""
        q = np.ascontiguousarray(query, np.uint8); o = np.ascontiguousarray(offs, np.int64)
        m = np.ascontiguousarray(matrix, np.int32)
        up = lambda a: a.ctypes.data_as(C.POINTER(C.c_ubyte))
        return self.lib.ref_bench_run(up(q), len(q), up(res), o.ctypes.data_as(C.POINTER(C.c_int64)), int(n), int(gap_open),
                                      int(gap_ext), m.ctypes.data_as(C.POINTER(C.c_int)), int(m.shape[0]), int(mode_code),
                                      int(threads), None)


def cpu_baseline(query, res, offs, matrix, gap_open, gap_ext, mode_code):
    """The reference timed on this host's cores; falls back to the scalar oracle port if the reference build is absent.
    Bounded to ~10-30 s of CPU work. `value` is the SUSTAINED all-thread figure (back-to-back runs for >= 1 s): a
    container with a CPU quota runs a short burst on every hardware thread far above what it can sustain."""
    import oracle_lib
    n_all = len(offs) - 1
    hw = os.cpu_count() or 1
    quota = cpu_quota_cores()
    ref = RefCpu()
    res = np.ascontiguousarray(res, np.uint8)
    if ref.ok:
        o = np.ascontiguousarray(offs, np.int64)
        run = lambda n, t: ref.run(query, res, o, n, matrix, gap_open, gap_ext, mode_code, t)
        # bounded samples: a prefix of the shard holding ~5e9 cells for the one-thread run (the reference's own
        # methodology, aligner_comparison.md:11) and at most 5e10 cells for all threads
        csum = float(len(query)) * o[1:].astype(np.float64)
        n1 = int(min(n_all, np.searchsorted(csum, 5e9) + 1))
        n_mt = int(min(n_all, np.searchsorted(csum, 5e10) + 1))
        run(min(n_all, 2000), 1)  # warm-up
        t1 = min(run(n1, 1) for _ in range(2))
        one = float(len(query)) * float(o[n1]) / t1 / 1e9
        nt_burst = min(hw, 256)
        burst = float(len(query)) * float(o[n_mt]) / min(run(n_mt, nt_burst) for _ in range(3)) / 1e9
        # sustained: as many threads as the container may keep busy, back-to-back runs for at least a second
        nt = max(1, min(nt_burst, int(quota + 0.999) if quota else nt_burst))
        reps, t0 = 0, time.perf_counter()
        while reps < 3 or time.perf_counter() - t0 < 1.0:
            run(n_mt, nt)
            reps += 1
        sustained = reps * float(len(query)) * float(o[n_mt]) / (time.perf_counter() - t0) / 1e9
        return {"value": round(sustained, 2), "unit": "GCUPS", "cores": nt, "kind": "reference",
                "sample": f"sustained: {reps} back-to-back runs of the {n_mt} first of {n_all} sequences on {nt} threads "
                          f"({time.perf_counter() - t0:.1f} s); burst: best single run on {nt_burst} threads; 1 thread on the {n1} first",
                "cpu_quota_cores": quota, "hardware_threads": hw, "burst_value": round(burst, 2), "burst_threads": nt_burst,
                "one_thread_value": round(one, 2)}
    orc = oracle_lib.Oracle()
    n = min(n_all, 400)
    seqs = [res[offs[i]:offs[i + 1]] for i in range(n)]
    t0 = time.perf_counter()
    orc.search(query, seqs, gap_open, gap_ext, matrix, 0, mode_code)
    t1 = time.perf_counter() - t0
    return {"value": round(float(len(query)) * float(offs[n]) / t1 / 1e9, 4), "unit": "GCUPS", "cores": 1, "kind": "port",
            "sample": f"scalar oracle on the first {n} sequences", "cpu_quota_cores": quota, "hardware_threads": hw}


def e2e_call(query, res_host, offs, matrix, gap_open, gap_ext, search_type, mode_code, want_scores, calls=6):
    """The call the reference's own timer brackets (opal_aligner.cpp:157-165): opalSearchDatabase with N separate
    host pointers in and N result structs out, steady clock, best of `calls`. Two figures:
      cold  the context holds ANOTHER database before every timed call (the same sequences in another order), so the
            timed call orders, gathers, uploads and searches from scratch: the cost of a database seen for the first time;
      same  the same database again (many queries against one database through the plain API): the library finds its
            residues unchanged while gathering them and uploads nothing.
    Returns (cold_ms, same_ms, scores_match)."""
    from opal_amd import _lib
    from opal_amd.api import OpalSearchResult
    L = _lib.lib()
    n = len(offs) - 1
    q = np.ascontiguousarray(query, np.uint8)
    m = np.ascontiguousarray(matrix, np.int32)
    lens = np.ascontiguousarray(np.diff(offs), np.int32)
    # "another database": the same sequences handed over in another order (rotated by one), so that its lengths differ
    # position by position and the library must order, gather, upload and search from scratch, as for any new database
    base = res_host.ctypes.data + np.asarray(offs[:-1], dtype=np.uint64)
    ptrs = np.ascontiguousarray(base, dtype=np.uint64)
    ptrs_other = np.ascontiguousarray(np.roll(base, 1), dtype=np.uint64)
    lens_other = np.ascontiguousarray(np.roll(lens, 1), np.int32)
    arr = (OpalSearchResult * n)()
    words = np.frombuffer(arr, dtype=np.int32).reshape(n, 10)
    rptrs = np.ascontiguousarray(C.addressof(arr) + 40 * np.arange(n, dtype=np.uint64), dtype=np.uint64)

    def call(p, ln=lens):
        words[:, 0] = 0            # scoreSet = 0 (opalInitSearchResult)
        words[:, 2:6] = -1
        t0 = time.perf_counter()
        rc = L.opalSearchDatabase(q.ctypes.data_as(C.POINTER(C.c_ubyte)), len(q), C.c_void_p(p.ctypes.data), n,
                                  ln.ctypes.data_as(C.POINTER(C.c_int)), int(gap_open), int(gap_ext),
                                  m.ctypes.data_as(C.POINTER(C.c_int)), int(m.shape[0]), C.c_void_p(rptrs.ctypes.data),
                                  int(search_type), int(mode_code), 1)
        dt = (time.perf_counter() - t0) * 1e3
        if rc != 0:
            raise SystemExit(f"opalSearchDatabase failed rc={rc}")
        return dt
    cold, same, ok = None, None, True
    for _ in range(calls):
        call(ptrs_other, lens_other)
        dt = call(ptrs)
        cold = dt if cold is None else min(cold, dt)
        ok = ok and bool((words[:, 0] == 1).all() and (words[:, 1] == want_scores).all())
    for _ in range(calls):
        dt = call(ptrs)
        same = dt if same is None else min(same, dt)
        ok = ok and bool((words[:, 0] == 1).all() and (words[:, 1] == want_scores).all())
    return cold, same, ok

def strong_leg(name, query, total, rank, world, multi, backend, dev, matrix, gap_open, gap_ext, mode, steps=5, warmup=1):
    """One strong-scaling leg of the default run (BASELINE.json configs[3] / [4]): `query` x the first `total`
    sequences of the synthetic database, DEALT to the ranks (64-sequence groups by residue count), SW score-only; per step every
    rank searches its shard and the scores are gathered to rank 0 (the path's one exchange). Same bracket as the headline:
    barrier + synchronize on both sides of exactly `steps` steps, the maximum over the ranks; a sample of every rank's shard is
    checked against the CPU oracle (checker only). Every rank calls this; rank 0 gets the dictionary, the others None."""
    import torch
    import torch.distributed as dist
    import oracle_lib
    from opal_amd import shard, synth
    from opal_amd.api import PreparedDb
    all_lens = synth.lengths(42, 0, total)
    owned_lists = shard.deal(all_lens, world) if multi else [np.arange(total, dtype=np.int64)]
    owned = owned_lists[rank]
    n = len(owned)
    lens = all_lens[owned]
    npad = max(len(o) for o in owned_lists)
    d_res, offs = synth.residues_device(owned, lens, 42, dev)
    torch.cuda.synchronize()
    pdb = PreparedDb(None, offs, device_ptr=d_res.data_ptr())
    outs = [torch.zeros(npad, dtype=torch.int32, device=dev) for _ in range(2)]
    plan = shard.GatherPlan(owned_lists, 1, dev if backend == "nccl" else "cpu", root=rank == 0) if multi else None
    torch.cuda.synchronize()
    rdev = dev if backend == "nccl" else torch.device("cpu")
    kms, xms = [], []

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def step(i):
        o = outs[i & 1]
        rc, k = pdb.search_device(query, gap_open, gap_ext, matrix, 0, mode, o.data_ptr())
        if rc != 0:
            raise SystemExit(f"also.{name}: search failed rc={rc}")
        kms.append(k)
        if multi:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            plan.gather(o if backend == "nccl" else o.cpu(), rank)
            b.record()
            xms.append((a, b))
    for i in range(warmup):
        step(i)
    kms.clear(); xms.clear()
    for o in outs:
        o.fill_(-2147483647)
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
    rank_k = torch.zeros(world, dtype=torch.float64, device=rdev)
    rank_k[rank] = float(np.mean(kms))
    # a sample of this rank's shard against the oracle
    mine = outs[(steps - 1) & 1].cpu().numpy()[:n]
    rng = np.random.RandomState(11 + rank)
    pick = np.unique(np.concatenate([rng.choice(n, min(n, 40), replace=False), np.argsort(lens)[-2:]]))
    sres, soffs, _ = synth.database_subset(owned[pick], 42, lens=lens[pick])
    sdb = [sres[soffs[i]:soffs[i + 1]] for i in range(len(pick))]
    orc, so, _, _ = oracle_lib.Oracle().search(query, sdb, gap_open, gap_ext, matrix, 0, mode)
    ok = bool(orc == 0 and (mine[pick] == so).all())
    vflag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=rdev)
    if multi:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(rank_k, op=dist.ReduceOp.SUM)
        dist.all_reduce(vflag, op=dist.ReduceOp.MIN)
    out = None
    if rank == 0:
        dt = float(tmax.item())
        cells = float(len(query)) * float(all_lens.astype(np.int64).sum())
        rank_cells = [float(all_lens[o].astype(np.int64).sum()) for o in owned_lists]
        rk = [float(v) for v in rank_k.cpu().numpy()]
        verified = bool(int(vflag.item()) == 1)
        if multi:  # the gathered block holds rank 0's own values at their database indices
            verified = verified and bool((plan.glob.cpu().numpy()[0][owned] == mine).all())
            torch.cuda.synchronize()
        peak = valu_peak_gcups(9.0)
        out = {"workload": f"{name}: {len(query)}-aa query x {total} synthetic sequences in total (seed 42), SW score-only, gaps {gap_open}/{gap_ext}, "
                           f"dealt to {world} GPU(s) in 64-sequence groups by residue count, one gather of the scores per step",
               "value": round(cells * steps / dt / 1e9, 1), "unit": "GCUPS", "scaling": "strong", "steps": steps, "warmup": warmup,
               "ms_per_step": round(dt / steps * 1e3, 3),
               "kernel_ms_per_rank": [round(v, 3) for v in rk],
               "kernel_gcups_slowest_rank": round(cells / (max(rk) * 1e-3) / 1e9, 1),
               "frac_of_valu_bound_all_gpus": round(cells / (max(rk) * 1e-3) / 1e9 / (peak * world), 4),
               "imbalance": round(max(rank_cells) / (sum(rank_cells) / len(rank_cells)), 5),
               "exchange_ms": round(float(np.mean([a.elapsed_time(b) for a, b in xms])), 4) if xms else None,
               "verified": verified}
    pdb.close()
    del d_res, outs, plan
    torch.cuda.empty_cache()
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))  # nothing GPU-related has been imported or touched in this process

    # stdout carries exactly ONE line, the JSON: libraries write there too (RCCL prints "Librccl path : ..." when the
    # process group goes away, i.e. AFTER the result), so everything else of this process is sent to stderr and the line
    # is written to the real stdout at the very end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("BENCH_FAIL_RANK") == str(rank):  # test hook: this rank dies before the rendezvous
        sys.stderr.write(f"bench.py: rank {rank} failing on request (BENCH_FAIL_RANK)\n")
        sys.exit(7)
    if str(rank) in os.environ.get("BENCH_HANG_RANK", "").split(","):  # test hook: this rank never reaches the rendezvous
        time.sleep(3600)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the measured path)")
    # Test hooks for the multi-rank path on a single-GPU box (never used by the driver): all ranks on device 0,
    # results gathered through gloo on host copies. The default is one GPU per rank and RCCL.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if os.environ.get("BENCH_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # BENCH_FORCE_DIST=1 (test hook): one rank goes through everything N ranks go through -- process group on the chosen
    # backend (RCCL by default), deal, padded gather, index_select, barrier, all-reduce -- so that the RCCL branch can be
    # exercised on a one-GPU box (RCCL refuses two ranks on one GPU).
    multi = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import opal_amd
    from opal_amd import _lib, shard, synth
    from opal_amd.api import MODES, PreparedDb
    from opal_amd.fasta import read_fasta
    from opal_amd.score_matrix import blosum50

    strong = args.total_seqs is not None
    total = args.total_seqs if strong else args.seqs * world
    alphabet, matrix = blosum50()
    qname = "seed 7"
    if args.query_fasta:
        query = read_fasta(args.query_fasta, alphabet)[0]
        args.qlen = len(query)
        qname = os.path.basename(args.query_fasta).replace("query_", "").replace(".fasta", "")
    else:
        query = synth.query(args.qlen, seed=7)
    mode = MODES[args.mode]
    stride = synth.SPROT_STRIDE if args.dist == "sprot" else 5000

    # ---- this rank's shard of the one global database. One rank: the database in its own order (results land at their
    # database index). Several: 64-sequence groups dealt by residue count (opal_amd/shard.py).
    all_lens = synth.lengths(42, 0, total, args.dist)
    planted_ids = np.zeros(0, dtype=np.int64)
    if args.workload == "c5x3":
        # every k-th sequence becomes a copy of the query (odd ones of them with 20 substitutions at fixed seeds): under the x3
        # matrix they score beyond 32 767 - 45, the 16-bit rung hands them to the re-run list, the 32-bit rung computes them
        matrix = np.ascontiguousarray(matrix * 3, np.int32)
        nplant = max(1, int(round(total * args.planted)))
        planted_ids = (np.arange(nplant, dtype=np.int64) * total) // nplant
        all_lens = all_lens.copy()
        all_lens[planted_ids] = args.qlen
    owned_lists = shard.deal(all_lens, world) if multi else [np.arange(total, dtype=np.int64)]
    owned = owned_lists[rank]
    n = len(owned)
    npad = max(len(o) for o in owned_lists)
    lens = all_lens[owned]
    d_res, offs = synth.residues_device(owned, lens, 42, dev, stride=stride)
    if len(planted_ids):
        mine_planted = np.nonzero(np.isin(owned, planted_ids))[0]
        if len(mine_planted):
            rows = np.tile(np.asarray(query, np.uint8), (len(mine_planted), 1))
            for r, li in enumerate(mine_planted):
                if int(owned[li]) & 1:
                    rr = np.random.RandomState(1000 + int(owned[li]) % 100000)
                    pos = rr.choice(args.qlen, 20, replace=False)
                    rows[r, pos] = rr.randint(0, 20, 20).astype(np.uint8)
            idx = (torch.as_tensor(offs[mine_planted], device=dev).unsqueeze(1) + torch.arange(args.qlen, device=dev).unsqueeze(0)).reshape(-1)
            d_res[idx] = torch.as_tensor(rows.reshape(-1), device=dev)
    torch.cuda.synchronize()  # the library reads the buffer on its own stream: everything that produced it must be complete
    pdb = PreparedDb(None, offs, device_ptr=d_res.data_ptr())
    cells = float(args.qlen) * float(offs[-1])
    total_cells = float(args.qlen) * float(all_lens.astype(np.int64).sum())

    want_end = args.search_type != 0
    nres = 3 if want_end else 1
    # {score, endQ, endT} blocks, padded to the largest shard. Two result buffers used in turn (N > 1): the search runs on
    # the library's own stream, the gather on torch's, so step i+1 must not overwrite what step i's gather is still
    # reading; a buffer is reused two steps later, after waiting on the event recorded behind its last reader.
    outs = [torch.zeros(nres * npad, dtype=torch.int32, device=dev) for _ in range(2 if (multi or args.in_flight > 1) else 1)]
    done = [None, None]
    done_ev = [torch.cuda.Event(), torch.cuda.Event()] if multi else None  # (reused: no event creation per step)
    # position of every database sequence in the gathered [rank][value][npad] block: ONE index operation per step
    plan = shard.GatherPlan(owned_lists, nres, dev if backend == "nccl" else "cpu", root=rank == 0) if multi else None
    torch.cuda.synchronize()  # (the zero fills above run on torch's stream, the searches on the library's)
    kernel_ms = []
    period_ms = []  # (N = 1) device time from the end of the search before to the end of this one: queued searches overlap
    step_no = [0]

    pending = [None]  # (N > 1) the result buffer whose exchange is still to be issued
    inflight = [0]    # (N = 1) searches enqueued and not yet waited for

    xev = []      # (N > 1) timed event pairs around every exchange of the timed region (torch's stream), made before it starts
    xused = [0]

    def exchange(b):
        # the path's only exchange: ONE gather of the per-sequence result block to rank 0, which puts the
        # blocks back into database order with the deal's permutation (one index_select for all result vectors)
        pair = xev[xused[0]] if xused[0] < len(xev) else None
        if pair:
            pair[0].record()
        plan.gather(outs[b] if backend == "nccl" else outs[b].cpu(), rank)
        done[b] = done_ev[b]
        done[b].record()
        if pair:
            pair[1].record()
            xused[0] += 1

    def step():
        b = step_no[0] % len(outs)
        step_no[0] += 1
        out = outs[b]
        if done[b] is not None:
            done[b].synchronize()
        score, endq, endt = out[:npad], (out[npad:2 * npad] if want_end else None), (out[2 * npad:] if want_end else None)
        if not multi:
            # Two searches in flight: step i is enqueued behind step i-1 on the library's stream (no host round trip between
            # them on the GPU), then the host waits for step i-1 and reads its kernel time. A search is still one complete
            # pass -- header upload, every DP launch, results in HBM -- and the K timed steps end with a wait for the last one.
            rc = pdb.search_device_async(query, args.gap_open, args.gap_ext, matrix, args.search_type, mode, score.data_ptr(),
                                         endq.data_ptr() if want_end else 0, endt.data_ptr() if want_end else 0)
            kms = None
            if rc == 0 and (inflight[0] == 1 or args.in_flight == 1):
                rc, kms, per = pdb.wait2()
            elif rc == 0:
                inflight[0] = 1
            if rc != 0:
                raise SystemExit(f"search failed rc={rc}")
            if kms is not None:
                kernel_ms.append(kms)
                period_ms.append(per)
            return
        else:
            # N > 1: the same two searches in flight; the host waits for step i-1 and issues ITS exchange (the path's one gather:
            # ~0.2 ms of Python / RCCL call overhead) while the GPU is on step i, whose grid was released when step i-1 had
            # started its last workgroup
            rc = pdb.search_device_async(query, args.gap_open, args.gap_ext, matrix, args.search_type, mode, score.data_ptr(),
                                         endq.data_ptr() if want_end else 0, endt.data_ptr() if want_end else 0)
            if rc == 0 and pending[0] is not None:
                rc, kms, per = pdb.wait2()
                if rc == 0:
                    kernel_ms.append(kms)
                    period_ms.append(per)
                    exchange(pending[0])
            pending[0] = b
        if rc != 0:
            raise SystemExit(f"search failed rc={rc}")

    def flush():
        if multi and pending[0] is not None:
            rc, kms, per = pdb.wait2()
            if rc != 0:
                raise SystemExit(f"search failed rc={rc}")
            kernel_ms.append(kms)
            period_ms.append(per)
            exchange(pending[0])
            pending[0] = None
        if not multi and inflight[0] == 1:  # the last step's search
            rc, kms, per = pdb.wait2()
            inflight[0] = 0
            if rc != 0:
                raise SystemExit(f"search failed rc={rc}")
            kernel_ms.append(kms)
            period_ms.append(per)

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # settle: the process has just generated and packed its database (host work, the GPU idle): the clock governor starts every
    # burst of launches at its low point and needs ~20 ms of sustained activity to reach what it holds under load
    # (N > 1: every step carries a collective, so every rank must run the SAME number of them: one probing step, its time
    # reduced over the ranks, decides how many follow)
    settle_steps = 0
    if args.settle_ms > 0:
        fence()
        t_settle = time.perf_counter()
        step()
        flush()
        fence()
        t_one = torch.tensor([time.perf_counter() - t_settle], dtype=torch.float64, device=dev if backend == "nccl" else torch.device("cpu"))
        if multi:
            dist.all_reduce(t_one, op=dist.ReduceOp.MAX)
        more = int(min(1000, max(0, np.ceil(args.settle_ms * 1e-3 / max(float(t_one.item()), 1e-5)) - 1)))
        for _ in range(more):
            step()
        settle_steps = 1 + more
    for _ in range(args.warmup):
        step()
    flush()
    kernel_ms.clear()
    period_ms.clear()
    # Every result buffer is POISONED before the timed region (outside it): the warm-up steps have already written the right
    # answers, so a timed step that did nothing would otherwise still verify. The check below reads what the LAST timed steps
    # wrote (one buffer per search in flight) and fails on any value the timed region left untouched.
    POISON = -2147483647
    for o in outs:
        o.fill_(POISON)
    if multi:
        xev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    flush()  # (the last step's exchange: inside the timed region)
    fence()
    dt = time.perf_counter() - t0
    rdev = dev if backend == "nccl" else torch.device("cpu")
    tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
    if multi:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # ---- verification, after the timed region: EVERY rank checks a fixed sample of its own shard against the CPU oracle
    # (checker only); the verdicts are reduced into `verified`
    import oracle_lib
    oracle_lib.build_oracle()
    mine = outs[(step_no[0] - 1) % len(outs)].cpu().numpy().reshape(nres, npad)[:, :n]
    untouched = bool((mine == POISON).any())  # (a value the timed region never wrote)
    if len(outs) > 1 and args.steps >= 2:     # ... and the step before the last one wrote the other buffer: the same answers
        other = outs[(step_no[0] - 2) % len(outs)].cpu().numpy().reshape(nres, npad)[:, :n]
        untouched = untouched or not bool((other == mine).all())
    if untouched:
        sys.stderr.write(f"bench.py: rank {rank}: the timed steps left result values unwritten (or two steps disagree)\n")
    rng = np.random.RandomState(1 + rank)
    nsamp = min(n, (300 if args.qlen <= 600 else 120) // (1 if world == 1 else 2))
    pick = np.sort(rng.choice(n, nsamp, replace=False))
    if args.dist == "sprot":  # the long tail must be part of the sample
        pick = np.unique(np.concatenate([pick, np.argsort(lens)[-3:]]))
    if len(planted_ids):  # (planted copies among them: their residues are what the device buffer holds)
        mp = np.nonzero(np.isin(owned, planted_ids))[0]
        pick = np.unique(np.concatenate([pick[: max(20, nsamp // 3)], mp[:6]]))
        sdb = [d_res[int(offs[i]):int(offs[i + 1])].cpu().numpy() for i in pick]
    else:
        sres, soffs, _ = synth.database_subset(owned[pick], 42, lens=lens[pick], stride=stride)
        sdb = [sres[soffs[i]:soffs[i + 1]] for i in range(len(pick))]
    orc_rc, os_, oq, ot = oracle_lib.Oracle().search(query, sdb, args.gap_open, args.gap_ext, matrix, args.search_type, mode)
    ok_local = bool(orc_rc == 0 and (mine[0][pick] == os_).all()) and not untouched
    if want_end:
        ok_local = ok_local and bool((mine[1][pick] == oq).all() and (mine[2][pick] == ot).all())
    if not ok_local:
        sys.stderr.write(f"bench.py: rank {rank}: results differ from the oracle on its shard sample\n")
    vflag = torch.tensor([1 if ok_local else 0], dtype=torch.int32, device=rdev)
    if multi:
        dist.all_reduce(vflag, op=dist.ReduceOp.MIN)
    verified = bool(int(vflag.item()) == 1)

    # what an N > 1 line needs to be READ: every rank's mean kernel time (all-reduced one-hot vector), the share of cells
    # the deal gave every rank, the time of the exchange on rank 0's stream
    # N = 1: queued searches take turns on two streams of the handle and OVERLAP (the grid of search i+1 is released when search i
    # has started its last workgroup and fills the CUs it leaves as it drains), so the time a search costs is the device time
    # between two consecutive ends -- the period --, not the life of its own launch (which shares the chip at both ends). The
    # first timed search's period starts at the last warm-up search's end, across the untimed poisoning of the buffers: left out.
    launch_kms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    my_kms = launch_kms
    if len(period_ms) >= 3 and all(v > 0 for v in period_ms[1:]):
        my_kms = float(np.mean(period_ms[1:]))
    rank_kms = torch.zeros(world, dtype=torch.float64, device=rdev)
    rank_kms[rank] = my_kms
    if multi:
        dist.all_reduce(rank_kms, op=dist.ReduceOp.SUM)
    rank_kms = [float(v) for v in rank_kms.cpu().numpy()]
    rank_cells = [float(args.qlen) * float(all_lens[o].astype(np.int64).sum()) for o in owned_lists]
    exchange_ms = None
    if multi and xused[0] > 0:
        torch.cuda.synchronize()
        exchange_ms = float(np.mean([a.elapsed_time(b) for a, b in xev[:xused[0]]]))

    status = 0
    if rank == 0:
        gcups = total_cells * args.steps / dt / 1e9
        kms = my_kms
        kernel_gcups = cells / (kms * 1e-3) / 1e9
        gall = plan.glob.cpu().numpy() if multi else mine
        gscore = np.ascontiguousarray(gall[0])
        if multi:  # the gathered, re-ordered block must hold rank 0's own verified values at their database indices
            verified = verified and bool((gall[:, owned] == mine).all())

        # ---- the clock the chip holds under this kernel (it clocks to its power budget): a few more searches with the
        # library's clock probe on, outside the timed region (the probe's read-back would otherwise be part of a step)
        L = _lib.lib()
        L.opalx_db_clock_probe.argtypes = [C.c_void_p, C.c_int]
        L.opalx_db_last_clock_ghz.restype = C.c_double
        L.opalx_db_last_clock_ghz.argtypes = [C.c_void_p]
        # (round 5: the probed searches are QUEUED like the timed ones -- two in flight, no host round trip between them -- and run
        # for >= 30 ms before the counters of the last one are read: a synchronous probe search starts after an idle gap, and the
        # clock governor answers idle gaps of 2 ms with a lower clock (profiles/r05_clock_c2.txt: C2 read 2.13 GHz that way while
        # its timed steps ran at ~2.3))
        L.opalx_db_clock_probe(pdb._h, 1)
        nprobe = int(min(40, max(4, np.ceil(30.0 / max(kms, 1e-3)))))
        queued = 0
        for i in range(nprobe):
            o = outs[i % len(outs)]
            if pdb.search_device_async(query, args.gap_open, args.gap_ext, matrix, args.search_type, mode, o.data_ptr(),
                                       o[npad:].data_ptr() if want_end else 0, o[2 * npad:].data_ptr() if want_end else 0) != 0:
                break
            queued += 1
            if queued == 2:
                pdb.wait(); queued -= 1
        clock_ghz = L.opalx_db_last_clock_ghz(pdb._h) if nprobe else 0.0  # (waits for the last search, reads ITS counters)
        while queued > 0:
            pdb.wait(); queued -= 1
        L.opalx_db_clock_probe(pdb._h, 0)
        clock_ghz = float(clock_ghz) if clock_ghz and clock_ghz > 0 else None

        instr_per_cell = 9.0 if args.mode == "SW" else 6.0
        peak = valu_peak_gcups(instr_per_cell)
        # compulsory HBM bytes of THIS design per step: one byte per residue, 16 B of metadata per sequence (offset,
        # length, database index), 4 B per result value. The 4*w B/cell of a column-streaming design is kept only as a
        # labelled comparison: this kernel keeps the DP column in VGPRs and never moves those bytes.
        alg_bytes = float(lens.astype(np.int64).sum() + 16 * n + 4 * nres * n)
        traffic, traffic_src = None, None
        try:  # HBM bytes per step measured offline with rocprofv3 PMC passes on this same workload (never guessed for another)
            key = (args.workload, args.qlen, total if strong else args.seqs, args.mode, args.search_type, args.gap_open, world)
            tags = {("c2", 300, 100000, "SW", 0): "c2", ("c3", 300, 100000, "SW", 1): "c2end", ("c3", 300, 100000, "NW", 1): "c3nw",
                    ("c3", 300, 100000, "HW", 1): "c3hw", ("c3", 300, 100000, "OV", 1): "c3ov", ("c4", 513, 625000, "SW", 0): "c4",
                    ("c5", 2000, 100000, "SW", 0): "c5", ("c5", 2000, 100000, "SW", 1): "c5end", ("c5", 2000, 100000, "NW", 1): "c5nw",
                    ("c5", 2000, 100000, "HW", 1): "c5hw", ("c5", 2000, 100000, "OV", 1): "c5ov"}
            tag = tags.get(key[:5]) if (key[5], key[6]) == (11, 1) else None
            for rnd in ("r06", "r05", "r04", "r03", "r02"):
                path = os.path.join(ROOT, "profiles", f"{rnd}_{tag}.json")
                if tag and os.path.exists(path):
                    with open(path) as fh:
                        traffic = json.load(fh)["hbm_bytes_per_step"]
                    traffic_src = (f"offline: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload (profiles/{rnd}_{tag}.json); "
                                   "`achieved` divides those bytes by the live kernel time")
                    break
        except (OSError, KeyError, ValueError, TypeError):
            traffic = None
        hbm = {"peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes": alg_bytes,
               "algorithmic_gbs": round(alg_bytes / (kms * 1e-3) / 1e9, 2),
               "achieved": round(traffic / (kms * 1e-3) / 1e9, 2) if traffic else None,
               "achieved_source": "offline profile bytes / live kernel time" if traffic else None,
               "frac": round(traffic / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None,
               "column_streaming_model_bytes": cells * 8.0,
               "note": "the DP column lives in VGPRs: HBM carries residue tokens in and results out only"}
        what = f"{args.mode} {'score-only' if not want_end else 'score+end'}"
        res_line = {
            "metric": f"GCUPS (DP cell updates/s), {what}, 1 query vs synthetic protein DB",
            "value": round(gcups, 2), "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "i16", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {args.qlen}-aa query ({qname}) vs "
                                   f"{total} synthetic protein seqs in total (seed 42, mean {float(all_lens.mean()):.0f} aa, longest "
                                   f"{int(all_lens.max())}, {'fixed total' if strong else str(args.seqs) + ' per GPU'}), {what}, BLOSUM50, gaps "
                                   f"{args.gap_open}/{args.gap_ext}, 16-bit first rung, 32-bit re-run",
                       "cells_per_step": total_cells, "cells_per_step_rank0": cells,
                       **({"rerun": {"matrix": "BLOSUM50 x 3", "planted_copies_of_the_query": int(len(planted_ids)),
                                     "sequences_re_run_at_32_bits": int((gscore >= 32767 - int(matrix.max())).sum()),
                                     "re_run_cells_share": round(float(args.qlen) * float(all_lens[planted_ids].astype(np.int64).sum()) / total_cells, 4),
                                     "what": "scores at or beyond 32 767 - max(matrix) leave the 16-bit rung for the re-run list; their cells are "
                                             "computed twice and counted once (reference opal_aligner.cpp:203-207)"}} if len(planted_ids) else {}),
                       "parallelism": (f"db-shard x{world} (64-seq groups dealt by residue count), 1 gather per step, issued while the GPU is on the next "
                                       "search; two searches queued per rank, on two streams in turn") if world > 1 else
                                      ("one GPU, results written at their database index; " +
                                       ("two searches queued on the handle (step i enqueued behind step i-1, then step i-1 waited for), on two streams in turn: the grid of "
                                        "search i is released when search i-1 has started its last workgroup and fills the CUs it leaves (OPAL_HIP_PIPELINE=0: one stream)" if args.in_flight == 2
                                        else "one search at a time: enqueued, waited for"))},
            "in_flight": 2 if multi else args.in_flight,
            "settle": {"ms": args.settle_ms, "steps": settle_steps,
                       "what": "untimed steps before the W warm-up steps, so that the timed region does not start on the clock governor's ramp "
                               "(shader clock 2.07 GHz after an idle gap, 2.38 after ~20 ms of back-to-back launches: profiles/r05_clock_c2.txt)"},
            "ranks": {"kernel_ms": {"min": round(min(rank_kms), 4), "max": round(max(rank_kms), 4), "mean": round(float(np.mean(rank_kms)), 4),
                                    "per_rank": [round(v, 4) for v in rank_kms]},
                      "cells_per_step": {"min": min(rank_cells), "max": max(rank_cells)},
                      "imbalance": round(max(rank_cells) / (sum(rank_cells) / len(rank_cells)), 5),
                      "exchange_ms": round(exchange_ms, 4) if exchange_ms is not None else None,
                      "what": "kernel_ms: every rank's mean DP time per step (HIP events on its launch stream); imbalance: largest share of "
                              "cells / mean share (the deal, opal_amd/shard.py); exchange_ms: the gather + the one index_select of a step on "
                              "rank 0's stream (timed events), issued while the GPU already runs the next search"},
            "verified": verified,
            "verify": {"sample": f"{len(pick)} sequences of every rank's shard (fixed seeds) vs oracle/opal_oracle.c, verdicts all-reduced"
                                 + ("; the gathered block vs rank 0's own values" if world > 1 else ""),
                       "score_crc32": zlib.crc32(gscore.tobytes())},
            "roofline": {"bound": "valu-issue", "achieved": round(kernel_gcups, 1), "peak": round(peak, 0), "unit": "GCUPS",
                         "frac": round(kernel_gcups / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_ms": round(kms, 4),
                         "launch_ms": round(launch_kms, 4),
                         "frac_per_launch": round(cells / (launch_kms * 1e-3) / 1e9 / peak, 4),
                         "clock_ghz": round(clock_ghz, 3) if clock_ghz else None,
                         "frac_at_measured_clock": round(kernel_gcups / valu_peak_gcups(instr_per_cell, clock_ghz * 1e9), 4) if clock_ghz else None,
                         "model": f"1024 SIMDs x 2.4 GHz x 64 lanes / ({instr_per_cell:.0f} full-rate 16-bit instr x 2 clk per cell); "
                                  "kernel_ms (N = 1) = device time per search in the queue: HIP events, from the end of the search before to the end of "
                                  "this one, mean over the timed steps but the first -- consecutive searches overlap on two streams (a search's grid is "
                                  "released when the one before has started its last workgroup), so launch_ms, the life of one search's launches from "
                                  "their release to their end (what rocprofv3 --kernel-trace calls its duration), is longer than what the search costs and "
                                  "frac_per_launch prices that -- in a rocprofv3 kernel trace kernel_ms is the average difference between the End_Timestamps of "
                                  "consecutive launches of the dominant kernel and launch_ms their average duration (scripts/trace_period.py, "
                                  "profiles/r06_trace_periods.txt); N > 1: the same per rank; clock_ghz = "
                                  "shader cycles / constant-rate ticks over one wave's lifetime per workgroup, measured in the DP kernels "
                                  "themselves on the last of >= 30 ms of searches queued two deep after the timed region (as the timed steps are); "
                                  "frac_at_measured_clock prices the bound at it",
                         "hbm": hbm},
        }
        if not verified:
            status = 3
        res_host = None
        if world == 1 and not args.no_e2e and total <= 1000000:
            # (a private copy made by THIS thread, the way an application holds the sequences it read: the pages of a tensor that
            # .cpu() filled were first touched by the runtime's staging thread, wherever that ran)
            res_host = d_res.cpu().numpy().copy()
            e2e_ms, same_ms, e2e_ok = e2e_call(query, res_host, offs, matrix, args.gap_open, args.gap_ext, args.search_type, mode, gscore,
                                               calls=6 if total <= 200000 else 3)
            res_line["e2e"] = {"ms": round(e2e_ms, 3), "gcups": round(cells / (e2e_ms * 1e-3) / 1e9, 1), "scores_match": e2e_ok,
                               "same_database_ms": round(same_ms, 3), "same_database_gcups": round(cells / (same_ms * 1e-3) / 1e9, 1),
                               "what": "opalSearchDatabase (host pointers in, result structs out), steady clock, best of the timed calls. "
                                       "ms: a database the library has not seen (before every timed call it is given another one of the "
                                       "same shape); same_database_ms: the same database again (residues compared while gathered, "
                                       "nothing uploaded)"}
            if not e2e_ok:
                res_line["verified"] = False
                status = 3
        if world > 1 and not args.no_e2e and total <= 1000000:
            # N > 1: the same call with the library itself sharding the database over the node's GPUs (OPAL_HIP_DEVICES: one
            # host thread and one cached context per GPU, the host is the gather root -- the results live in host structs).
            # The other ranks wait on the rendezvous store, on the CPU: their GPUs are free for this process.
            try:
                hres, hoffs, _ = synth.database_subset(np.arange(total, dtype=np.int64), 42, lens=all_lens, stride=stride)
                devs = ",".join(["0"] * world) if os.environ.get("BENCH_SHARE_GPU") == "1" else "all"
                opal_amd.set_option("OPAL_HIP_DEVICES", devs)
                e2e_ms, same_ms, e2e_ok = e2e_call(query, hres, hoffs, matrix, args.gap_open, args.gap_ext, args.search_type, mode, gscore, calls=3)
                res_line["e2e"] = {"ms": round(e2e_ms, 3), "gcups": round(total_cells / (e2e_ms * 1e-3) / 1e9, 1), "scores_match": e2e_ok,
                                   "same_database_ms": round(same_ms, 3), "same_database_gcups": round(total_cells / (same_ms * 1e-3) / 1e9, 1),
                                   "what": f"ONE opalSearchDatabase call from this process over {world} GPUs (OPAL_HIP_DEVICES={devs}: "
                                           "the library deals 64-sequence groups by residue count, one host thread + cached context per GPU, "
                                           "results merged on the host), whole database, steady clock, best of 3; ms: a database the library has "
                                           "not seen; same_database_ms: the same database again"}
                if not e2e_ok:
                    res_line["verified"] = False
                    status = 3
                # ... and the prepared multi-GPU handle (opalx_multi_*: shards stay packed on their GPUs, results gathered on
                # devices[0] and permuted into database order there), once per gather kind: peer copies, one ncclGather per search
                from opal_amd.api import PreparedMultiDb
                devlist = [0] * world if os.environ.get("BENCH_SHARE_GPU") == "1" else list(range(world))
                handle = {}
                for kind in ("peer", "rccl"):
                    opal_amd.set_option("OPAL_HIP_MULTI_GATHER", kind)
                    mh = PreparedMultiDb(hres, hoffs, devlist)
                    L = _lib.lib()
                    L.opalx_multi_gather_kind.restype = C.c_int
                    L.opalx_multi_gather_kind.argtypes = [C.c_void_p]
                    used = "rccl" if L.opalx_multi_gather_kind(mh._h) == 1 else "peer"
                    mo = torch.full((total,), -7, dtype=torch.int32, device=dev)
                    torch.cuda.synchronize()
                    ts = []
                    for _ in range(12):
                        t1 = time.perf_counter()
                        rcm, _k = mh.search_device(query, args.gap_open, args.gap_ext, matrix, 0, mode, mo.data_ptr())
                        ts.append((time.perf_counter() - t1) * 1e3)
                    okm = bool(rcm == 0 and (mo.cpu().numpy() == gscore).all())
                    handle[kind] = {"gather_used": used, "ms": round(float(np.median(ts[2:])), 3),
                                    "gcups": round(total_cells / (float(np.median(ts[2:])) * 1e-3) / 1e9, 1), "scores_match": okm}
                    mh.close()
                    if not okm:
                        res_line["verified"] = False
                        status = 3
                res_line["multi_handle"] = dict(handle, what=f"opalx_multi_search_device from this process over {world} GPUs, whole database "
                                                             "(scores in database order on devices[0]), wall clock per call, median of 10")
            except Exception as e:  # noqa: BLE001 -- the line must still be printed
                res_line.setdefault("e2e", {"error": repr(e)[:300]})
                res_line["multi_handle_error"] = repr(e)[:300]
            finally:
                opal_amd.set_option("OPAL_HIP_DEVICES", None)
                opal_amd.set_option("OPAL_HIP_MULTI_GATHER", None)
        if world == 1 and not args.no_e2e and args.workload == "c2" and (args.mode, args.search_type) == ("SW", 0):
            # SURVEY.md 8d asks for both gap settings of the workload; the other one is timed here, outside the
            # K timed steps (11/1: nothing exceeds the 8-bit range of the reference; 3/1: the reference's CLI default)
            other = 3 if args.gap_open == 11 else 11
            ks = []
            for _ in range(10):
                rc, k = pdb.search_device(query, other, args.gap_ext, matrix, 0, mode, outs[0].data_ptr())
                ks.append(k)
            res_line["also"] = {f"kernel_gcups_gaps_{other}_{args.gap_ext}": round(cells / (float(np.median(ks)) * 1e-3) / 1e9, 1)}
        if args.workload == "sprot" and world == 1:
            # the published table (reference aligner_comparison.md:27-34): seconds per search for each of its four queries,
            # next to the reference's published single-thread AVX2 seconds on an i7-4770K (context only: other hardware)
            # and, with the reference build present, its time on THIS box (all threads it may keep busy; 1 thread on a sample)
            if res_host is None:
                res_host = d_res.cpu().numpy()
            ref = RefCpu() if not args.no_cpu_baseline else None
            quota = cpu_quota_cores()
            table = {}
            for name in ("O74807", "P19930", "Q3ZAI3", "P18080"):
                qq = read_fasta(os.path.join(ROOT, "tests", "golden", "data", f"query_{name}.fasta"), alphabet)[0]
                ks, ts = [], []
                for _ in range(5):
                    t1 = time.perf_counter()
                    rc, k = pdb.search_device(qq, args.gap_open, args.gap_ext, matrix, 0, MODES["SW"], outs[0].data_ptr())
                    ts.append(time.perf_counter() - t1)
                    ks.append(k)
                row = {"query_length": len(qq), "gpu_search_s": round(float(np.median(ts)), 5), "gpu_kernel_ms": round(float(np.median(ks)), 3),
                       "gpu_kernel_gcups": round(len(qq) * float(offs[-1]) / (float(np.median(ks)) * 1e-3) / 1e9, 1),
                       "published_opal_avx2_i7_4770K_1thread_s": PUBLISHED_S[name]}
                if ref is not None and ref.ok:
                    nt = max(1, min(os.cpu_count() or 1, int(quota + 0.999) if quota else 256, 256))
                    row[f"reference_avx2_this_box_{nt}_threads_s"] = round(ref.run(qq, res_host, offs, n, matrix, args.gap_open, args.gap_ext, MODES["SW"], nt), 3)
                    n1 = max(1, n // 64)
                    t_one = ref.run(qq, res_host, offs, n1, matrix, args.gap_open, args.gap_ext, MODES["SW"], 1)
                    row["reference_avx2_this_box_1_thread_s_extrapolated"] = round(t_one * float(offs[-1]) / float(offs[n1]), 2)
                table[name] = row
            res_line["published_shape"] = table
        if not args.no_cpu_baseline and world == 1:
            if res_host is None:
                res_host = d_res.cpu().numpy()
            res_line["cpu_baseline"] = cpu_baseline(query, res_host, offs, matrix, args.gap_open, args.gap_ext, mode)
    if multi:
        # (rank 0 may still be timing the in-library N-GPU call above: the others wait on the store, not on their GPUs)
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            store.set("bench_rank0_done", "1")
        else:
            store.wait(["bench_rank0_done"])
        dist.barrier()
    # ---- the default run also carries north_star's strong-scaling table (BASELINE.json configs[3] and [4]), so that the driver's
    # N = 1, 2, 4, 8 runs give the C4 / C5 curve as well: P18080 and a 2000-aa query x 5 000 000 sequences IN TOTAL, dealt to the N
    # GPUs, 5 timed steps each, outside the headline's timed region. (BENCH_ALSO_TOTAL: a smaller total, for rehearsals on one GPU.)
    default_run = (args.workload == "c2" and args.seqs == 100000 and args.total_seqs is None and args.qlen == 300 and args.mode == "SW"
                   and args.search_type == 0 and not args.query_fasta)
    if default_run and not args.no_also and os.environ.get("BENCH_FORCE_DIST") != "1":
        also_total = int(os.environ.get("BENCH_ALSO_TOTAL", "5000000"))
        pdb.close()
        del d_res, outs, plan
        torch.cuda.empty_cache()
        legs = {}
        for leg, q in (("c4", read_fasta(os.path.join(ROOT, "tests", "golden", "data", "query_P18080.fasta"), alphabet)[0]),
                       ("c5", synth.query(2000, seed=7))):
            try:
                r = strong_leg(leg, q, also_total, rank, world, multi, backend, dev, matrix, 11, 1, MODES["SW"])
            except Exception as e:  # noqa: BLE001 -- the headline line must still be printed (a rank that fails here ends the group: the launcher reports it)
                r = {"error": repr(e)[:300]}
                if multi:
                    raise
            if rank == 0:
                legs[leg] = r  # (a leg that fails or does not verify says so in its own entry: the exit status is the headline's)
        if rank == 0:
            res_line.setdefault("also", {}).update(legs)
    if multi:
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res_line) + "\n").encode())
    sys.exit(status)


if __name__ == "__main__":
    main()
