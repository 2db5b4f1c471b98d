"""Deterministic synthetic protein database / query generator.

Counter-based splitmix64 (no libc rand, no numpy RandomState): element i of a
stream with seed s is mix(s + (i+1)*GOLDEN), so generation is vectorised and
any slice can be produced independently (ranks of a multi-GPU run generate
disjoint shards of the same database without talking to each other).

Lengths: L = clamp(30, 5000, round(exp(5.70 + 0.55*z))), z = Box-Muller normal
-> median ~299, mean ~348 (UniProt-like log-normal, BASELINE.json "~350 aa").
Residues: the 20 standard amino acids drawn from Swiss-Prot background
frequencies, encoded in BLOSUM50 alphabet order ``ARNDCQEGHILKMFPSTWYVBZX*``.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

# percent, in the order of _AA below (Swiss-Prot release statistics, rounded as in SURVEY.md 8d)
_AA = "LAGVESIKRDTPNQFYMHCW"
_PCT = [9.9, 8.3, 7.1, 6.9, 6.8, 6.6, 5.9, 5.8, 5.5, 5.5, 5.3, 4.7, 4.1, 3.9, 3.9, 2.9, 2.4, 2.3, 1.4, 1.1]
_BLOSUM_ORDER = "ARNDCQEGHILKMFPSTWYVBZX*"


def splitmix64(seed, start, count):
    """uint64[count]: outputs start .. start+count-1 of the stream `seed`."""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _uniform(seed, start, count):
    """float64 in (0,1): top 53 bits, offset by half an ulp so log() is finite."""
    return ((splitmix64(seed, start, count) >> np.uint64(11)).astype(np.float64) + 0.5) / 9007199254740992.0


def _residue_table():
    cdf = np.cumsum(np.array(_PCT, dtype=np.float64))
    cdf /= cdf[-1]
    codes = np.array([_BLOSUM_ORDER.index(a) for a in _AA], dtype=np.uint8)
    return cdf, codes


SPROT_SEQS = 541762     # UniProtKB/Swiss-Prot as the reference was published on it (reference aligner_comparison.md:6)
SPROT_MAX_LEN = 35213   # its longest entry (titin)
SPROT_STRIDE = 36000    # residue-stream stride of the "sprot" distribution (see database_subset)


def lengths(seed, first, count, dist="c2"):
    """int32[count]: lengths of sequences first .. first+count-1 of database `seed`.

    dist "c2" (SURVEY.md 8d): log-normal clamped to [30, 5000].
    dist "sprot": the same body with the long tail a real protein database has -- 0.2 % of the sequences are drawn from
    a Pareto tail L = 2000 * (1-u)^(-1/1.6), capped at 35 213 (on 541 762 sequences: ~250 beyond 5 000 residues, ~80
    beyond 10 000, ~10 at the cap)."""
    u1 = _uniform(seed, 2 * first, 2 * count)
    z = np.sqrt(-2.0 * np.log(u1[0::2])) * np.cos(2.0 * np.pi * u1[1::2])
    body = np.clip(np.rint(np.exp(5.70 + 0.55 * z)), 30, 5000)
    if dist == "c2":
        return body.astype(np.int32)
    if dist != "sprot":
        raise ValueError(dist)
    ut = _uniform(seed ^ 0x7A11, 2 * first, 2 * count)
    tail = np.minimum(np.rint(2000.0 * (1.0 - ut[1::2]) ** (-1.0 / 1.6)), SPROT_MAX_LEN)
    return np.where(ut[0::2] < 0.002, tail, body).astype(np.int32)


def _residues(seed, start, count):
    cdf, codes = _residue_table()
    u = _uniform(seed ^ 0x5851F42D4C957F2D, start, count)
    return codes[np.minimum(np.searchsorted(cdf, u, side="right"), len(codes) - 1)]


def database(num_seqs, seed=42, first=0):
    """Sequences first .. first+num_seqs-1 of the synthetic database `seed`.

    Returns (residues:uint8[sum L], offsets:int64[num_seqs+1], lens:int32[num_seqs]); sequence j is
    residues[offsets[j]:offsets[j+1]]. Residue k of sequence g (global index) is element g*5000 + k
    of the residue stream, so shards are independent of how the database is cut.
    """
    lens = lengths(seed, first, num_seqs)
    offsets = np.zeros(num_seqs + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    total = int(offsets[-1])
    cdf, codes = _residue_table()
    out = np.empty(total, dtype=np.uint8)
    rseed = np.uint64(seed ^ 0x5851F42D4C957F2D)
    batch = 512  # sequences per step: temporaries stay a few MB (fresh pages are slow to fault in)
    for a in range(0, num_seqs, batch):
        b = min(a + batch, num_seqs)
        ln = lens[a:b]
        n = int(offsets[b] - offsets[a])
        seq_of = np.repeat(np.arange(a, b, dtype=np.int64), ln)
        pos = np.arange(n, dtype=np.int64) - (offsets[a:b] - offsets[a])[seq_of - a]
        c = ((seq_of + first) * 5000 + pos).astype(np.uint64)
        with np.errstate(over="ignore"):
            z = rseed + (c + np.uint64(1)) * _GOLDEN
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            z = z ^ (z >> np.uint64(31))
        u = ((z >> np.uint64(11)).astype(np.float64) + 0.5) / 9007199254740992.0
        out[offsets[a]:offsets[b]] = codes[np.minimum(np.searchsorted(cdf, u, side="right"), len(codes) - 1)]
    return out, offsets, lens


def database_subset(ids, seed=42, lens=None, stride=5000):
    """The sequences with GLOBAL indices `ids` (any order, any subset) of database `seed`, as
    (residues, offsets, lens). Used by the multi-GPU sharding (opal_amd/shard.py): a rank owns a
    length-balanced, non-contiguous set of sequences and generates exactly those.
    `stride`: residue k of sequence g is element g*stride + k of the residue stream (5000 for the clamped "c2"
    lengths; SPROT_STRIDE for the "sprot" lengths, which must be passed in `lens`)."""
    ids = np.asarray(ids, dtype=np.int64)
    n = len(ids)
    if lens is None:
        lens = np.array([lengths(seed, int(i), 1)[0] for i in ids], dtype=np.int32) if n < 64 else _lengths_of(seed, ids)
    lens = np.asarray(lens, dtype=np.int32)
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    cdf, codes = _residue_table()
    out = np.empty(int(offsets[-1]), dtype=np.uint8)
    rseed = np.uint64(seed ^ 0x5851F42D4C957F2D)
    batch = 512
    for a in range(0, n, batch):
        b = min(a + batch, n)
        m = int(offsets[b] - offsets[a])
        loc = np.repeat(np.arange(a, b, dtype=np.int64), lens[a:b])
        pos = np.arange(m, dtype=np.int64) - (offsets[a:b] - offsets[a])[loc - a]
        c = (ids[loc] * stride + pos).astype(np.uint64)
        with np.errstate(over="ignore"):
            z = rseed + (c + np.uint64(1)) * _GOLDEN
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            z = z ^ (z >> np.uint64(31))
        u = ((z >> np.uint64(11)).astype(np.float64) + 0.5) / 9007199254740992.0
        out[offsets[a]:offsets[b]] = codes[np.minimum(np.searchsorted(cdf, u, side="right"), len(codes) - 1)]
    return out, offsets, lens


def _lengths_of(seed, ids):
    """Lengths of the sequences `ids` (computed over the covering contiguous range, in blocks)."""
    ids = np.asarray(ids, dtype=np.int64)
    out = np.empty(len(ids), dtype=np.int32)
    lo, hi = int(ids.min()), int(ids.max()) + 1
    block = 1 << 20
    for a in range(lo, hi, block):
        b = min(a + block, hi)
        sel = np.nonzero((ids >= a) & (ids < b))[0]
        if len(sel):
            out[sel] = lengths(seed, a, b - a)[ids[sel] - a]
    return out


def residues_device(ids, lens, seed, device, stride=5000):
    """Same residues as database_subset(ids)[0], generated ON THE GPU with torch integer arithmetic
    (bench.py: a 5-million-sequence shard is 1.7 GB of residues; numpy needs minutes for that).
    Lengths come from the host (float64 log/cos are not guaranteed bit-identical on the device); the
    residue stream is pure 64-bit integer arithmetic plus one exact float64 division, so the result is
    bit-identical to the numpy generator (tests/test_gpu_fullsize.py checks a sample).
    Returns (uint8 tensor[sum lens] on `device`, int64 numpy offsets)."""
    import torch
    ids_t = torch.as_tensor(np.asarray(ids, dtype=np.int64), device=device)
    lens_np = np.asarray(lens, dtype=np.int64)
    offsets = np.zeros(len(lens_np) + 1, dtype=np.int64)
    np.cumsum(lens_np, out=offsets[1:])
    total = int(offsets[-1])
    out = torch.empty(total, dtype=torch.uint8, device=device)
    cdf, codes = _residue_table()
    cdf_t = torch.as_tensor(cdf, device=device)
    codes_t = torch.as_tensor(codes, device=device)
    lens_t = torch.as_tensor(lens_np, device=device)
    offs_t = torch.as_tensor(offsets, device=device)

    def s64(v):  # two's-complement int64 view of a uint64 constant
        v = int(v) & 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(x, k):  # logical shift right on int64
        return (x >> k) & ((1 << (64 - k)) - 1)

    G, M1, M2 = s64(_GOLDEN), s64(_M1), s64(_M2)
    rseed = s64(seed ^ 0x5851F42D4C957F2D)
    batch = 1 << 16  # sequences per step (temporaries: a few hundred MB at most)
    for a in range(0, len(lens_np), batch):
        b = min(a + batch, len(lens_np))
        m = int(offsets[b] - offsets[a])
        if m == 0:
            continue
        loc = torch.repeat_interleave(torch.arange(a, b, device=device), lens_t[a:b])
        pos = torch.arange(m, device=device, dtype=torch.int64) - (offs_t[loc] - int(offsets[a]))
        c = ids_t[loc] * stride + pos
        z = rseed + (c + 1) * G
        z = (z ^ lsr(z, 30)) * M1
        z = (z ^ lsr(z, 27)) * M2
        z = z ^ lsr(z, 31)
        u = (lsr(z, 11).to(torch.float64) + 0.5) / 9007199254740992.0
        idx = torch.clamp(torch.searchsorted(cdf_t, u, right=True), max=len(codes) - 1)
        out[int(offsets[a]):int(offsets[b])] = codes_t[idx]
    return out, offsets


def query(length, seed=7):
    """uint8[length] synthetic query drawn from the same background distribution."""
    return _residues(seed, 0, length)
