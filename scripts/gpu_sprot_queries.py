"""The published-shape workload (bench.py --workload sprot) query by query: kernel ms per search for the four published
queries (reference aligner_comparison.md:27-34) on the 541 762-sequence synthetic Swiss-Prot shape, with the scheduler's
debug lines of one search each (OPAL_HIP_DEBUG=1 in the environment) -- which kernels ran, where the swapped kernel took
the tail. Usage: python scripts/gpu_sprot_queries.py [names...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opal_amd import synth  # noqa: E402
from opal_amd.api import MODES, PreparedDb  # noqa: E402
from opal_amd.fasta import read_fasta  # noqa: E402
from opal_amd.score_matrix import blosum50  # noqa: E402

names = sys.argv[1:] or ["O74807", "P19930", "Q3ZAI3", "P18080"]
dev = torch.device("cuda:0")
alphabet, matrix = blosum50()
n = synth.SPROT_SEQS
lens = synth.lengths(42, 0, n, "sprot")
d_res, offs = synth.residues_device(np.arange(n, dtype=np.int64), lens, 42, dev, stride=synth.SPROT_STRIDE)
torch.cuda.synchronize()
pdb = PreparedDb(None, offs, device_ptr=d_res.data_ptr())
out = torch.zeros(n, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
reps = int(os.environ.get("REPS", "7"))
ST = int(os.environ.get("ST", "0"))  # 1: with end locations
MODE = os.environ.get("MODE", "SW")
e1 = torch.zeros(n, dtype=torch.int32, device=dev); e2 = torch.zeros(n, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for name in names:
    # (a name that is a number: a synthetic query of that length)
    q = synth.query(int(name), seed=7) if name.isdigit() else read_fasta(os.path.join(ROOT, "tests", "golden", "data", f"query_{name}.fasta"), alphabet)[0]
    ks = []
    for _ in range(reps):
        rc, k = pdb.search_device(q, 3, 1, matrix, ST, MODES[MODE], out.data_ptr(), e1.data_ptr() if ST else 0, e2.data_ptr() if ST else 0)
        assert rc == 0
        ks.append(k)
    cells = len(q) * float(offs[-1])
    med = float(np.median(ks))
    print(f"{MODE} {name} Q={len(q)}: kernel ms median {med:.3f} min {min(ks):.3f} -> {cells / med / 1e6:.0f} GCUPS, frac of the {9 if MODE == 'SW' else 6}-instr bound {cells / med / 1e6 / (8738 if MODE == 'SW' else 13107):.3f}; checksum {int(out.sum().item())}" + (f" end checksum {int(e1.sum().item())} {int(e2.sum().item())}" if ST else ""), flush=True)
