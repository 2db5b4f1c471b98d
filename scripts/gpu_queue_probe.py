"""Do the side launches of a search share a hardware queue with something else? Runs the C2 search in fresh processes
with extra user streams created first (what a framework around the library may do) and prints the kernel time."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import numpy as np, torch
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50
extra = [torch.cuda.Stream() for _ in range(int(os.environ.get('EXTRA_STREAMS', '0')))]
for st in extra:
    with torch.cuda.stream(st): torch.zeros(8, device='cuda')
res, offs, lens = synth.database(100000, seed=42); q = synth.query(300, seed=7); m = blosum50()[1]
pdb = PreparedDb(res, offs); score = torch.zeros(100000, dtype=torch.int32, device='cuda')
ks = []
for it in range(15):
    rc, kms = pdb.search_device(q, 11, 1, m, 0, 3, score.data_ptr())
    if it >= 3: ks.append(kms)
print('extra streams', len(extra), 'side prio', os.environ.get('OPAL_HIP_SIDE_STREAM_PRIO', 'high'), 'sides', os.environ.get('OPAL_HIP_SIDE_STREAMS', '2'), 'kernel_ms median %%.4f' %% np.median(ks))
""" % (ROOT, ROOT)
for extra in (0, 1, 2, 3, 5):
    for prio in ("1", "0"):
        for sides in ("2", "3"):
            env = dict(os.environ, EXTRA_STREAMS=str(extra), OPAL_HIP_SIDE_STREAM_PRIO=prio, OPAL_HIP_SIDE_STREAMS=sides)
            out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
            print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
