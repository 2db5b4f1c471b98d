import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import opal_amd
from opal_amd import synth
from opal_amd.api import PreparedDb
from opal_amd.score_matrix import blosum50
import test_gpu_pipeline as T
n = 100000
res, offs, lens = synth.database(n, seed=42)
m = blosum50()[1]
kinds = [(synth.query(300, seed=7), 11, 1, m, 0, 3), (synth.query(280, seed=17), 11, 1, m, 1, 1)]
jobs = [kinds[i & 1] for i in range(40)]
def run(tag, pdb):
    got, kms, per = T._queue(pdb, n, jobs, 2)
    print(tag, "kms", np.round(kms[5:9], 3), "per", np.round(per[5:9], 3), "ratio", round(np.mean(per[5:]) / np.mean(kms[5:]), 3), "timeouts", T._timeouts(), flush=True)
pdb = PreparedDb(res, offs)
run("fresh handle", pdb)
want = [T._sync(pdb, n, *k) for k in kinds]
run("after two sync searches", pdb)
run("again", pdb)
opal_amd.set_option("OPAL_HIP_PIPELINE", "1")
run("option set to 1", pdb)
opal_amd.set_option("OPAL_HIP_PIPELINE", "0")
run("option set to 0", pdb)
opal_amd.set_option("OPAL_HIP_PIPELINE", "1")
run("option set to 1 again", pdb)
pdb2 = PreparedDb(res, offs)
want = [T._sync(pdb2, n, *k) for k in kinds]
run("second handle after sync", pdb2)
