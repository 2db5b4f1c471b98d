"""Load the committed golden vectors (tests/golden/golden_small.json + golden_synth.npz)."""
import json
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_all():
    """-> {case_name: {"input_sha1": str, "runs": {mode: {"rc","score","endQ","endT"}}, ["char_sw"]}}"""
    with open(os.path.join(HERE, "golden_small.json")) as fh:
        out = json.load(fh)
    for case in out.values():
        for run in case["runs"].values():
            for k in ("score", "endQ", "endT"):
                run[k] = np.array(run[k], dtype=np.int32)
    npz = np.load(os.path.join(HERE, "golden_synth.npz"))
    names = sorted({k.split("/")[0] for k in npz.files})
    for name in names:
        meta = json.loads(bytes(npz[name + "/meta"]).decode())
        runs = {}
        for mode, rc in meta["rc"].items():
            runs[mode] = {"rc": rc, **{k: npz[f"{name}/{mode}/{k}"] for k in ("score", "endQ", "endT")}}
        out[name] = {"name": name, "input_sha1": meta["input_sha1"], "runs": runs}
    return out


def input_sha1(query, seqs):
    import hashlib
    h = hashlib.sha1()
    for s in [query] + list(seqs):
        h.update(np.ascontiguousarray(s, dtype=np.uint8).tobytes())
        h.update(b"|")
    return h.hexdigest()
