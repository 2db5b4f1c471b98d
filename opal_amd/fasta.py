"""FASTA -> alphabet-index arrays, with the reference CLI's conventions
(reference src/opal_aligner.cpp:247-301): header lines start with '>', CR/LF are
skipped, every other byte is one residue mapped through the letter table."""
import numpy as np

from .score_matrix import letter_index


def read_fasta(path, alphabet):
    """Return a list of uint8 arrays (alphabet indices), one per record with at least one residue."""
    table = letter_index(alphabet)
    seqs, cur = [], None
    with open(path, "rb") as fh:
        for line in fh:
            if line.startswith(b">"):
                cur = None
                continue
            body = line.replace(b"\r", b"").replace(b"\n", b"")
            if not body:
                continue
            if cur is None:
                cur = []
                seqs.append(cur)
            cur.append(np.frombuffer(body, dtype=np.uint8))
    return [table[np.concatenate(parts)] for parts in seqs]
