"""Generates tests/golden/*.npz (run from the repo root: python tests/golden/make_golden.py).

PARITY UNPINNED: these vectors are produced by THIS repo's oracle
(oracle/dtc_oracle.cpp), not by the reference -- the reference cannot be built
in this image (OpenCV missing) and ships no fixtures.  They pin the oracle and
the HIP path against regressions; the hand-derived case in
tests/test_oracle.py::test_hand_built_stream pins the format reading itself.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deltapq_amd import synth          # noqa: E402
from oracle import dtc_oracle as O     # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def make(name, n, nq, k, seed, mean_diffs):
    orc = O.Oracle()
    cb = synth.make_codebook(8, 256, 16, seed)
    # the codebook goes through the reference's text format (6 significant digits)
    tmp = os.path.join(OUT, "_tmp_codewords.txt")
    synth.write_codewords_txt(tmp, cb)
    cb = synth.read_codewords_txt(tmp)
    os.remove(tmp)
    qs = synth.make_queries(nq, 128, seed + 1)
    tree = synth.synth_tree(n, 8, seed + 2, mean_diffs=mean_diffs)
    payload, nb = synth.encode_dtc(tree)
    ids = np.zeros((nq, k), np.int32)
    dists = np.zeros((nq, k), np.float32)
    for i in range(nq):
        ids[i], dists[i] = orc.query_in_memory(payload, n, cb, qs[i], k)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), codebook=cb,
                        queries=qs, payload=payload, n_codes=n, top_k=k, ids=ids, dist_bits=dists.view(np.uint32))
    print(name, "n", n, "bytes", nb, "nq", nq, "k", k)


if __name__ == "__main__":
    make("small_even", 2000, 8, 10, 11, 3.0)       # even N: last node reported as N
    make("small_odd", 2001, 8, 10, 21, 3.0)
    make("dup_heavy", 1500, 8, 20, 31, 0.35)       # many zero-diff children -> exact ties
