import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """Native pieces: the C-ABI library + CLI (hipcc, cross-compiles) and the oracle (g++)."""
    import __graft_entry__ as ge
    ge.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    from oracle import dtc_oracle
    return dtc_oracle.Oracle()


@pytest.fixture(scope="session")
def lib(built):
    from deltapq_amd import _lib
    return _lib.load()


@pytest.fixture(scope="session")
def codebook():
    from deltapq_amd import synth
    return synth.make_codebook(8, 256, 16, seed=0)


def make_case(n, seed, mean_diffs=3.0, dup_heavy=False):
    """(tree, payload, n_bytes) of a seeded synthetic DeltaTree."""
    from deltapq_amd import synth
    tree = synth.synth_tree(n, 8, seed=seed, mean_diffs=0.35 if dup_heavy else mean_diffs)
    payload, nb = synth.encode_dtc(tree)
    return tree, payload, nb


def oracle_topk(oracle, payload, n, cb, queries, k):
    """[(ids, dists, all_dists)] per query from the oracle."""
    out = []
    for q in queries:
        lut = oracle.build_lut(cb, q)
        ids, d, alld, _ = oracle.scan_lut(payload, n, lut, k, want_all=True)
        out.append((ids, d, alld))
    return out


def assert_parity(ids, dists, ref, n):
    from oracle.dtc_oracle import tie_aware_equal
    for i, (oi, od, alld) in enumerate(ref):
        ok, msg = tie_aware_equal(ids[i], dists[i], oi, od, alld, n)
        assert ok, "query %d: %s\n got %s %s\n ref %s %s" % (i, msg, ids[i][:8], dists[i][:4], oi[:8], od[:4])
