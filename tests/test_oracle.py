"""CPU tests of the oracle itself (test infrastructure): the C++ restatement
against the independent Python restatement, against committed golden vectors,
and against a hand-built stream whose answer is derived by hand from the
format definition (deltapq_create_approx_tree.h:1771-1826, 2866-2975)."""
import os

import numpy as np
import pytest

from conftest import make_case

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_lut_matches_python_restatement(oracle, codebook):
    from deltapq_amd import synth
    from oracle import dtc_oracle as O
    q = synth.make_queries(1, 128, seed=3)[0]
    sub = codebook[:, :8, :]                                   # 8 centroids per sub-space keeps Python fast
    lut_c = oracle.build_lut(sub, q)
    lut_p = O.py_build_lut(sub, q)
    assert np.array_equal(lut_c.view(np.uint32), lut_p.view(np.uint32))
    # fractional inputs exercise the fp32-subtract / fp64-square / float+=double chain
    rng = np.random.default_rng(0)
    cbf = rng.normal(0, 30, size=(8, 8, 16)).astype(np.float32)
    qf = rng.normal(0, 30, size=128).astype(np.float32)
    assert np.array_equal(oracle.build_lut(cbf, qf).view(np.uint32), O.py_build_lut(cbf, qf).view(np.uint32))
    # and it is NOT the same as a pure fp32 or pure fp64 accumulation (the emulation matters)
    d = (cbf - qf.reshape(8, 1, 16)).astype(np.float32)
    pure64 = (d.astype(np.float64) ** 2).sum(-1).astype(np.float32)
    assert np.max(np.abs(pure64 - oracle.build_lut(cbf, qf)) / pure64) < 1e-6


@pytest.mark.parametrize("n,k", [(1, 1), (2, 2), (3, 2), (64, 10), (65, 10), (501, 50), (1000, 100)])
def test_cpp_vs_python_scan(oracle, codebook, n, k):
    from deltapq_amd import synth
    from oracle import dtc_oracle as O
    tree, payload, nb = make_case(n, seed=100 + n)
    codes = synth.decode_tree_codes(tree)
    for q in synth.make_queries(3, 128, seed=n):
        lut = oracle.build_lut(codebook, q)
        ids, d, alld, allc = oracle.scan_lut(payload, n, lut, k, want_all=True)
        pids, pd, palld, pallc, off = O.py_scan(payload, n, lut, k)
        assert off == nb                                        # whole stream consumed
        assert np.array_equal(allc, codes) and np.array_equal(pallc, codes)   # lossless
        assert np.array_equal(alld.view(np.uint32), palld.view(np.uint32))
        assert np.array_equal(ids, pids)                        # same libstdc++ heap order
        assert np.array_equal(d.view(np.uint32), pd.view(np.uint32))
        assert np.all(np.diff(d) >= 0)
        # fp64 sum of the 8 fp32 entries, rounded once == the incremental fp64 stack
        s = sum(lut[m, codes[:, m]].astype(np.float64) for m in range(8))
        assert np.array_equal(s.astype(np.float32).view(np.uint32), alld.view(np.uint32))


def test_even_n_reports_last_node_as_n(oracle, codebook):
    """h:2949, 2970: the trailing node of an even-N index is pushed with id i+1 == N."""
    from deltapq_amd import synth
    n = 10
    tree, payload, _ = make_case(n, seed=5)
    q = synth.make_queries(1, 128, seed=6)[0]
    ids, _ = oracle.query_in_memory(payload, n, codebook, q, n)
    assert sorted(ids.tolist()) == [0, 1, 2, 3, 4, 5, 6, 7, 8, 10]
    tree, payload, _ = make_case(11, seed=5)
    ids, _ = oracle.query_in_memory(payload, 11, codebook, q, 11)
    assert sorted(ids.tolist()) == list(range(11))


def test_hand_built_stream(oracle):
    """A 5-node stream written byte by byte from the format definition, with the
    expected codes and distances worked out by hand (K = 4 toy codebook)."""
    M, K, Ds = 8, 4, 1
    cb = np.zeros((M, K, Ds), np.float32)
    for m in range(M):
        cb[m, :, 0] = [0.0, 1.0, 2.0, 3.0]                     # centroid k of every sub-space is the scalar k
    query = np.zeros(M, np.float32)                            # T[m][k] = k^2
    root = [1, 1, 1, 1, 1, 1, 1, 1]                            # dist 8
    payload = bytes(root
                    + [0x21]                                   # pair byte: node1 depth 1, node2 depth 2
                    + [0b00000001, 2]                          # node1: position 0 -> 2   => code 2,1,1,1,1,1,1,1  dist 11
                    + [0b10000010, 0, 3]                       # node2 (child of node1): pos1 -> 0, pos7 -> 3 => 2,0,1,1,1,1,1,3 dist 18
                    + [0x11]                                   # pair byte: node3 depth 1, node4 depth 1
                    + [0b00000000]                             # node3: duplicate of the root, dist 8
                    + [0b11111111, 0, 0, 0, 0, 0, 0, 0, 0])    # node4: all zeros, dist 0
    arr = np.frombuffer(payload, np.uint8)
    lut = oracle.build_lut(cb, query)
    assert np.array_equal(lut, np.tile(np.array([0, 1, 4, 9], np.float32), (8, 1)))
    ids, d, alld, allc = oracle.scan_lut(arr, 5, lut, 5, want_all=True)
    assert allc.tolist() == [root, [2, 1, 1, 1, 1, 1, 1, 1], [2, 0, 1, 1, 1, 1, 1, 3], root, [0] * 8]
    assert alld.tolist() == [8.0, 11.0, 18.0, 8.0, 0.0]
    assert d.tolist() == [0.0, 8.0, 8.0, 11.0, 18.0]
    assert ids[0] == 4 and set(ids[1:3].tolist()) == {0, 3} and ids[3:].tolist() == [1, 2]
    # n_bytes formula of the writer (h:1765): 8 + n_diffs + (3*(N-1)+1)//2
    assert len(payload) == 8 + (1 + 2 + 0 + 8) + (3 * 4 + 1) // 2


@pytest.mark.parametrize("name", ["small_even", "small_odd", "dup_heavy"])
def test_golden_vectors(oracle, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n, k = int(g["n_codes"]), int(g["top_k"])
    for i, q in enumerate(g["queries"]):
        ids, d = oracle.query_in_memory(g["payload"], n, g["codebook"], q, k)
        assert np.array_equal(ids, g["ids"][i])
        assert np.array_equal(d.view(np.uint32), g["dist_bits"][i])


def test_o_direct_variant_equals_in_memory(oracle, codebook, tmp_path):
    """h:2805-2984 (4 KB block reads) and h:3731-3892 (in memory) are the same arithmetic."""
    from deltapq_amd import synth
    n = 5000                                                   # > 4 KB of payload: crosses block boundaries
    tree, payload, nb = make_case(n, seed=9)
    path = str(tmp_path / "M8K256_Approx_compressed_codes_opt_N5000")
    synth.write_dtc_file(path, n, payload)
    assert nb > 3 * 4096
    for q in synth.make_queries(3, 128, seed=10):
        a = oracle.query_in_memory(payload, n, codebook, q, 25)
        b = oracle.query_o_direct(path, n, codebook, q, 25)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def test_topk_larger_than_n_is_rejected(oracle, codebook):
    from deltapq_amd import synth
    tree, payload, _ = make_case(5, seed=1)
    with pytest.raises(ValueError):
        oracle.query_in_memory(payload, 5, codebook, synth.make_queries(1, 128)[0], 6)
