"""Hand-derived known answers for the DeltaTree builder.

Still PARITY UNPINNED (the reference cannot be built here): what these cases remove is the common
author of the product's builder and of oracle/builder_oracle.py.  Every expectation below is a LITERAL worked out by
hand from the reference's text ("h:" = /root/reference/deltapq_create_approx_tree.h) -- clique grouping and the tallest
parent (partition_linear_opt_approx_with_constraint h:445-627), the pass order (nchoosek, create_tree.h:75-92: the kept
positions in lexicographic order, i.e. the dropped position runs 7, 6, ... 0), the height cap M*h - 2 and the finalists
(h:570-575, h:1292-1313), sibling order by max_dist2p (h:1396-1426), DFS numbering and the QNode fields
(dfs_node_layout h:1156-1183, h:1428-1459), the DTC stream (h:1765-1826).  Where the reference sorts unstably
(__gnu_parallel::sort h:524, std::sort h:1077 / h:1422) the cases are built so that the order does not matter, or the
stable order is taken and said.

Case G (9 codes, M = 8, codebook c[m][k] = k with Ds = 1, so that the centroid table main:101-118 is (a - b)^2):

    id  code                  joins
    0   1 1 1 1 1 1 1 1       root
    1   1 1 1 1 1 1 1 1       diff 0: clique {0, 1}; heights tie (0 = 0): parent 0 (the first), height[0] = 1
    2   1 1 1 1 1 1 1 5       diff 1, position 7 dropped: clique {0, 2, 3}: parent 0 (the tallest), no tie
    3   1 1 1 1 1 1 1 9
    4   2 1 1 1 1 1 1 1       diff 1, position 1 dropped: clique {4, 5}: 5 is the TALLER one by then -> edge (5, 4)
    5   2 3 1 1 1 1 1 1       diff 2, positions 0 and 1 dropped (the last of the 28 combinations): clique {0, 5}: tie 1 = 1,
                              parent 0, height[0] = 2
    6   7 7 7 7 7 7 7 7       diff 8 (nothing kept): clique {0, 6}, parent 0 (2 > 1)
    7   7 7 7 7 7 7 7 8       diff 1, position 7 dropped: clique {6, 7}: tie, parent 6, height[6] = 1
    8   2 3 4 1 1 1 1 1       diff 1, position 2 dropped: clique {5, 8}: tie, parent 5, height[5] = 1

    edges in the order they are made: (0,1) (0,2) (0,3) (6,7) (5,8) (5,4) (0,5) (0,6); one node is left: the root 0.
    (In the position-7 pass the keys of {0,2,3} (seven bytes 01) sort below those of {6,7} (seven bytes 07): h:493-520.)

    squared distances to the parent (h:186-194 on the (a-b)^2 tables): d(1,0) = 0, d(2,0) = 16, d(3,0) = 64, d(5,0) = 1 + 4,
    d(6,0) = 8 * 36 = 288, d(7,6) = 1, d(7,0) = 7 * 36 + 49 = 301, d(8,5) = 9, d(8,0) = 1 + 4 + 9, d(4,5) = 4, d(4,0) = 1.
    max_dist2p[x] = the largest distance of x or a descendant of x to x's parent: 1: 0, 2: 16, 3: 64, 5: max(5, 14, 1) = 14,
    6: max(288, 301) = 301, 8: 9, 4: 4, 7: 1.  Children of the root by max_dist2p descending: 6, 3, 2, 5, 1; of 5: 8, 4.
    DFS: 0, 6, 7, 3, 2, 5, 8, 4, 1.

Case H (all 256 codes over two values per position: the height cap).  Every pass of diff 1 pairs up equally tall nodes:
position 7 dropped: 128 cliques of two nodes of height 0 (parents reach height 1), position 6: 64 of height 1, ... position 3:
8 cliques whose parents reach height 5; position 2: 4 cliques of two nodes of height 5: max_height = 5 + 1 = 6 >= M*h - 2 =
6, so these four parents are FROZEN as finalists (h:570-575) instead of merging on under positions 1 and 0, and in the end
three of them hang directly under the first (h:1297-1313).  A finalist is the root of a binomial tree B6 (C(6, d) nodes at
depth d); depth histogram of the whole tree: C(6, d) + 3 C(6, d - 1) = 1, 9, 33, 65, 75, 51, 19, 3 -- without the cap it
would be B8: C(8, d), depth 8, which the 3-bit depth field of the stream could not even hold.  Every merge edge changes
one position; the finalists differ from the first in positions {0}, {1}, {0, 1}: 252 + 4 = 256 changed bytes,
n_bytes = 8 + 256 + (3 * 255 + 1) / 2 = 647 (h:1765).  None of this depends on which of two equally tall nodes becomes
the parent.
"""
import math
import os
import struct

import numpy as np
import pytest

G_CODES = [[1, 1, 1, 1, 1, 1, 1, 1],
           [1, 1, 1, 1, 1, 1, 1, 1],
           [1, 1, 1, 1, 1, 1, 1, 5],
           [1, 1, 1, 1, 1, 1, 1, 9],
           [2, 1, 1, 1, 1, 1, 1, 1],
           [2, 3, 1, 1, 1, 1, 1, 1],
           [7, 7, 7, 7, 7, 7, 7, 7],
           [7, 7, 7, 7, 7, 7, 7, 8],
           [2, 3, 4, 1, 1, 1, 1, 1]]
G_EDGES = [(0, 1), (0, 2), (0, 3), (6, 7), (5, 8), (5, 4), (0, 5), (0, 6)]
G_VEC_ID = [0, 6, 7, 3, 2, 5, 8, 4, 1]
G_PARENT_POS = [0xFFFFFFFF, 0, 1, 0, 0, 0, 5, 5, 0]
G_DEPTH = [0, 1, 2, 1, 1, 1, 2, 2, 1]
G_MASK = [0xFF, 0xFF, 0x80, 0x80, 0x80, 0x03, 0x04, 0x02, 0x00]     # [0]: the root carries all M positions (h:1437-1443)
G_CHILD_NUM = [8, 1, 0, 0, 0, 2, 0, 0, 0]                            # descendants (h:1182)
G_MAX_DIST_SQ = [301, 1, 0, 0, 0, 9, 0, 0, 0]                        # by DFS position, before the sqrt of h:1453
G_MAX_DIST2P_SQ = [0, 301, 1, 64, 16, 14, 9, 4, 0]
G_STREAM = bytes([1, 1, 1, 1, 1, 1, 1, 1,                             # root code
                  0x21, 0xFF, 7, 7, 7, 7, 7, 7, 7, 7, 0x80, 8,        # nodes 1 (depth 1: code 6) and 2 (depth 2: code 7)
                  0x11, 0x80, 9, 0x80, 5,                             # nodes 3, 4 (codes 3, 2)
                  0x21, 0x03, 2, 3, 0x04, 4,                          # nodes 5 (code 5), 6 (code 8, depth 2)
                  0x12, 0x02, 1, 0x00])                               # nodes 7 (code 4, depth 2), 8 (code 1: a copy of the root)


def sqrt_f32_bits(x):
    """Bit pattern of the float nearest to sqrt(x), x a non-negative integer, by integer arithmetic alone."""
    if x == 0:
        return 0
    s = 40
    r = math.isqrt(x << (2 * s))                   # floor(sqrt(x) * 2^s)
    e = r.bit_length() - 24                        # keep 24 significant bits
    n, rem = r >> e, r & ((1 << e) - 1)
    if rem >= 1 << (e - 1):                        # sqrt of a non-square is never a tie; squares have rem == 0
        n += 1
    if n == 1 << 24:
        n, e = n >> 1, e + 1
    exp = e - s + 23                               # value = n * 2^(e - s), n in [2^23, 2^24)
    return ((exp + 127) << 23) | (n - (1 << 23))


def g_codebook():
    cb = np.zeros((8, 256, 1), dtype=np.float32)
    cb[:, :, 0] = np.arange(256, dtype=np.float32)[None, :]
    return cb


def test_the_literals_are_consistent():
    """What can be checked without any builder: the stream decodes to the codes in DFS order, the byte count is h:1765's,
    the square roots are the ones the sqrt of h:1453 gives."""
    codes = {0: G_CODES[0]}
    stack = {0: list(G_CODES[0])}
    s, pos, i = G_STREAM, 8, 1
    while i + 1 < 9:
        depths = s[pos]
        pos += 1
        for d in (depths & 7, depths >> 4 & 7):
            code = list(stack[d - 1])
            mask = s[pos]
            pos += 1
            for m in range(8):
                if mask >> m & 1:
                    code[m] = s[pos]
                    pos += 1
            stack[d] = code
            assert d == G_DEPTH[i] and mask == G_MASK[i] and code == G_CODES[G_VEC_ID[i]], i
            i += 1
    n_diffs = sum(bin(m).count("1") for m in G_MASK[1:])
    assert pos == len(G_STREAM) == 8 + n_diffs + (3 * 8 + 1) // 2 == 35
    assert [sqrt_f32_bits(v) for v in (0, 1, 4, 9, 16, 64)] == [struct.unpack("<I", struct.pack("<f", float(v)))[0] for v in (0, 1, 2, 3, 4, 8)]
    for v in (14, 301):
        f = struct.unpack("<f", struct.pack("<I", sqrt_f32_bits(v)))[0]
        up = struct.unpack("<f", struct.pack("<I", sqrt_f32_bits(v) + 1))[0]
        dn = struct.unpack("<f", struct.pack("<I", sqrt_f32_bits(v) - 1))[0]
        # nearest: the exact squares of the midpoints to the neighbours bracket v (integers scaled by 2^60: exact)
        from fractions import Fraction as F
        assert ((F(f) + F(dn)) / 2) ** 2 < v < ((F(f) + F(up)) / 2) ** 2


def check_tree_g(t, edges, what):
    assert [tuple(e) for e in np.asarray(edges).reshape(-1, 2).tolist()] == G_EDGES, what
    assert t["vec_id"].tolist() == G_VEC_ID and t["depth"].tolist() == G_DEPTH, what
    assert t["parent_pos"].tolist() == G_PARENT_POS, what
    assert [int(m) for m in t["mask"][1:]] == G_MASK[1:], what
    assert bytes(t["payload"]) == G_STREAM, what


def test_oracle_builder_on_the_hand_derived_tree():
    from deltapq_amd import synth
    from oracle import builder_oracle
    codes = np.array(G_CODES, dtype=np.uint8)
    ref = builder_oracle.build(codes, g_codebook())
    check_tree_g(dict(vec_id=ref["vec_id"], depth=ref["depths"], parent_pos=ref["parent_pos"], mask=ref["masks"],
                      payload=synth.encode_dtc(ref)[0]), ref["edges"], "oracle/builder_oracle.py")


def qnode_records(path, n):
    rec = np.fromfile(path, dtype=np.uint8)
    assert rec.size == 60 * (n + 1)                                    # h:1484: N + 1 records of sizeof(QNode) = 60
    return rec.reshape(n + 1, 60)


def check_qnode_file_g(path):
    """Every field of the 60-byte QNode records (h:79-101; offsets: vec_id 0, parent_pos 4, child_pos_start 8, child_num 12,
    sub_tree_size 16, qdist 20, max_dist 24, max_dist2p 28, diff_num 32, depth 33, diffs[8] x (m, from, to) 34..57)."""
    rec = qnode_records(path, 9)
    u32 = lambda r, o: struct.unpack("<I", bytes(rec[r, o:o + 4]))[0]
    for pos in range(9):
        vid = G_VEC_ID[pos]
        assert u32(pos, 0) == vid and u32(pos, 4) == G_PARENT_POS[pos], pos
        assert u32(pos, 8) == pos + 1 and u32(pos, 12) == G_CHILD_NUM[pos], pos        # h:1160, h:1182
        assert u32(pos, 16) == 1 and u32(pos, 20) == 0, pos                            # sub_tree_size stays 1 (h:1431), qdist 0
        assert u32(pos, 24) == sqrt_f32_bits(G_MAX_DIST_SQ[pos]), pos                  # h:1453
        assert u32(pos, 28) == sqrt_f32_bits(G_MAX_DIST2P_SQ[pos]), pos                # h:1454
        assert rec[pos, 33] == G_DEPTH[pos], pos
        if pos == 0:
            want = [(m, 255, G_CODES[0][m]) for m in range(8)]                         # h:1437-1443: from = (uchar)-1
        else:
            parent = G_CODES[G_VEC_ID[G_PARENT_POS[pos]]]
            want = [(m, parent[m], G_CODES[vid][m]) for m in range(8) if parent[m] != G_CODES[vid][m]]
        assert rec[pos, 32] == len(want), pos
        assert rec[pos, 34:34 + 3 * len(want)].tolist() == [v for d in want for v in d], pos
    assert u32(9, 16) == 1 and not rec[9, :16].any() and not rec[9, 20:].any()         # the spare record (h:1429-1432)


def test_host_builder_on_the_hand_derived_tree(tmp_path):
    from deltapq_amd import api
    codes = np.array(G_CODES, dtype=np.uint8)
    t = api.DeltaTree(codes, codebook=g_codebook())
    check_tree_g(dict(vec_id=t.vec_id, depth=t.depth, parent_pos=t.parent_pos, mask=t.mask, payload=t.payload()),
                 t.edges, "host builder")
    assert t.stats["n_bytes"] == 35 and t.stats["n_diffs"] == 15 and t.stats["max_depth"] == 2
    t.write_files(str(tmp_path))
    check_qnode_file_g(os.path.join(str(tmp_path), "M8K256_Approx_TreeNodesDFS_N9"))
    edges = np.fromfile(os.path.join(str(tmp_path), "M8K256H1_Approx_Edges_N9"), dtype=np.uint32)
    # root id, then the (parent, child) pairs as they were made (h:1316-1327 writes them before the layout sorts them)
    assert edges[0] == 0 and [tuple(e) for e in edges[1:].reshape(-1, 2).tolist()] == G_EDGES


@pytest.mark.gpu
def test_gpu_builder_on_the_hand_derived_tree(tmp_path):
    from deltapq_amd import api
    if api.device_count() < 1:
        pytest.fail("no GPU")
    codes = np.array(G_CODES, dtype=np.uint8)
    t = api.DeltaTree(codes, codebook=g_codebook(), device=0)
    check_tree_g(dict(vec_id=t.vec_id, depth=t.depth, parent_pos=t.parent_pos, mask=t.mask, payload=t.payload()),
                 t.edges, "GPU builder")
    t.write_files(str(tmp_path))
    check_qnode_file_g(os.path.join(str(tmp_path), "M8K256_Approx_TreeNodesDFS_N9"))


# ---------------------------------------------------------------------------
# H: the height cap
# ---------------------------------------------------------------------------
H_DEPTH_HIST = [1, 9, 33, 65, 75, 51, 19, 3]


def h_codes():
    """Code i (0..255): position p holds 10 + 3 p where bit p of i is clear, 200 - p where it is set."""
    return np.array([[(200 - p) if i >> p & 1 else (10 + 3 * p) for p in range(8)] for i in range(256)], dtype=np.uint8)


def check_tree_h(depth, parent_pos, mask, n_bytes, what):
    assert [math.comb(6, d) + 3 * math.comb(6, d - 1) if d else 1 for d in range(8)] == H_DEPTH_HIST and sum(H_DEPTH_HIST) == 256
    assert np.bincount(np.asarray(depth), minlength=8).tolist() == H_DEPTH_HIST, what
    pc = np.array([bin(int(m)).count("1") for m in mask[1:]])
    assert int(pc.sum()) == 256 and sorted(pc.tolist())[-2:] == [1, 2] and int((pc == 1).sum()) == 254, what
    assert n_bytes == 647, what
    # the three other finalists are children of the root; each carries a B6 like the root's own: 63 descendants
    kids = np.flatnonzero(np.asarray(parent_pos)[1:] == 0) + 1
    assert len(kids) == 9, what
    sizes = sorted(int((kids[j + 1] if j + 1 < len(kids) else 256) - kids[j]) for j in range(len(kids)))
    assert sizes == [1, 2, 4, 8, 16, 32, 64, 64, 64], what              # B0..B5 under the root, then three whole B6


def test_oracle_builder_respects_the_height_cap():
    from deltapq_amd import synth
    from oracle import builder_oracle
    ref = builder_oracle.build(h_codes(), None)
    check_tree_h(ref["depths"], ref["parent_pos"], ref["masks"], len(synth.encode_dtc(ref)[0]), "oracle/builder_oracle.py")


def test_host_builder_respects_the_height_cap():
    from deltapq_amd import api
    t = api.DeltaTree(h_codes())
    check_tree_h(t.depth, t.parent_pos, t.mask, len(t.payload()), "host builder")
    assert t.stats["depth_hist"][:8] == H_DEPTH_HIST and t.stats["max_depth"] == 7


@pytest.mark.gpu
def test_gpu_builder_respects_the_height_cap():
    from deltapq_amd import api
    if api.device_count() < 1:
        pytest.fail("no GPU")
    t = api.DeltaTree(h_codes(), device=0)
    check_tree_h(t.depth, t.parent_pos, t.mask, len(t.payload()), "GPU builder")
