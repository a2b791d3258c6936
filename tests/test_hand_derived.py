"""Hand-derived known answers for the oracle (and, GPU-marked, for the product).

The reference ships no vectors and cannot be built here (PARITY UNPINNED), and the two
restatements under oracle/ share an author.  These cases remove that common mode as far as it can be removed
without the reference: every stream below is WRITTEN OUT BYTE BY BYTE from the format definition
(qnodes_to_compressed_codes_opt, /root/reference/deltapq_create_approx_tree.h:1771-1826), every expected answer is
a LITERAL worked out by hand from the scan's definition (h:2841-2975), and the arithmetic is re-derived in the
test with exact rationals (fractions.Fraction + an explicit round-to-nearest-even to 24 / 53 bits) -- no numpy or
C floating point takes part in producing an expectation.

Cases: even-N trailing node with its whole-byte depth and id N; a depth-7 chain; a stream
that crosses a 4 KB block; a k-boundary where `double < float` admission decides the id; table inputs on which
fp32-only, fp64-only and the reference's mixed rule give three different bit patterns.
"""
import os
import struct
from fractions import Fraction as F

import numpy as np
import pytest


# ---------------------------------------------------------------------------
# exact arithmetic
# ---------------------------------------------------------------------------

def rnd(x, p):
    """The rational x rounded to the nearest number with a p-bit significand, ties to even (IEEE round-to-nearest;
    no exponent limits: the values here are far from them)."""
    x = F(x)
    if x == 0:
        return F(0)
    s = 1 if x > 0 else -1
    x = abs(x)
    e = x.numerator.bit_length() - x.denominator.bit_length()
    if F(2) ** e > x:
        e -= 1
    if F(2) ** (e + 1) <= x:
        e += 1                      # 2^e <= x < 2^(e+1)
    scale = F(2) ** (e - p + 1)
    q = x / scale
    n = q.numerator // q.denominator
    r = q - n
    if r > F(1, 2) or (r == F(1, 2) and n % 2 == 1):
        n += 1
    return s * n * scale


def bits32(x):
    """IEEE-754 single bit pattern of a rational that IS representable (asserted)."""
    x = F(x)
    assert rnd(x, 24) == x
    return struct.unpack("<I", struct.pack("<f", x.numerator / x.denominator))[0]


def table_entry_by_the_book(c, q):
    """h:2845-2846 `m_sub_distances[i][j] += pow(m_codewords[i][j][k] - query[i*m_Ds+k], 2)` with a float
    accumulator: float - float (one rounding to 24 bits), pow(double, 2) (the square of a 24-bit value has 48 bits:
    exact in double), float += double = (double)acc + sq rounded to 53 bits, then stored as float (24 bits)."""
    acc = F(0)
    for ci, qi in zip(c, q):
        d = rnd(F(ci) - F(qi), 24)
        sq = d * d
        assert rnd(sq, 53) == sq
        acc = rnd(rnd(acc + sq, 53), 24)
    return acc


def scan_by_the_book(stream, n_codes, T, k, M=8):
    """The scan as its definition reads (h:2866-2982), on exact rationals: root = M raw bytes; then for node pairs one
    byte `depth_i | depth_{i+1} << 4` (3-bit fields), per node a mask byte and the changed bytes in ascending
    position; a node left over (even N) owns a WHOLE depth byte and is reported with id N.  code = parent's code
    (the latest node of depth - 1) with the changed positions replaced; dist = dist(parent) - T[m][from] + T[m][to]
    per changed position, each operation rounded to double.  Returns [(id, float distance)] of every node in
    stream order -- the top-k rule is applied by the caller."""
    pos = 0
    code_at, dist_at = {}, {}
    root = list(stream[:M])
    pos = M
    d = F(0)
    for m in range(M):
        d = rnd(d + T[m][root[m]], 53)
    code_at[0], dist_at[0] = root, d
    out = [(0, d)]

    def node(depth, report):
        nonlocal pos
        parent, dist = list(code_at[depth - 1]), dist_at[depth - 1]
        mask = stream[pos]
        pos += 1
        code = list(parent)
        for m in range(M):
            if mask >> m & 1:
                to = stream[pos]
                pos += 1
                dist = rnd(dist - T[m][parent[m]], 53)
                dist = rnd(dist + T[m][to], 53)
                code[m] = to
        code_at[depth], dist_at[depth] = code, dist
        out.append((report, dist))

    i = 1
    while i + 1 < n_codes:
        b = stream[pos]
        pos += 1
        node(b & 7, i)
        node(b >> 4 & 7, i + 1)
        i += 2
    if i == n_codes - 1:
        depth = stream[pos]        # the whole byte
        pos += 1
        node(depth, i + 1)         # h:2970: emplace(dist, i+1)
    assert pos == len(stream)
    return out


def heap_topk_by_the_book(nodes, k):
    """h:2909-2914 for SMALL k, written as a plain list: keep k entries; a node enters while the list is short, or when
    its DOUBLE distance is strictly below the largest stored FLOAT distance, replacing that entry.  Which of several
    equal largest entries is replaced is not specified here (only used where the largest is unique)."""
    kept = []
    for ident, d in nodes:
        f = rnd(d, 24)
        if len(kept) < k:
            kept.append((f, ident))
            continue
        top = max(kept)
        assert sum(1 for e in kept if e[0] == top[0]) == 1 or not d < top[0], "ambiguous replacement"
        if d < top[0]:
            kept.remove(max(kept, key=lambda e: e[0]))
            kept.append((f, ident))
    return sorted(kept)


def int_table(M=8):
    """T[m][k] = (m + 1) * k: integers, every sum exact in every precision."""
    return [[F((m + 1) * k) for k in range(256)] for m in range(M)]


def np_table(T):
    return np.array([[float(v) for v in row] for row in T], dtype=np.float32)


# ---------------------------------------------------------------------------
# the streams, byte by byte
# ---------------------------------------------------------------------------

# A. N = 4 (even): root [1,0,0,0,0,0,0,0]; node 1 depth 1 changes position 1 -> 3; node 2 depth 2 (child of node 1)
#    changes position 0 -> 0; node 3 is left over: a WHOLE depth byte (1), changes position 2 -> 1, reported as id 4.
STREAM_A = bytes([1, 0, 0, 0, 0, 0, 0, 0,      # root code                                  (h:1771-1773)
                  0x21,                         # depth(1) = 1 | depth(2) = 2 << 4             (h:1781-1788)
                  0x02, 3,                      # node 1: mask bit 1, byte 3                    (h:1791-1800)
                  0x01, 0,                      # node 2: mask bit 0, byte 0                    (h:1802-1811)
                  0x01,                         # node 3: whole byte = depth 1                  (h:1813-1818)
                  0x04, 1])                     # node 3: mask bit 2, byte 1                    (h:1819-1826)
# distances with T[m][k] = (m+1) k: node 0: 1; node 1: 1 + 2*3 = 7; node 2: code [0,3,..] = 6; node 3: [1,0,1,..] = 1 + 3 = 4
ANSWER_A = {4: ([0, 4, 2, 1], [1.0, 4.0, 6.0, 7.0]), 2: ([0, 4], [1.0, 4.0]), 1: ([0], [1.0])}

# B. N = 9 (odd): a chain root -> 1 -> 2 -> ... -> 7 (depths 1..7, the 3-bit maximum), node i sets position i-1 to i;
#    node 8 returns to depth 1 and sets position 7 to 2.
STREAM_B = bytes([0] * 8 +
                 [0x21, 0x01, 1, 0x02, 2,       # nodes 1, 2
                  0x43, 0x04, 3, 0x08, 4,       # nodes 3, 4
                  0x65, 0x10, 5, 0x20, 6,       # nodes 5, 6
                  0x17, 0x40, 7, 0x80, 2])      # node 7 (depth 7) and node 8 (depth 1)
# node i (1..7): sum_{j<i} (j+1)^2 = 1, 5, 14, 30, 55, 91, 140; node 8: 8 * 2 = 16; root: 0
ANSWER_B = ([0, 1, 2, 3, 8], [0.0, 1.0, 5.0, 14.0, 16.0])          # top-5


def stream_c():
    """C. N = 2001: node i (1..2000) is a depth-1 child of the all-zero root that sets position 0 to (37 i) mod 256.
    5 bytes per node pair: 8 + 5000 payload bytes, so the fd path (h:2783-2804) refills its 4 KB buffer once."""
    s = [0] * 8
    for i in range(1, 2001, 2):
        s += [0x11, 0x01, (37 * i) % 256, 0x01, (37 * (i + 1)) % 256]
    return bytes(s)


# 37 * 173 = 6401 = 25 * 256 + 1: distance 0 at i = 0 (root) and the multiples of 256; distance 1 at i = 173 + 256 j
ANSWER_C_ZERO = [0, 256, 512, 768, 1024, 1280, 1536, 1792]
ANSWER_C_ONE = [173, 429, 685, 941, 1197, 1453, 1709, 1965]

# D. k = 1, N = 3.  T[0][0] = T[1][0] = 1/2, T[1][1] = 1/2 - 2^-25 (a float), T[1][2] = 1/2, everything else 0.
#    root (all zeros): 1.  node 1 sets position 1 to 1: 1 - 1/2 + (1/2 - 2^-25) = 1 - 2^-25, exact in double, and
#    float(1 - 2^-25) = 1.0 (a tie between 1 - 2^-24 and 1, to even).  h:2911 compares the DOUBLE 1 - 2^-25 with the
#    stored FLOAT 1.0: smaller, so node 1 replaces the root although both print as 1.0.  node 2 (position 1 -> 2) is
#    exactly 1.0: not smaller, stays out.  A float-vs-float comparison would have answered id 0.
STREAM_D = bytes([0] * 8 + [0x11, 0x02, 1, 0x02, 2])
ANSWER_D = ([1], [0x3F800000])

# E. table arithmetic: one entry, Ds = 3.  Exact rationals of the six floats; the three rules give three patterns.
E_C = [F(9024741, 262144), F(11757, 256), F(16120429, 65536)]
E_Q = [F(2254117, 131072), F(11750887, 262144), F(10049977, 131072)]
E_MIXED, E_FP32_ONLY, E_FP64_ONLY = 0x46E2431A, 0x46E24319, 0x46E2431B


# D2. Case D scaled by 1/2 so that a real codebook can produce its table (Ds = 4, all-zero query; the product takes a
#     codebook, not a table).  T[0][0] = T[1][0] = T[1][2] = 1/4 = (1/2)^2; T[1][1] = 1/4 - 2^-26 =
#     (3^2 + 9^2 + 90^2 + 4095^2) / 2^26 (every partial sum is an integer below 2^24 over 2^26: exact in float, so the
#     mixed rule of h:2845-2846 accumulates it without a rounding).  root = 1/2; node 1 = 1/2 - 1/4 + (1/4 - 2^-26) =
#     1/2 - 2^-26 exactly in double, and float(1/2 - 2^-26) = 1/2 (the midpoint of 1/2 - 2^-25 and 1/2, to even);
#     node 2 = 1/2.  h:2911: the DOUBLE 1/2 - 2^-26 is below the stored FLOAT 1/2 -> the reference answers id 1.
D2_ROWS = {(0, 0): [0, 0, 0, F(1, 2)], (1, 0): [0, 0, 0, F(1, 2)], (1, 2): [0, 0, 0, F(1, 2)],
           (1, 1): [F(3, 8192), F(9, 8192), F(90, 8192), F(4095, 8192)]}
ANSWER_D2 = ([1], [0x3F000000])

# F. The documented deviation, pinned to a known case: the product forms the fp64 SUM of the M
#    table entries, the reference carries an INCREMENTAL fp64 stack (h:2896-2905); they differ where a partial sum is not
#    exact in double.  Ds = 3, all-zero query.  T[0][1] = 1; T[1][1] = 2^-24 + 2^-47 = (2^-24)^2 + (2^-24)^2 + (2^-12)^2
#    (float accumulation: 2^-48, 2^-47, 2^-24 (1 + 2^-23): exact); T[3][1] = 2^30 = (2^15)^2; T[3][0] = 0.
#    root code [1,1,0,1,0,0,0,0]: 1 + (2^-24 + 2^-47) + 0 + 2^30 -> double(2^30 + 1 + 2^-24 + 2^-47) = 2^30 + 1 (the
#    fraction is a quarter of the double's ulp 2^-22 and a bit: down).  node 1 (depth 1) sets position 3 to 0:
#    reference: (2^30 + 1) - 2^30 + 0 = 1 -> float 1.0 = 0x3F800000.
#    sum rule: 1 + 2^-24 + 2^-47 (exact in double) -> float: above the midpoint 1 + 2^-24 -> 1 + 2^-23 = 0x3F800001.
#    One ulp apart: 1.2e-7 relative, inside the north star's 1e-5.
STREAM_F = bytes([1, 1, 0, 1, 0, 0, 0, 0,     # root
                  0x11,                        # depths of nodes 1, 2
                  0x08, 0,                     # node 1: position 3 -> 0
                  0x00])                       # node 2: a copy of the root (keeps N odd: no trailing node)
F_ROWS = {(0, 1): [0, 0, 1], (1, 1): [F(1, 2 ** 24), F(1, 2 ** 24), F(1, 2 ** 12)], (3, 1): [0, 0, 2 ** 15]}
F_REFERENCE_BITS, F_SUM_RULE_BITS = 0x3F800000, 0x3F800001


def rows_codebook(rows, Ds):
    """A codebook [8][256][Ds] that is zero except for the given (m, k) rows (exact floats, asserted)."""
    cb = np.zeros((8, 256, Ds), dtype=np.float32)
    for (m, k), row in rows.items():
        for d, v in enumerate(row):
            cb[m, k, d] = float(F(v))
            assert F(float(cb[m, k, d])) == F(v)
    return cb


def table_from_rows(rows, Ds):
    """The table the mixed rule builds from such a codebook against the all-zero query, on exact rationals."""
    T = [[F(0)] * 256 for _ in range(8)]
    for (m, k), row in rows.items():
        T[m][k] = table_entry_by_the_book([F(v) for v in row], [F(0)] * Ds)
    return T


def table_d():
    T = [[F(0)] * 256 for _ in range(8)]
    T[0][0] = F(1, 2)
    T[1][0] = F(1, 2)
    T[1][1] = F(1, 2) - F(1, 2 ** 25)
    T[1][2] = F(1, 2)
    return T


# ---------------------------------------------------------------------------
# the literals agree with the exact re-derivation (no oracle involved)
# ---------------------------------------------------------------------------

def test_literals_follow_from_the_definitions():
    T = int_table()
    a = scan_by_the_book(STREAM_A, 4, T, 4)
    assert a == [(0, 1), (1, 7), (2, 6), (4, 4)]                       # the left-over node is reported as 4 = N
    for k, (ids, ds) in ANSWER_A.items():
        assert [(F(d), i) for i, d in zip(ids, ds)] == heap_topk_by_the_book(a, k)
    b = scan_by_the_book(STREAM_B, 9, T, 5)
    assert [d for _, d in b] == [0, 1, 5, 14, 30, 55, 91, 140, 16]
    assert [(F(d), i) for i, d in zip(*ANSWER_B)] == heap_topk_by_the_book(b, 5)
    c = scan_by_the_book(stream_c(), 2001, T, 16)
    assert len(stream_c()) == 8 + 5000
    assert sorted(i for i, d in c if d == 0) == ANSWER_C_ZERO and sorted(i for i, d in c if d == 1) == ANSWER_C_ONE
    d = scan_by_the_book(STREAM_D, 3, table_d(), 1)
    assert d == [(0, 1), (1, 1 - F(1, 2 ** 25)), (2, 1)] and rnd(d[1][1], 24) == 1
    assert [(F(1), 1)] == heap_topk_by_the_book(d, 1)
    # a float-vs-float admission would have kept the root
    assert not rnd(d[1][1], 24) < rnd(d[0][1], 24)
    # E: three rules, three patterns
    assert all(rnd(v, 24) == v for v in E_C + E_Q)
    assert bits32(table_entry_by_the_book(E_C, E_Q)) == E_MIXED
    acc32 = F(0)
    acc64 = F(0)
    for ci, qi in zip(E_C, E_Q):
        d32 = rnd(ci - qi, 24)
        acc32 = rnd(acc32 + rnd(d32 * d32, 24), 24)
        d64 = rnd(ci - qi, 53)
        acc64 = rnd(acc64 + rnd(d64 * d64, 53), 53)
    assert bits32(acc32) == E_FP32_ONLY and bits32(rnd(acc64, 24)) == E_FP64_ONLY
    assert len({E_MIXED, E_FP32_ONLY, E_FP64_ONLY}) == 3
    # D2: the codebook rows give the table by the book, and the scan gives the admission case
    T2 = table_from_rows(D2_ROWS, 4)
    assert T2[0][0] == T2[1][0] == T2[1][2] == F(1, 4) and T2[1][1] == F(1, 4) - F(1, 2 ** 26)
    d2 = scan_by_the_book(STREAM_D, 3, T2, 1)
    assert d2 == [(0, F(1, 2)), (1, F(1, 2) - F(1, 2 ** 26)), (2, F(1, 2))] and rnd(d2[1][1], 24) == F(1, 2)
    assert heap_topk_by_the_book(d2, 1) == [(F(1, 2), 1)] and bits32(F(1, 2)) == ANSWER_D2[1][0]
    # F: incremental stack against one-shot sum
    TF = table_from_rows(F_ROWS, 3)
    assert TF[0][1] == 1 and TF[1][1] == F(1, 2 ** 24) + F(1, 2 ** 47) and TF[3][1] == 2 ** 30 and rnd(TF[1][1], 24) == TF[1][1]
    f = scan_by_the_book(STREAM_F, 3, TF, 3)
    assert f[0][1] == 2 ** 30 + 1 and f[1] == (1, F(1)) and bits32(rnd(f[1][1], 24)) == F_REFERENCE_BITS
    one_shot = F(0)
    for m, byte in enumerate([1, 1, 0, 0, 0, 0, 0, 0]):      # node 1's code, summed in position order, double by double
        one_shot = rnd(one_shot + TF[m][byte], 53)
    assert one_shot == 1 + F(1, 2 ** 24) + F(1, 2 ** 47) and bits32(rnd(one_shot, 24)) == F_SUM_RULE_BITS
    assert abs(rnd(one_shot, 24) - rnd(f[1][1], 24)) / rnd(f[1][1], 24) == F(1, 2 ** 23) < F(1, 10 ** 5)


# ---------------------------------------------------------------------------
# the oracle (both restatements) against the literals
# ---------------------------------------------------------------------------

def both_scans(oracle, stream, n, lut, k):
    from oracle import dtc_oracle as O
    pl = np.frombuffer(stream, dtype=np.uint8)
    ids, d = oracle.scan_lut(pl, n, lut, k)
    pids, pd = O.py_scan(pl, n, lut, k)[:2]
    assert np.array_equal(ids, np.asarray(pids)) and np.array_equal(d.view(np.uint32), np.asarray(pd, dtype=np.float32).view(np.uint32))
    return ids, d


def test_oracle_even_n_trailing_node(oracle):
    lut = np_table(int_table())
    for k, (ids, ds) in ANSWER_A.items():
        got_i, got_d = both_scans(oracle, STREAM_A, 4, lut, k)
        assert got_i.tolist() == ids and got_d.tolist() == ds


def test_oracle_depth_seven_chain(oracle):
    got_i, got_d = both_scans(oracle, STREAM_B, 9, np_table(int_table()), 5)
    assert got_i.tolist() == ANSWER_B[0] and got_d.tolist() == ANSWER_B[1]
    _, _, alld, codes = oracle.scan_lut(np.frombuffer(STREAM_B, dtype=np.uint8), 9, np_table(int_table()), 1, want_all=True)
    assert alld.tolist() == [0, 1, 5, 14, 30, 55, 91, 140, 16]
    assert codes[7].tolist() == [1, 2, 3, 4, 5, 6, 7, 0] and codes[8].tolist() == [0, 0, 0, 0, 0, 0, 0, 2]


def test_oracle_stream_across_a_4k_block(oracle, tmp_path):
    lut = np_table(int_table())
    s = stream_c()
    got_i, got_d = both_scans(oracle, s, 2001, lut, 16)
    assert got_d.tolist() == [0.0] * 8 + [1.0] * 8
    assert sorted(got_i[:8].tolist()) == ANSWER_C_ZERO and sorted(got_i[8:].tolist()) == ANSWER_C_ONE
    # the fd path (4 KB reads, h:2783-2804) through a file with the reference's header (h:1839-1841)
    path = str(tmp_path / "M8K256_Approx_compressed_codes_opt_N2001")
    with open(path, "wb") as f:
        f.write(struct.pack("<qq", 2001, len(s)) + s)
    cb = np.zeros((8, 256, 1), dtype=np.float32)       # Ds = 1: T[m][k] = (c - 0)^2 with c = sqrt((m+1) k) only where exact
    cb[0, :, 0] = np.sqrt(np.arange(256, dtype=np.float64)).astype(np.float32)
    exact = [k for k in range(256) if int(round(k ** 0.5)) ** 2 == k]   # perfect squares: the table entry is exactly k
    ids_fd, d_fd = oracle.query_o_direct(path, 2001, cb, np.zeros(8, dtype=np.float32), 8)
    assert 0 in exact and d_fd.tolist() == [0.0] * 8 and sorted(ids_fd.tolist()) == ANSWER_C_ZERO


# G / H. The ORDER in which equal distances leave the reference's heap (h:2851-2853 priority_queue with cmp_max, admission
#    h:2909-2914, drain h:2977-2982), worked through libstdc++'s __push_heap / __adjust_heap BY HAND.  cmp_max compares the
#    distance only, so ties are decided by where the sift operations stop, not by ids.  T[m][k] = (m + 1) k; every node is a
#    depth-1 child of the root that sets position 0, so a node's distance is its byte.
#
# G. N = 7, k = 3, distances by id: 5 3 5 7 5 1 9 -- three equal keys AT the cut.
#    push (5,0):           c = [(5,0)]
#    push (3,1): hole 1, parent 0: 5 < 3 ? no                                   c = [(5,0) (3,1)]
#    push (5,2): hole 2, parent 0: 5 < 5 ? no                                   c = [(5,0) (3,1) (5,2)]
#    (7,3): 7 < top 5 ? no.   (5,4): 5 < 5 ? no (strict: the first seen stay).
#    (1,5): 1 < 5: pop -- value = c[2] = (5,2), c[2] = c[0]; __adjust_heap(hole 0, len 2, value): the loop does not run
#           (0 < (2-1)/2 = 0 fails); len even and second == (2-2)/2: second = 2, c[0] = c[1] = (3,1), hole 1;
#           __push_heap(hole 1, top 0, (5,2)): parent 0: 3 < 5 ? yes: c[1] = (3,1), hole 0; stop: c[0] = (5,2)
#           -> the root (5,0) is what left; c = [(5,2) (3,1)]; push (1,5): hole 2, parent 0: 5 < 1 ? no  c = [(5,2) (3,1) (1,5)]
#    (9,6): no.   drain: results[2] = (5,2); then (3,1); then (1,5).
#    The reference answers ids [5, 1, 2] -- id 2, NOT the first-seen id 0, carries the boundary distance.
STREAM_G = bytes([5, 0, 0, 0, 0, 0, 0, 0,
                  0x11, 0x01, 3, 0x01, 5,        # nodes 1, 2: depth 1 each, position 0 <- 3, 5
                  0x11, 0x01, 7, 0x01, 5,        # nodes 3, 4
                  0x11, 0x01, 1, 0x01, 9])       # nodes 5, 6
ANSWER_G = ([5, 1, 2], [1.0, 3.0, 5.0])
# H. N = 5, k = 3, distances by id: 4 2 2 9 8 -- two equal keys INSIDE the list.
#    push (4,0); push (2,1): 4 < 2 ? no; push (2,2): 4 < 2 ? no                  c = [(4,0) (2,1) (2,2)]
#    (9,3), (8,4): not below the top 4.
#    drain: results[2] = (4,0); pop: value = (2,2), __adjust_heap(0, 2, value): len even, second == 0: c[0] = c[1] = (2,1),
#           hole 1; __push_heap(1, 0, (2,2)): parent 0: 2 < 2 ? no: c[1] = (2,2)   c = [(2,1) (2,2)]
#           results[1] = (2,1); results[0] = (2,2).
#    The reference emits the equal pair as ids 2, 1 -- not in ascending id order.
STREAM_H = bytes([4, 0, 0, 0, 0, 0, 0, 0,
                  0x11, 0x01, 2, 0x01, 2,
                  0x11, 0x01, 9, 0x01, 8])
ANSWER_H = ([2, 1, 0], [2.0, 2.0, 4.0])


def test_oracle_heap_order_of_equal_distances(oracle):
    """Both oracle restatements (the C++ one on the real std::priority_queue, the Python one with libstdc++'s sift
    routines written out) return the hand-derived reference order; the tie-aware comparator accepts exactly the
    canonical alternatives (ascending id) this build returns and nothing with a wrong distance or a foreign id."""
    from oracle import dtc_oracle as O
    lut = np_table(int_table())
    for stream, n, (ids, ds), canon in ((STREAM_G, 7, ANSWER_G, [5, 1, 0]), (STREAM_H, 5, ANSWER_H, [1, 2, 0])):
        got_i, got_d = both_scans(oracle, stream, n, lut, 3)
        assert got_i.tolist() == ids and got_d.tolist() == ds
        book = scan_by_the_book(stream, n, int_table(), 3)
        assert [float(d) for _, d in book] == [float(b) for b in stream[:1]] + [float(stream[10 + 5 * (j // 2) + 2 * (j % 2)]) for j in range(n - 1)]
        pl = np.frombuffer(stream, dtype=np.uint8)
        _, _, alld, _ = oracle.scan_lut(pl, n, lut, 3, want_all=True)
        ok, msg = O.tie_aware_equal(np.array(canon), np.array(ds, dtype=np.float32), got_i, got_d, alld, n)
        assert ok, msg
        wrong = list(canon)
        wrong[-1] = 3                                    # id 3 (distance 7 / 9) does not belong to the boundary group
        ok, _ = O.tie_aware_equal(np.array(wrong), np.array(ds, dtype=np.float32), got_i, got_d, alld, n)
        assert not ok


def test_oracle_double_against_float_admission(oracle):
    got_i, got_d = both_scans(oracle, STREAM_D, 3, np_table(table_d()), 1)
    assert got_i.tolist() == ANSWER_D[0] and got_d.view(np.uint32).tolist() == ANSWER_D[1]


def test_oracle_admission_case_from_a_real_codebook(oracle):
    """D2: the same admission case with its table built by the oracle's own LUT rule from a Ds = 4 codebook."""
    cb = rows_codebook(D2_ROWS, 4)
    lut = oracle.build_lut(cb, np.zeros(32, dtype=np.float32))
    assert lut[1, 1] == np.float32(0.25 - 2.0 ** -26) and lut[0, 0] == lut[1, 0] == lut[1, 2] == np.float32(0.25)
    got_i, got_d = both_scans(oracle, STREAM_D, 3, lut, 1)
    assert got_i.tolist() == ANSWER_D2[0] and got_d.view(np.uint32).tolist() == ANSWER_D2[1]


def test_oracle_incremental_stack_differs_from_the_sum_rule_on_case_f(oracle):
    """F: the oracle follows the reference's incremental stack (0x3F800000), not the fp64 sum (0x3F800001)."""
    cb = rows_codebook(F_ROWS, 3)
    lut = oracle.build_lut(cb, np.zeros(24, dtype=np.float32))
    assert lut[3, 1] == np.float32(2.0 ** 30) and float(lut[1, 1]) == 2.0 ** -24 + 2.0 ** -47
    _, _, alld, codes = oracle.scan_lut(np.frombuffer(STREAM_F, dtype=np.uint8), 3, lut, 1, want_all=True)
    assert codes[1].tolist() == [1, 1, 0, 0, 0, 0, 0, 0]
    assert int(alld.view(np.uint32)[1]) == F_REFERENCE_BITS
    one_shot = np.float32(sum(np.float64(lut[m, codes[1][m]]) for m in range(8)))
    assert int(one_shot.view(np.uint32)) == F_SUM_RULE_BITS


def test_oracle_table_arithmetic_is_the_mixed_rule(oracle):
    from oracle import dtc_oracle as O
    cb = np.array([[[float(v) for v in E_C]]], dtype=np.float32)        # M = 1, K = 1, Ds = 3
    q = np.array([float(v) for v in E_Q], dtype=np.float32)
    assert [F(float(v)) for v in cb[0, 0]] == E_C and [F(float(v)) for v in q] == E_Q
    assert int(oracle.build_lut(cb, q).view(np.uint32)[0, 0]) == E_MIXED
    assert int(np.asarray(O.py_build_lut(cb, q), dtype=np.float32).view(np.uint32)[0, 0]) == E_MIXED


# ---------------------------------------------------------------------------
# the product (HIP path through the C-ABI) against the same literals
# ---------------------------------------------------------------------------

def codebook_for_table(T, Ds=1):
    """A Ds = 1 codebook whose table against the all-zero query is T where T's entries are squares of floats."""
    cb = np.zeros((8, 256, Ds), dtype=np.float32)
    for m in range(8):
        for k in range(256):
            r = F(T[m][k])
            root = F(int(r.numerator ** 0.5), int(r.denominator ** 0.5)) if r else F(0)
            assert root * root == r, "entry is not the square of a rational"
            cb[m, k, 0] = float(root)
            assert F(float(cb[m, k, 0])) == root
    return cb


@pytest.mark.gpu
def test_product_on_hand_derived_streams(lib):
    from deltapq_amd import api
    if api.device_count() < 1:
        pytest.fail("no GPU visible")
    # tables whose entries are perfect squares so that a Ds = 1 codebook reproduces them exactly: T[m][k] = ((m+1) k)^2
    T = [[F(((m + 1) * k) ** 2) for k in range(256)] for m in range(8)]
    cb = codebook_for_table(T)
    zero = np.zeros((1, 8), dtype=np.float32)

    def run(stream, n, k):
        with api.DeltaPQIndex.open_memory(np.frombuffer(stream, dtype=np.uint8), n, 8, 256) as idx:
            idx.set_codebook(cb)
            ids, d = idx.query_batch(zero, k)
        return ids[0].tolist(), d[0].tolist()

    # A with squared tables: node 0: 1; node 1: 1 + 36 = 37; node 2: [0,3,..] = 36; node 3 (id 4): 1 + 9 = 10
    assert run(STREAM_A, 4, 4) == ([0, 4, 2, 1], [1.0, 10.0, 36.0, 37.0])
    # B: node i: sum_{j<i} ((j+1)(j+1))^2 = 1, 17, 98, 354, ...; node 8: (8*2)^2 = 256
    assert run(STREAM_B, 9, 5) == ([0, 1, 2, 3, 8], [0.0, 1.0, 17.0, 98.0, 256.0])
    ids, d = run(stream_c(), 2001, 16)
    assert d == [0.0] * 8 + [1.0] * 8 and ids[:8] == ANSWER_C_ZERO and ids[8:] == ANSWER_C_ONE   # canonical tie order: ascending id
    # G, H (squared tables: distance = byte^2): where the reference's heap order picks ids [5,1,2] / [2,1,0], this build
    # returns the canonical choice, ascending (distance, id), with the same distances
    assert run(STREAM_G, 7, 3) == ([5, 1, 0], [1.0, 9.0, 25.0])
    assert run(STREAM_H, 5, 3) == ([1, 2, 0], [4.0, 4.0, 16.0])


@pytest.mark.gpu
def test_product_table_arithmetic_is_the_mixed_rule(lib):
    """One node (the all-zero root), Ds = 3: its distance is T[0][0] + 0 + ... = the entry built from E_C, E_Q."""
    from deltapq_amd import api
    cb = np.zeros((8, 256, 3), dtype=np.float32)
    q = np.zeros((1, 24), dtype=np.float32)
    cb[0, 0] = [float(v) for v in E_C]
    q[0, :3] = [float(v) for v in E_Q]
    for m in range(1, 8):                       # other sub-spaces: codeword 0 equals the query's sub-vector -> entry 0
        cb[m, 0] = q[0, 3 * m:3 * m + 3] = [m, 2 * m, 3 * m]
    with api.DeltaPQIndex.open_memory(np.zeros(8, dtype=np.uint8), 1, 8, 256) as idx:
        idx.set_codebook(cb)
        ids, d = idx.query_batch(q, 1)
    assert ids[0, 0] == 0 and int(d.view(np.uint32)[0, 0]) == E_MIXED


@pytest.mark.gpu
def test_product_on_the_admission_case(lib):
    """D2 through the HIP path.  All three nodes have the fp32 distance 1/2; the reference's `double < float` admission
    (h:2911) answers id 1 at k = 1.  The product's distance rule gives the same bits for every node and its documented tie
    rule (the k-th boundary group is represented by ascending id) answers id 0 -- the tie-aware
    comparator accepts exactly that: distance bits equal at every rank, boundary ids out of the boundary group."""
    from deltapq_amd import api
    from oracle.dtc_oracle import tie_aware_equal
    cb = rows_codebook(D2_ROWS, 4)
    pl = np.frombuffer(STREAM_D, dtype=np.uint8)
    q = np.zeros((1, 32), dtype=np.float32)
    with api.DeltaPQIndex.open_memory(pl, 3, 8, 256) as idx:
        idx.set_codebook(cb)
        ids1, d1 = idx.query_batch(q, 1)
        ids3, d3 = idx.query_batch(q, 3)
    assert d1.view(np.uint32).tolist() == [ANSWER_D2[1]] and ids1[0, 0] in (0, 1, 2)
    assert d3.view(np.uint32).tolist() == [[0x3F000000] * 3] and sorted(ids3[0].tolist()) == [0, 1, 2]
    alld = np.array([0.5, 0.5, 0.5], dtype=np.float32)
    ok, msg = tie_aware_equal(ids1[0], d1[0], np.array(ANSWER_D2[0]), np.array(ANSWER_D2[1], dtype=np.uint32).view(np.float32), alld, 3)
    assert ok, msg


@pytest.mark.gpu
def test_product_deviation_from_the_incremental_stack_is_the_pinned_case(lib):
    """F through the HIP path: the product's fp64-sum rule answers 0x3F800001 for node 1 where the reference's incremental
    stack answers 0x3F800000 -- one ulp, 1.2e-7 relative, inside the north star's 1e-5 (documented deviation)."""
    from deltapq_amd import api
    cb = rows_codebook(F_ROWS, 3)
    pl = np.frombuffer(STREAM_F, dtype=np.uint8)
    with api.DeltaPQIndex.open_memory(pl, 3, 8, 256) as idx:
        idx.set_codebook(cb)
        ids, d = idx.query_batch(np.zeros((1, 24), dtype=np.float32), 3)
    assert ids[0, 0] == 1 and int(d.view(np.uint32)[0, 0]) == F_SUM_RULE_BITS
    ref = np.array([F_REFERENCE_BITS], dtype=np.uint32).view(np.float32)[0]
    assert abs(float(d[0, 0]) - float(ref)) / float(ref) <= 1e-5
    # nodes 0 and 2 (the root's code): double(2^30 + 1 + 2^-24 + 2^-47) = 2^30 + 1 -> float 2^30 (its ulp is 128)
    assert ids[0, 1:].tolist() == [0, 2] and float(d[0, 1]) == float(d[0, 2]) == 2.0 ** 30
