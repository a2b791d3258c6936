"""oracle/dtc_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Two things live here:

1. ``Oracle``: a ctypes wrapper over oracle/_build/liboracle.so (dtc_oracle.cpp),
   the C++ restatement of the reference's ``deltapq -task query`` path.
2. ``py_*``: a second, independent restatement in pure Python + numpy scalars
   (stack machine, fp64 incremental distances, and libstdc++'s
   push_heap/pop_heap written out by hand) used only to cross-check (1) on
   small cases.

PARITY UNPINNED: the reference has no tests/fixtures for this path and cannot
be built here (OpenCV missing: pq.h:8), see the header of dtc_oracle.cpp and
DESIGN.md.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module.

Citations: h: = /root/reference/deltapq_create_approx_tree.h,
main: = /root/reference/deltapq_approx_tree_main.cpp.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    """Compile liboracle.so with g++ (oracle/Makefile)."""
    src = os.path.join(_HERE, "dtc_oracle.cpp")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB


_c_f = ctypes.POINTER(ctypes.c_float)
_c_i = ctypes.POINTER(ctypes.c_int)
_c_u8 = ctypes.POINTER(ctypes.c_ubyte)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class Oracle:
    """ctypes face of dtc_oracle.cpp."""

    def __init__(self):
        self.lib = ctypes.CDLL(build())
        L = self.lib
        L.oracle_build_lut.argtypes = [_c_f, _c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_f]
        L.oracle_build_lut.restype = None
        L.oracle_query_in_memory.argtypes = [_c_u8, ctypes.c_longlong, _c_f, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_longlong, _c_f, _c_i, _c_f]
        L.oracle_scan_lut.argtypes = [_c_u8, ctypes.c_longlong, _c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_longlong, _c_i, _c_f, _c_f, _c_u8]
        L.oracle_query_o_direct.argtypes = [ctypes.c_char_p, _c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_longlong, _c_f, _c_i, _c_f]
        L.oracle_pqscan_plain.argtypes = [_c_u8, ctypes.c_longlong, _c_f, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, _c_i, _c_f]
        L.oracle_query_many.argtypes = [_c_u8, ctypes.c_longlong, _c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_longlong, _c_f, ctypes.c_int, _c_i, _c_f]
        L.oracle_o_direct_supported.argtypes = [ctypes.c_char_p]
        L.oracle_read_codewords.argtypes = [ctypes.c_char_p, _c_i, _c_i, _c_i, _c_f]
        L.oracle_read_vecs.argtypes = [ctypes.c_char_p, ctypes.c_int, _c_i, _c_f, ctypes.c_longlong]
        L.oracle_read_vecs.restype = ctypes.c_longlong

    # a3 (h:2841-2849)
    def build_lut(self, codebook, query):
        cb = np.ascontiguousarray(codebook, dtype=np.float32)
        M, K, Ds = cb.shape
        q = np.ascontiguousarray(query, dtype=np.float32)
        assert q.size == M * Ds
        lut = np.empty((M, K), dtype=np.float32)
        self.lib.oracle_build_lut(_p(cb, _c_f), _p(q, _c_f), M, K, Ds, _p(lut, _c_f))
        return lut

    # a2 twin (h:3731-3892)
    def query_in_memory(self, payload, n_codes, codebook, query, top_k):
        cb = np.ascontiguousarray(codebook, dtype=np.float32)
        M, K, Ds = cb.shape
        q = np.ascontiguousarray(query, dtype=np.float32)
        pl = np.ascontiguousarray(payload, dtype=np.uint8)
        ids = np.empty(top_k, dtype=np.int32)
        dists = np.empty(top_k, dtype=np.float32)
        rc = self.lib.oracle_query_in_memory(_p(pl, _c_u8), pl.size, _p(q, _c_f), top_k, M, K, Ds, n_codes,
                                             _p(cb, _c_f), _p(ids, _c_i), _p(dists, _c_f))
        if rc != 0:
            raise ValueError("oracle_query_in_memory rc=%d" % rc)
        return ids, dists

    def query_many(self, payload, n_codes, codebook, queries, top_k, n_threads):
        """The in-memory query for a batch of queries on n_threads C++ threads (CPU baseline of bench.py)."""
        cb = np.ascontiguousarray(codebook, dtype=np.float32)
        M, K, Ds = cb.shape
        q = np.ascontiguousarray(queries, dtype=np.float32)
        pl = np.ascontiguousarray(payload, dtype=np.uint8)
        ids = np.empty((len(q), top_k), dtype=np.int32)
        dists = np.empty((len(q), top_k), dtype=np.float32)
        rc = self.lib.oracle_query_many(_p(pl, _c_u8), pl.size, _p(q, _c_f), len(q), top_k, M, K, Ds, n_codes,
                                        _p(cb, _c_f), n_threads, _p(ids, _c_i), _p(dists, _c_f))
        if rc != 0:
            raise ValueError("oracle_query_many rc=%d" % rc)
        return ids, dists

    def o_direct_supported(self, path):
        return bool(self.lib.oracle_o_direct_supported(path.encode()))

    def scan_lut(self, payload, n_codes, lut, top_k, want_all=False):
        lut = np.ascontiguousarray(lut, dtype=np.float32)
        M, K = lut.shape
        pl = np.ascontiguousarray(payload, dtype=np.uint8)
        ids = np.empty(top_k, dtype=np.int32)
        dists = np.empty(top_k, dtype=np.float32)
        all_d = np.empty(n_codes, dtype=np.float32) if want_all else None
        all_c = np.empty((n_codes, M), dtype=np.uint8) if want_all else None
        rc = self.lib.oracle_scan_lut(_p(pl, _c_u8), pl.size, _p(lut, _c_f), top_k, M, K, n_codes,
                                      _p(ids, _c_i), _p(dists, _c_f), _p(all_d, _c_f), _p(all_c, _c_u8))
        if rc != 0:
            raise ValueError("oracle_scan_lut rc=%d" % rc)
        if want_all:
            return ids, dists, all_d, all_c
        return ids, dists

    # a2 (h:2805-2984), 4 KB block reads
    def query_o_direct(self, path, n_codes, codebook, query, top_k):
        cb = np.ascontiguousarray(codebook, dtype=np.float32)
        M, K, Ds = cb.shape
        q = np.ascontiguousarray(query, dtype=np.float32)
        ids = np.empty(top_k, dtype=np.int32)
        dists = np.empty(top_k, dtype=np.float32)
        rc = self.lib.oracle_query_o_direct(path.encode(), _p(q, _c_f), top_k, M, K, Ds, n_codes, _p(cb, _c_f),
                                            _p(ids, _c_i), _p(dists, _c_f))
        if rc != 0:
            raise ValueError("oracle_query_o_direct rc=%d" % rc)
        return ids, dists

    # comparator #9 (h:2590-2678), fp32 accumulation
    def pqscan_plain(self, codes, lut, top_k):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        lut = np.ascontiguousarray(lut, dtype=np.float32)
        n, M = codes.shape
        ids = np.empty(top_k, dtype=np.int32)
        dists = np.empty(top_k, dtype=np.float32)
        rc = self.lib.oracle_pqscan_plain(_p(codes, _c_u8), n, _p(lut, _c_f), top_k, M, lut.shape[1],
                                          _p(ids, _c_i), _p(dists, _c_f))
        if rc != 0:
            raise ValueError("oracle_pqscan_plain rc=%d" % rc)
        return ids, dists

    def read_codewords(self, path):
        M, K, Ds = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        rc = self.lib.oracle_read_codewords(path.encode(), M, K, Ds, None)
        if rc != 0:
            raise IOError("oracle_read_codewords rc=%d" % rc)
        out = np.empty((M.value, K.value, Ds.value), dtype=np.float32)
        rc = self.lib.oracle_read_codewords(path.encode(), M, K, Ds, _p(out, _c_f))
        if rc != 0:
            raise IOError("oracle_read_codewords rc=%d" % rc)
        return out

    def read_vecs(self, path, ext):
        D = ctypes.c_int()
        n = self.lib.oracle_read_vecs(path.encode(), int(ext == "bvecs"), D, None, 0)
        if n < 0:
            raise IOError("oracle_read_vecs rc=%d" % n)
        out = np.empty((n, D.value), dtype=np.float32)
        self.lib.oracle_read_vecs(path.encode(), int(ext == "bvecs"), D, _p(out, _c_f), n)
        return out


# ---------------------------------------------------------------------------
# Independent pure-Python restatement (small cases only).
# ---------------------------------------------------------------------------

def py_build_lut(codebook, query):
    """h:2841-2849 with numpy scalar types standing in for C float/double."""
    cb = np.asarray(codebook, dtype=np.float32)
    M, K, Ds = cb.shape
    q = np.asarray(query, dtype=np.float32)
    lut = np.zeros((M, K), dtype=np.float32)
    for i in range(M):
        for j in range(K):
            acc = np.float32(0.0)
            for k in range(Ds):
                diff = np.float32(cb[i, j, k] - q[i * Ds + k])        # fp32 subtract
                sq = np.float64(diff) * np.float64(diff)               # pow(.,2) in double
                acc = np.float32(np.float64(acc) + sq)                 # float += double
            lut[i, j] = acc
    return lut


class _LibstdcxxMaxHeap:
    """std::priority_queue<pair<float,uint>, vector, cmp_max> (h:2851-2853) with
    libstdc++'s __push_heap / __adjust_heap written out, so the order in which
    equal-distance entries leave the heap is the reference's."""

    def __init__(self):
        self.c = []

    @staticmethod
    def _comp(a, b):            # cmp_max, h:2054-2056
        return a[0] < b[0]

    def _push_heap(self, hole, top, value):
        c = self.c
        parent = (hole - 1) // 2
        while hole > top and self._comp(c[parent], value):
            c[hole] = c[parent]
            hole = parent
            parent = (hole - 1) // 2
        c[hole] = value

    def _adjust_heap(self, hole, length, value):
        c = self.c
        top = hole
        second = hole
        while second < (length - 1) // 2:
            second = 2 * (second + 1)
            if self._comp(c[second], c[second - 1]):
                second -= 1
            c[hole] = c[second]
            hole = second
        if (length & 1) == 0 and second == (length - 2) // 2:
            second = 2 * (second + 1)
            c[hole] = c[second - 1]
            hole = second - 1
        self._push_heap(hole, top, value)

    def push(self, value):
        self.c.append(value)
        self._push_heap(len(self.c) - 1, 0, value)

    def pop(self):
        c = self.c
        if len(c) > 1:
            value = c[-1]
            c[-1] = c[0]
            self._adjust_heap(0, len(c) - 1, value)
        c.pop()

    def top(self):
        return self.c[0]

    def __len__(self):
        return len(self.c)


def py_scan(payload, n_codes, lut, top_k, M=8):
    """h:3760-3890 as a Python stack machine.  Returns (ids, dists, all_dists,
    all_codes) with the even-N id quirk."""
    buf = bytes(np.asarray(payload, dtype=np.uint8).tobytes())
    lut = np.asarray(lut, dtype=np.float32)
    T = [[float(np.float64(lut[m, k])) for k in range(lut.shape[1])] for m in range(M)]  # exact f32->f64
    off = 0
    heap = _LibstdcxxMaxHeap()
    stack = [[0] * M for _ in range(M)]
    dstack = [0.0] * M
    all_d = np.zeros(n_codes, dtype=np.float32)
    all_c = np.zeros((n_codes, M), dtype=np.uint8)

    q = 0.0
    for m in range(M):
        cid = buf[off]; off += 1
        q += T[m][cid]                     # python float == C double
        stack[0][m] = cid
    dstack[0] = q
    heap.push((float(np.float32(q)), 0))
    all_d[0] = np.float32(q)
    all_c[0] = stack[0]

    def process(depth, report_id, pos, store):
        nonlocal off
        stack[depth] = list(stack[depth - 1])
        dist = dstack[depth - 1]
        bitmap = buf[off]; off += 1
        for m in range(8):
            if bitmap & (1 << m):
                cid = buf[off]; off += 1
                stack[depth][m] = cid
                frm = stack[depth - 1][m]
                dist -= T[m][frm]
                dist += T[m][cid]
        if store:
            dstack[depth] = dist
        f = float(np.float32(dist))
        if len(heap) < top_k:
            heap.push((f, report_id))
        elif dist < heap.top()[0]:         # double < float
            heap.pop()
            heap.push((f, report_id))
        all_d[pos] = np.float32(dist)
        all_c[pos] = stack[depth]

    i = 1
    while i + 1 < n_codes:
        depths = buf[off]; off += 1
        process(depths & 7, i, i, True)
        process((depths >> 4) & 7, i + 1, i + 1, True)
        i += 2
    if i == n_codes - 1:
        depth = buf[off]; off += 1
        process(depth, i + 1, i, False)
    ids = np.zeros(top_k, dtype=np.int32)
    dists = np.zeros(top_k, dtype=np.float32)
    for r in range(top_k - 1, -1, -1):
        ids[r] = heap.top()[1]
        dists[r] = heap.top()[0]
        heap.pop()
    assert off <= len(buf)
    return ids, dists, all_d, all_c, off


# ---------------------------------------------------------------------------
# Comparators shared by the parity tests.
# ---------------------------------------------------------------------------

def tie_aware_equal(ids_a, dists_a, ids_b, dists_b, all_dists=None, n_codes=None):
    """SURVEY.md section 7 'tie semantics': distances must be bit-equal and
    ascending; ids must agree as sets inside every equal-distance group except
    the last (boundary) group, where any node whose distance equals the boundary
    value is acceptable (checked against all_dists when given).

    Returns (ok, message)."""
    da = np.asarray(dists_a, dtype=np.float32)
    db = np.asarray(dists_b, dtype=np.float32)
    ia = np.asarray(ids_a).astype(np.int64)
    ib = np.asarray(ids_b).astype(np.int64)
    if da.shape != db.shape:
        return False, "shape mismatch"
    if not np.array_equal(da.view(np.uint32), db.view(np.uint32)):
        bad = int(np.flatnonzero(da.view(np.uint32) != db.view(np.uint32))[0])
        return False, "distance bits differ at rank %d: %r vs %r" % (bad, da[bad], db[bad])
    if np.any(np.diff(da) < 0):
        return False, "distances not ascending"
    k = len(da)
    if k == 0:
        return True, ""
    boundary = da[-1]
    start = 0
    while start < k:
        end = start
        while end < k and da[end] == da[start]:
            end += 1
        ga, gb = set(ia[start:end].tolist()), set(ib[start:end].tolist())
        if len(ga) != end - start or len(gb) != end - start:
            return False, "duplicate ids inside a tie group at rank %d" % start
        if da[start] != boundary:
            if ga != gb:
                return False, "id sets differ in tie group at rank %d: %s vs %s" % (start, sorted(ga), sorted(gb))
        elif all_dists is not None:
            ad = np.asarray(all_dists, dtype=np.float32)
            n = len(ad) if n_codes is None else n_codes
            for g in (ga, gb):
                for i in g:
                    pos = i
                    if n % 2 == 0 and i == n:          # even-N quirk (h:2949, 2970)
                        pos = n - 1
                    if pos < 0 or pos >= n or ad[pos] != boundary:
                        return False, "boundary id %d does not have the boundary distance" % i
        start = end
    return True, ""
