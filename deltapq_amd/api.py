"""Python face of the C-ABI (include/deltapq_amd.h) for tests, bench.py and the
multi-GPU driver.  Everything here is plumbing: numpy/torch buffers in, the HIP
library does the work.  Names of the two convenience functions at the bottom
mirror the reference's entry points (deltapq_create_approx_tree.h:2805, 3731).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import DpqError, Info, OpenOpts, Profile, check  # noqa: F401


def _np_ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def device_count():
    return _lib.load().dpq_device_count()


def read_codewords(path):
    """PQ::ReadCodewords (pq.cpp:288-312) -> float32 [M][K][Ds]."""
    lib = _lib.load()
    M, K, Ds = _lib.c_i32(), _lib.c_i32(), _lib.c_i32()
    check(lib.dpq_read_codewords(path.encode(), M, K, Ds, None), "dpq_read_codewords")
    out = np.empty((M.value, K.value, Ds.value), dtype=np.float32)
    check(lib.dpq_read_codewords(path.encode(), M, K, Ds, _np_ptr(out)), "dpq_read_codewords")
    return out


def read_vecs(path, ext="fvecs", top_n=-1):
    """ReadTopN (utils.cpp:96-110) over .fvecs/.bvecs -> float32 [n][D]."""
    lib = _lib.load()
    n, D = _lib.c_i64(), _lib.c_i32()
    check(lib.dpq_read_vecs(path.encode(), int(ext == "bvecs"), n, D, None, 0), "dpq_read_vecs")
    keep = n.value if top_n < 0 else min(n.value, top_n)
    out = np.empty((keep, D.value), dtype=np.float32)
    check(lib.dpq_read_vecs(path.encode(), int(ext == "bvecs"), n, D, _np_ptr(out), keep), "dpq_read_vecs")
    return out


def read_dtc_file(path):
    """(n_codes, payload u8[n_bytes]) of a reference DTC index file (h:1839-1842)."""
    lib = _lib.load()
    n_codes, n_bytes = _lib.c_i64(), _lib.c_i64()
    check(lib.dpq_read_dtc_header(path.encode(), n_codes, n_bytes), "dpq_read_dtc_header")
    payload = np.fromfile(path, dtype=np.uint8, offset=16, count=n_bytes.value)
    return n_codes.value, payload


def dtc_validate(payload, n_codes, M=8):
    lib = _lib.load()
    pl = np.ascontiguousarray(payload, dtype=np.uint8)
    st = _lib.DtcStats()
    check(lib.dpq_dtc_validate(_np_ptr(pl), pl.size, n_codes, M, st), "dpq_dtc_validate")
    return dict(n_codes=st.n_codes, n_bytes=st.n_bytes, n_diffs=st.n_diffs, max_depth=st.max_depth,
                depth_hist=list(st.depth_hist))


def dtc_encode(root, depths, masks, deltas, M=8):
    """qnodes_to_compressed_codes_opt (h:1765-1826) through the C-ABI."""
    lib = _lib.load()
    root = np.ascontiguousarray(root, dtype=np.uint8)
    depths = np.ascontiguousarray(depths, dtype=np.uint8)
    masks = np.ascontiguousarray(masks, dtype=np.uint16)
    deltas = np.ascontiguousarray(deltas, dtype=np.uint8)
    nb = _lib.c_i64()
    check(lib.dpq_dtc_encode(_np_ptr(root), _np_ptr(depths), _np_ptr(masks), _np_ptr(deltas), len(depths), M, None,
                             nb), "dpq_dtc_encode")
    out = np.empty(nb.value, dtype=np.uint8)
    check(lib.dpq_dtc_encode(_np_ptr(root), _np_ptr(depths), _np_ptr(masks), _np_ptr(deltas), len(depths), M,
                             _np_ptr(out), nb), "dpq_dtc_encode")
    return out


class HostSoA:
    """The transcoded structure-of-arrays image, built on the host (no GPU)."""

    def __init__(self, payload, n_codes, M=8, shard_rank=0, shard_count=1, chunks_per_segment=0, num_codes=0,
                 multi_index_stride=0):
        lib = _lib.load()
        pl = np.ascontiguousarray(payload, dtype=np.uint8)
        opts = OpenOpts(0, shard_rank, shard_count, chunks_per_segment, 0, num_codes, multi_index_stride)
        h = ctypes.c_void_p()
        check(lib.dpq_soa_build(_np_ptr(pl), pl.size, n_codes, M, opts, h), "dpq_soa_build")
        self._h = h
        info = Info()
        check(lib.dpq_soa_info(h, info), "dpq_soa_info")
        self.info = info.as_dict()
        names = ["nib", "mask", "delta", "seg_delta_off", "seg_ckpt", "mi_cell_start", "mi_code", "mi_id", "par", "carry",
                 "st_ckpt", "st_mask", "st_poff", "st_pbase", "st_delta", "st_depth"]
        views = {"seg_delta_off": np.uint64, "st_ckpt": np.uint64, "st_mask": np.uint32, "st_poff": np.uint16, "st_pbase": np.uint32,
                 "st_depth": np.uint16}
        for which, name in enumerate(names):
            ptr, nb = ctypes.c_void_p(), _lib.c_i64()
            check(lib.dpq_soa_array(h, which, ptr, nb), "dpq_soa_array")
            buf = (ctypes.c_ubyte * nb.value).from_address(ptr.value) if nb.value else b""
            arr = np.frombuffer(buf, dtype=np.uint8).copy()
            if name in views:
                arr = arr.view(views[name])
            if name.startswith("mi_"):
                arr = arr.view(np.uint32)
            setattr(self, name, arr)
        lib.dpq_soa_free(h)
        self._h = None


class DeltaTree:
    """DeltaTree built from raw PQ codes on the host (`deltapq -task approx_tree`, method 1;
    create_approx_tree, deltapq_create_approx_tree.h:970-1065)."""

    _ARRAYS = [("vec_id", np.uint32), ("parent_pos", np.uint32), ("depth", np.uint8), ("mask", np.uint16),
               ("deltas", np.uint8), ("root", np.uint8), ("edges", np.uint32)]

    def __init__(self, codes, K=256, max_height_folds=1, codebook=None, device=None):
        """device=None: everything on the host; device=i: the edge search runs on GPU i
        (dpq_tree_build_gpu) and yields the same tree."""
        lib = _lib.load()
        self._lib = lib
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        assert c.ndim == 2
        n, M = c.shape
        cb = None if codebook is None else np.ascontiguousarray(codebook, dtype=np.float32)
        h = ctypes.c_void_p()
        cbp, ds = (None, 0) if cb is None else (_np_ptr(cb), cb.shape[2])
        if device is None:
            check(lib.dpq_tree_build(_np_ptr(c), n, M, K, max_height_folds, cbp, ds, h), "dpq_tree_build")
        else:
            check(lib.dpq_tree_build_gpu(_np_ptr(c), n, M, K, max_height_folds, cbp, ds, device, h),
                  "dpq_tree_build_gpu")
        self._h = h
        self.M, self.K, self.n = M, K, n
        st = _lib.DtcStats()
        check(lib.dpq_tree_stats(h, st), "dpq_tree_stats")
        self.stats = dict(n_codes=st.n_codes, n_bytes=st.n_bytes, n_diffs=st.n_diffs, max_depth=st.max_depth,
                          depth_hist=list(st.depth_hist))
        for which, (name, dt) in enumerate(self._ARRAYS):
            ptr, nb = ctypes.c_void_p(), _lib.c_i64()
            check(lib.dpq_tree_array(h, which, ptr, nb), "dpq_tree_array")
            buf = (ctypes.c_ubyte * nb.value).from_address(ptr.value) if nb.value else b""
            setattr(self, name, np.frombuffer(buf, dtype=np.uint8).copy().view(dt))
        self.edges = self.edges.reshape(-1, 2)

    def payload(self):
        nb = _lib.c_i64()
        check(self._lib.dpq_tree_encode(self._h, None, nb), "dpq_tree_encode")
        out = np.empty(nb.value, dtype=np.uint8)
        check(self._lib.dpq_tree_encode(self._h, _np_ptr(out), nb), "dpq_tree_encode")
        return out

    def write_files(self, dataset_dir):
        check(self._lib.dpq_tree_write_files(self._h, dataset_dir.encode()), "dpq_tree_write_files")

    def close(self):
        if self._h is not None:
            self._lib.dpq_tree_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def encode_pq(vectors, codebook, device=0):
    """PQTree::EncodePlain (pq_tree.cpp:215-237) on the GPU: uint8 codes [n][M]."""
    lib = _lib.load()
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    cb = np.ascontiguousarray(codebook, dtype=np.float32)
    M, K, Ds = cb.shape
    out = np.empty((v.shape[0], M), dtype=np.uint8)
    check(lib.dpq_encode_pq(_np_ptr(v), v.shape[0], v.shape[1], _np_ptr(cb), M, K, Ds, device, _np_ptr(out)),
          "dpq_encode_pq")
    return out


def read_codes_plain(path, M):
    """PQTree::Read (pq_tree.cpp:1032-1081): uint8 [N][M]."""
    lib = _lib.load()
    n = _lib.c_i64()
    check(lib.dpq_read_codes_plain(path.encode(), M, n, None), "dpq_read_codes_plain")
    out = np.empty((n.value, M), dtype=np.uint8)
    check(lib.dpq_read_codes_plain(path.encode(), M, n, _np_ptr(out)), "dpq_read_codes_plain")
    return out


def read_codes_plain_ex(path, M, K=256, with_id=False):
    """PQTree::Read's other layouts (pq_tree.cpp:1050-1078): uint16 codes for K > 256, (code, int32 id) records with_id.
    Returns (codes [N][M] uint8 or uint16, ids int32 [N] or None)."""
    lib = _lib.load()
    n = _lib.c_i64()
    check(lib.dpq_read_codes_plain_ex(path.encode(), M, K, int(with_id), n, None, None), "dpq_read_codes_plain_ex")
    codes = np.empty((n.value, M), dtype=np.uint16 if K > 256 else np.uint8)
    ids = np.empty(n.value, dtype=np.int32) if with_id else None
    check(lib.dpq_read_codes_plain_ex(path.encode(), M, K, int(with_id), n, _np_ptr(codes), None if ids is None else _np_ptr(ids)),
          "dpq_read_codes_plain_ex")
    return codes, ids


def write_codes_plain(path, codes):
    c = np.ascontiguousarray(codes, dtype=np.uint8)
    check(_lib.load().dpq_write_codes_plain(path.encode(), _np_ptr(c), c.shape[0], c.shape[1]), "dpq_write_codes_plain")


def read_qnode_ids(path, n_codes):
    """DFS position -> original vector id (QNode.vec_id) from a TreeNodesDFS file."""
    out = np.empty(n_codes, dtype=np.uint32)
    check(_lib.load().dpq_read_qnode_ids(path.encode(), n_codes, _np_ptr(out)), "dpq_read_qnode_ids")
    return out


def _apply_tuning(opts, tune):
    """dpq_open_opts' plan and tiling knobs by name (stream_max_queries, coarse_below, plan_ratios, boot_cap,
    boot_target, flags, batch_tile_nodes); 0 / absent = the measured default."""
    for name, value in tune.items():
        if name == "plan_ratios":
            vals = list(value) + [0, 0, 0]
            for i in range(3):
                opts.plan_ratios[i] = int(vals[i])
        elif name in ("stream_max_queries", "coarse_below", "boot_cap", "boot_target", "flags", "batch_tile_nodes"):
            v = int(value)
            if name == "stream_max_queries" and v > MAX_STREAM_QUERIES:
                raise ValueError("stream_max_queries above %d: the stream pass answers at most four queries per pass" % MAX_STREAM_QUERIES)
            if name == "boot_cap" and v != 0 and not 2048 <= v <= 16384:
                raise ValueError("boot_cap outside 2048..16384")
            setattr(opts, name, v)
        else:
            raise TypeError("unknown dpq_open_opts field %r" % name)
    return opts


# dpq_open_opts.flags (include/deltapq_amd.h DPQ_OPT_*)
OPT_NO_RELABEL, OPT_NO_FUSE_QUANTISE, OPT_NO_ASYNC_OVERLAP, OPT_BOOT_FULLSORT = 1, 2, 4, 8
OPT_NO_TIGHTEN, OPT_NO_STRANDS, OPT_FORCE_STRANDS, OPT_NO_STRAND1 = 16, 32, 64, 128
MAX_STREAM_QUERIES = 16   # a larger stream_max_queries would send a big batch through ceil(nq / 4) full passes over the index


class DeltaPQIndex:
    """One DTC index (or one shard) resident on one MI355X."""

    def __init__(self, handle):
        self._h = handle
        self._lib = _lib.load()

    @classmethod
    def open_file(cls, path, M=8, K=256, device=0, shard_rank=0, shard_count=1, chunks_per_segment=0,
                  cand_capacity=0, num_codes=0, bootstrap=0, batch_decode=0, **tune):
        """num_codes > 0: scan only the first num_codes codes (the reference's -N below the header's n_codes).
        bootstrap: 0 auto, 1 on, -1 off (dpq_open_opts.bootstrap).  batch_decode: 0 auto, 1 always decode once per batch
        into the plain-code scratch, -1 always decode inside the scan, n >= 2 scratch in tiles of n segments
        (dpq_open_opts.batch_decode)."""
        lib = _lib.load()
        opts = _apply_tuning(OpenOpts(device, shard_rank, shard_count, chunks_per_segment, cand_capacity, num_codes, bootstrap,
                                      batch_decode), tune)
        h = ctypes.c_void_p()
        check(lib.dpq_open_file(path.encode(), M, K, opts, h), "dpq_open_file")
        return cls(h)

    @classmethod
    def open_memory(cls, payload, n_codes, M=8, K=256, device=0, shard_rank=0, shard_count=1, chunks_per_segment=0,
                    cand_capacity=0, num_codes=0, bootstrap=0, global_offset=0, global_n_codes=0, batch_decode=0, **tune):
        """global_offset / global_n_codes: the payload is a self-contained part of a larger index (ids are
        reported as global_offset + local position; dpq_open_opts)."""
        lib = _lib.load()
        pl = np.ascontiguousarray(payload, dtype=np.uint8)
        opts = _apply_tuning(OpenOpts(device, shard_rank, shard_count, chunks_per_segment, cand_capacity, num_codes, bootstrap,
                                      batch_decode, global_offset, global_n_codes), tune)
        h = ctypes.c_void_p()
        check(lib.dpq_open_memory(_np_ptr(pl), pl.size, n_codes, M, K, opts, h), "dpq_open_memory")
        return cls(h)

    @classmethod
    def open_plain(cls, codes, K=256, device=0, shard_rank=0, shard_count=1, chunks_per_segment=0, cand_capacity=0,
                   num_codes=0, bootstrap=0, **tune):
        """Uncompressed comparator index (`-task pqscan`, h:2590-2678): raw codes, fp32-accumulated distances."""
        lib = _lib.load()
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        opts = _apply_tuning(OpenOpts(device, shard_rank, shard_count, chunks_per_segment, cand_capacity, num_codes, bootstrap), tune)
        h = ctypes.c_void_p()
        check(lib.dpq_open_plain_memory(_np_ptr(c), c.shape[0], c.shape[1], K, opts, h), "dpq_open_plain_memory")
        return cls(h)

    def set_codebook(self, codebook):
        cb = np.ascontiguousarray(codebook, dtype=np.float32)
        assert cb.ndim == 3
        check(self._lib.dpq_set_codebook(self._h, _np_ptr(cb), cb.shape[2]), "dpq_set_codebook")
        return self

    def info(self):
        i = Info()
        check(self._lib.dpq_get_info(self._h, i), "dpq_get_info")
        return i.as_dict()

    def query_batch(self, queries, top_k):
        """Host buffers in/out.  Returns (ids int32 [nq][k], dists float32 [nq][k])."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        ids = np.empty((nq, top_k), dtype=np.int32)
        dists = np.empty((nq, top_k), dtype=np.float32)
        check(self._lib.dpq_query_batch(self._h, _np_ptr(q), nq, top_k, _np_ptr(ids), _np_ptr(dists)),
              "dpq_query_batch")
        return ids, dists

    def query_batch_host_async(self, queries, top_k, ids, dists):
        """dpq_query_batch_host_async: host arrays in and out, enqueued only -- up to four batches in flight, queries up and
        results down beside the kernels.  `queries` (float32 [nq][D], C-contiguous), `ids` (int32 [nq][k]) and `dists`
        (float32 [nq][k]) belong to the library until finish(); pin them (pin_host) for the copies to overlap."""
        assert queries.dtype == np.float32 and queries.flags.c_contiguous and ids.dtype == np.int32 and dists.dtype == np.float32
        assert ids.flags.c_contiguous and dists.flags.c_contiguous and ids.shape == dists.shape == (queries.shape[0], top_k)
        check(self._lib.dpq_query_batch_host_async(self._h, _np_ptr(queries), queries.shape[0], top_k, _np_ptr(ids), _np_ptr(dists)),
              "dpq_query_batch_host_async")

    def query_batch_torch(self, queries, top_k, out_ids=None, out_dists=None, wait=True, ordered=False):
        """Device tensors in/out on torch's current stream (no host copies).  wait=False
        (dpq_query_batch_device_async) only enqueues the batch: keep the tensors alive and do
        not read the results before finish().  wait=False, ordered=True (dpq_query_batch_device_ordered):
        enqueued on the current stream itself, later work on that stream sees the result unless
        finish() reports a rerun."""
        import torch
        assert queries.is_cuda and queries.dtype == torch.float32 and queries.is_contiguous()
        nq = queries.shape[0]
        if out_ids is None:
            out_ids = torch.empty((nq, top_k), dtype=torch.int32, device=queries.device)
        if out_dists is None:
            out_dists = torch.empty((nq, top_k), dtype=torch.float32, device=queries.device)
        stream = torch.cuda.current_stream(queries.device).cuda_stream
        fn = self._lib.dpq_query_batch_device if wait else (
            self._lib.dpq_query_batch_device_ordered if ordered else self._lib.dpq_query_batch_device_async)
        check(fn(self._h, ctypes.c_void_p(queries.data_ptr()), nq, top_k, ctypes.c_void_p(out_ids.data_ptr()),
                 ctypes.c_void_p(out_dists.data_ptr()), ctypes.c_void_p(stream)),
              "dpq_query_batch_device" if wait else "dpq_query_batch_device_async")
        return out_ids, out_dists

    def finish(self):
        """Wait for the batches enqueued with wait=False and settle their overflow checks (dpq_finish).
        Returns the number of batches that had to be answered again."""
        n = _lib.c_i32()
        check(self._lib.dpq_finish_count(self._h, n), "dpq_finish_count")
        return n.value

    def profile_enable(self, on=True):
        check(self._lib.dpq_profile_enable(self._h, int(on)), "dpq_profile_enable")

    def profile_reset(self):
        check(self._lib.dpq_profile_reset(self._h), "dpq_profile_reset")

    def profile_read(self):
        p = Profile()
        check(self._lib.dpq_profile_read(self._h, p), "dpq_profile_read")
        return p.as_dict()

    def close(self):
        if self._h is not None:
            self._lib.dpq_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def pin_host(arr):
    """Page-lock a numpy array's memory (hipHostRegister through the library): host-to-host batches then overlap their copies."""
    check(_lib.load().dpq_pin_host(_np_ptr(arr), arr.nbytes), "dpq_pin_host")
    return arr


def unpin_host(arr):
    check(_lib.load().dpq_unpin_host(_np_ptr(arr)), "dpq_unpin_host")


def merge_topk_host(ids, dists):
    """ids/dists [n_lists][nq][k] -> merged [nq][k] by (distance, id)."""
    lib = _lib.load()
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    dists = np.ascontiguousarray(dists, dtype=np.float32)
    n_lists, nq, k = ids.shape
    oi = np.empty((nq, k), dtype=np.int32)
    od = np.empty((nq, k), dtype=np.float32)
    check(lib.dpq_merge_topk_host(_np_ptr(ids), _np_ptr(dists), n_lists, nq, k, _np_ptr(oi), _np_ptr(od)),
          "dpq_merge_topk_host")
    return oi, od


def merge_topk_torch(ids, dists):
    """Device merge after an all-gather: ids/dists [n_lists][nq][k] cuda tensors."""
    import torch
    lib = _lib.load()
    assert ids.is_cuda and ids.is_contiguous() and dists.is_contiguous()
    n_lists, nq, k = ids.shape
    oi = torch.empty((nq, k), dtype=torch.int32, device=ids.device)
    od = torch.empty((nq, k), dtype=torch.float32, device=ids.device)
    stream = torch.cuda.current_stream(ids.device).cuda_stream
    check(lib.dpq_merge_topk_device(ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(dists.data_ptr()), n_lists, nq,
                                    k, ctypes.c_void_p(oi.data_ptr()), ctypes.c_void_p(od.data_ptr()),
                                    ids.device.index or 0, ctypes.c_void_p(stream)), "dpq_merge_topk_device")
    return oi, od


def merge_topk_packed_torch(gathered, k):
    """Device merge straight from the all-gathered tensor [n_lists][nq][2k] int32 (ids | distance bits)."""
    import torch
    lib = _lib.load()
    assert gathered.is_cuda and gathered.is_contiguous() and gathered.dtype == torch.int32 and gathered.shape[2] == 2 * k
    n_lists, nq = gathered.shape[0], gathered.shape[1]
    oi = torch.empty((nq, k), dtype=torch.int32, device=gathered.device)
    od = torch.empty((nq, k), dtype=torch.float32, device=gathered.device)
    stream = torch.cuda.current_stream(gathered.device).cuda_stream
    check(lib.dpq_merge_topk_device_packed(ctypes.c_void_p(gathered.data_ptr()), n_lists, nq, k, ctypes.c_void_p(oi.data_ptr()),
                                           ctypes.c_void_p(od.data_ptr()), gathered.device.index or 0, ctypes.c_void_p(stream)),
          "dpq_merge_topk_device_packed")
    return oi, od


# ---------------------------------------------------------------------------
# Reference-named conveniences (one call per query, like main:328-339).
# ---------------------------------------------------------------------------

def query_processing_scan_compressed_codes_opt_in_memory(codes, n_bytes, query, top_k, M, K, m_Ds, num_codes,
                                                         m_codewords, device=0):
    """Signature of deltapq_create_approx_tree.h:3731-3736 (decoder argument dropped).
    Returns results as a list of (id, dist) ascending, like `results` there."""
    payload = np.asarray(codes, dtype=np.uint8)[:n_bytes]
    cb = np.asarray(m_codewords, dtype=np.float32).reshape(M, K, m_Ds)
    with DeltaPQIndex.open_memory(payload, num_codes, M, K, device=device) as idx:
        idx.set_codebook(cb)
        ids, dists = idx.query_batch(np.asarray(query, dtype=np.float32)[None, :], top_k)
    return list(zip(ids[0].tolist(), dists[0].tolist()))


def query_processing_scan_compressed_codes_opt_o_direct(dataset_path, query, top_k, M, K, m_Ds, num_codes,
                                                        m_codewords, device=0):
    """Signature of deltapq_create_approx_tree.h:2805-2810 (decoder argument dropped)."""
    lib = _lib.load()
    buf = ctypes.create_string_buffer(4096)
    check(lib.dpq_dtc_file_name(dataset_path.encode(), M, K, num_codes, buf, 4096), "dpq_dtc_file_name")
    cb = np.asarray(m_codewords, dtype=np.float32).reshape(M, K, m_Ds)
    with DeltaPQIndex.open_file(buf.value.decode(), M, K, device=device) as idx:
        idx.set_codebook(cb)
        ids, dists = idx.query_batch(np.asarray(query, dtype=np.float32)[None, :], top_k)
    return list(zip(ids[0].tolist(), dists[0].tolist()))
