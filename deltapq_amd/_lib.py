"""ctypes binding of include/deltapq_amd.h (deltapq_amd/csrc/libdeltapq_amd.so).

The library is the product: if it is missing this module raises ImportError --
there is no Python or CPU implementation of the query path to fall back to.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DPQ_LIB_PATH") or os.path.join(_HERE, "csrc", "libdeltapq_amd.so")   # DPQ_LIB_PATH: developer A/B builds

c_i32, c_i64, c_f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
P = ctypes.POINTER


class OpenOpts(ctypes.Structure):
    _fields_ = [("device", c_i32), ("shard_rank", c_i32), ("shard_count", c_i32), ("chunks_per_segment", c_i32),
                ("cand_capacity", c_i32), ("num_codes", c_i32), ("bootstrap", c_i32), ("batch_decode", c_i32),
                ("global_offset", c_i64), ("global_n_codes", c_i64),
                # plan and tiling knobs (0 = measured default), see include/deltapq_amd.h
                ("stream_max_queries", c_i32), ("coarse_below", c_i32), ("plan_ratios", c_i32 * 3), ("boot_cap", c_i32),
                ("boot_target", c_i32), ("flags", c_i32), ("batch_tile_nodes", c_i64)]


class Info(ctypes.Structure):
    _fields_ = [("n_codes_total", c_i64), ("n_bytes_total", c_i64), ("node_lo", c_i64), ("node_hi", c_i64),
                ("algorithmic_bytes", c_i64), ("device_bytes", c_i64), ("n_diffs", c_i64), ("M", c_i32),
                ("K", c_i32), ("Ds", c_i32), ("n_segments", c_i32), ("chunks_per_segment", c_i32),
                ("max_depth", c_i32), ("device", c_i32), ("cand_capacity", c_i32), ("bootstrap_bytes", c_i64),
                ("bootstrap_stride", c_i32), ("batch_decode_mb", c_i32), ("strand_bytes", c_i64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Profile(ctypes.Structure):
    _fields_ = [("lut_ms", ctypes.c_double), ("scan_ms", ctypes.c_double), ("select_ms", ctypes.c_double),
                ("lut_launches", c_i64), ("scan_launches", c_i64), ("select_launches", c_i64),
                ("scan_node_query_pairs", c_i64), ("scan_stream_bytes", c_i64), ("query_batches", c_i64),
                ("queries", c_i64), ("overflow_reruns", c_i64), ("exact_checks", c_i64), ("candidates", c_i64),
                ("quantise_ms", ctypes.c_double), ("decode_ms", ctypes.c_double), ("bootstrap_ms", ctypes.c_double),
                ("bootstrap_launches", c_i64), ("stream_launches", c_i64), ("strand_launches", c_i64),
                ("strand1_launches", c_i64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class DtcStats(ctypes.Structure):
    _fields_ = [("n_codes", c_i64), ("n_bytes", c_i64), ("n_diffs", c_i64), ("depth_hist", c_i64 * 16),
                ("max_depth", c_i32), ("M", c_i32)]


# every symbol include/deltapq_amd.h declares: (name, restype, argtypes)
_VP = ctypes.c_void_p
SYMBOLS = [
    ("dpq_version", ctypes.c_int, []),
    ("dpq_strerror", ctypes.c_char_p, [ctypes.c_int]),
    ("dpq_last_error", ctypes.c_char_p, []),
    ("dpq_device_count", ctypes.c_int, []),
    ("dpq_read_dtc_header", ctypes.c_int, [ctypes.c_char_p, P(c_i64), P(c_i64)]),
    ("dpq_read_codewords", ctypes.c_int, [ctypes.c_char_p, P(c_i32), P(c_i32), P(c_i32), _VP]),
    ("dpq_read_vecs", ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, P(c_i64), P(c_i32), _VP, c_i64]),
    ("dpq_dtc_file_name", ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, c_i64, ctypes.c_char_p, c_i64]),
    ("dpq_dtc_validate", ctypes.c_int, [_VP, c_i64, c_i64, ctypes.c_int, P(DtcStats)]),
    ("dpq_soa_build", ctypes.c_int, [_VP, c_i64, c_i64, ctypes.c_int, P(OpenOpts), P(_VP)]),
    ("dpq_soa_info", ctypes.c_int, [_VP, P(Info)]),
    ("dpq_soa_array", ctypes.c_int, [_VP, ctypes.c_int, P(_VP), P(c_i64)]),
    ("dpq_soa_free", None, [_VP]),
    ("dpq_dtc_encode", ctypes.c_int, [_VP, _VP, _VP, _VP, c_i64, ctypes.c_int, _VP, P(c_i64)]),
    ("dpq_tree_build", ctypes.c_int, [_VP, c_i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _VP, ctypes.c_int, P(_VP)]),
    ("dpq_tree_build_gpu", ctypes.c_int,
     [_VP, c_i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _VP, ctypes.c_int, ctypes.c_int, P(_VP)]),
    ("dpq_tree_stats", ctypes.c_int, [_VP, P(DtcStats)]),
    ("dpq_tree_array", ctypes.c_int, [_VP, ctypes.c_int, P(_VP), P(c_i64)]),
    ("dpq_tree_encode", ctypes.c_int, [_VP, _VP, P(c_i64)]),
    ("dpq_tree_write_files", ctypes.c_int, [_VP, ctypes.c_char_p]),
    ("dpq_tree_free", None, [_VP]),
    ("dpq_read_qnode_ids", ctypes.c_int, [ctypes.c_char_p, c_i64, _VP]),
    ("dpq_read_codes_plain", ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, P(c_i64), _VP]),
    ("dpq_write_codes_plain", ctypes.c_int, [ctypes.c_char_p, _VP, c_i64, ctypes.c_int]),
    ("dpq_read_codes_plain_ex", ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, P(c_i64), _VP, _VP]),
    ("dpq_encode_pq", ctypes.c_int,
     [_VP, c_i64, ctypes.c_int, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _VP]),
    ("dpq_open_plain_memory", ctypes.c_int, [_VP, c_i64, ctypes.c_int, ctypes.c_int, P(OpenOpts), P(_VP)]),
    ("dpq_open_plain_file", ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, P(OpenOpts), P(_VP)]),
    ("dpq_open_file", ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, P(OpenOpts), P(_VP)]),
    ("dpq_open_memory", ctypes.c_int, [_VP, c_i64, c_i64, ctypes.c_int, ctypes.c_int, P(OpenOpts), P(_VP)]),
    ("dpq_set_codebook", ctypes.c_int, [_VP, _VP, ctypes.c_int]),
    ("dpq_get_info", ctypes.c_int, [_VP, P(Info)]),
    ("dpq_close", ctypes.c_int, [_VP]),
    ("dpq_query_batch", ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, _VP, _VP]),
    ("dpq_query_batch_device", ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, _VP, _VP, _VP]),
    ("dpq_query_batch_device_async", ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, _VP, _VP, _VP]),
    ("dpq_query_batch_device_ordered", ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, _VP, _VP, _VP]),
    ("dpq_finish_count", ctypes.c_int, [_VP, P(c_i32)]),
    ("dpq_finish", ctypes.c_int, [_VP]),
    ("dpq_merge_topk_host", ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, _VP, _VP]),
    ("dpq_merge_topk_device", ctypes.c_int,
     [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, _VP, _VP, ctypes.c_int, _VP]),
    ("dpq_merge_topk_device_packed", ctypes.c_int, [_VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, _VP, _VP, ctypes.c_int, _VP]),
    ("dpq_query_batch_host_async", ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, _VP, _VP]),
    ("dpq_pin_host", ctypes.c_int, [_VP, c_i64]),
    ("dpq_unpin_host", ctypes.c_int, [_VP]),
    ("dpq_profile_enable", ctypes.c_int, [_VP, ctypes.c_int]),
    ("dpq_profile_reset", ctypes.c_int, [_VP]),
    ("dpq_profile_read", ctypes.c_int, [_VP, P(Profile)]),
    # developer diagnostics (DPQ_DEV=1 only; scripts/)
    ("dpq_debug_scan_time", ctypes.c_int, [_VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P(ctypes.c_float)]),
    ("dpq_debug_scan_stamps", ctypes.c_int, [_VP, ctypes.c_int, ctypes.c_int, P(ctypes.c_ulonglong), ctypes.c_int, P(ctypes.c_float)]),
    ("dpq_debug_boot_stamps", ctypes.c_int, [_VP, ctypes.c_int, P(ctypes.c_double)]),
    ("dpq_debug_select_time", ctypes.c_int, [_VP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P(ctypes.c_float)]),
    ("dpq_debug_strand1_stamps", ctypes.c_int, [_VP, _VP, ctypes.c_int]),
]

_lib = None
_hip_runtime = None


def _bind_hip_runtime():
    """Load THE HIP runtime of this process with RTLD_GLOBAL before the C-ABI
    library (which is built with -no-hip-rt and resolves hip* against it).
    A process must not mix two runtimes: PyTorch-ROCm wheels bundle their own
    libamdhip64.so, so when torch is installed we bind to that one (whether or
    not torch has been imported yet); otherwise to /opt/rocm's."""
    global _hip_runtime
    if _hip_runtime is not None:
        return _hip_runtime
    import importlib.util
    candidates = []
    override = os.environ.get("DPQ_HIP_RUNTIME")
    if override:
        candidates.append(override)
    try:
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.submodule_search_locations:
            candidates.append(os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so"))
    except (ImportError, ValueError):
        pass
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    candidates += [os.path.join(rocm, "lib", "libamdhip64.so"), "libamdhip64.so"]
    last = None
    for path in candidates:
        if os.path.isabs(path) and not os.path.exists(path):
            continue
        try:
            _hip_runtime = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            return _hip_runtime
        except OSError as e:
            last = e
    raise ImportError("cannot load a HIP runtime (libamdhip64.so): %s" % last)


def load():
    """Load the C-ABI library, binding every declared symbol (raises if one is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `make -C deltapq_amd/csrc` (or __graft_entry__.build()); "
            "the DeltaPQ query path has no CPU fallback" % LIB_PATH)
    _bind_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


class DpqError(RuntimeError):
    def __init__(self, status, where):
        lib = load()
        self.status = status
        detail = lib.dpq_last_error().decode(errors="replace")
        super().__init__("%s failed: %s (%d)%s" % (where, lib.dpq_strerror(status).decode(), status,
                                                  ": " + detail if detail else ""))


def check(status, where):
    if status != 0:
        raise DpqError(status, where)
