"""How much of the scan time is LDS bank conflicts?  Pure-scan time (nothing survives) for trees whose
codes are all identical (every gather is a broadcast) ... fully random (worst case)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deltapq_amd import synth, api, _lib
n, nq = 1000000, 1024
cb = synth.make_codebook(8, 256, 16, 100); qs = synth.make_queries(nq, 128, 101)
lib = _lib.load()
lib.dpq_debug_scan_time.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
for md in (0.0, 0.5, 1.5, 3.0, 8.0):
    tree = synth.synth_tree(n, 8, seed=5, mean_diffs=md)
    payload, nb = synth.encode_dtc(tree)
    with api.DeltaPQIndex.open_memory(payload, n, 8, 256) as idx:
        idx.set_codebook(cb); idx.query_batch(qs, 100)
        ms = ctypes.c_float(); rc = lib.dpq_debug_scan_time(idx._h, nq, 0, 10, 0, ms); assert rc == 0
        print("mean_diffs=%.1f: pure scan %.3f ms (%.2f B/code)" % (md, ms.value, nb / n), flush=True)
