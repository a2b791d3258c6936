"""Developer probe: fixed cost of one scan launch (prologue + epilogue) vs segments scanned."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deltapq_amd import synth, api, _lib
n, nq = 1000000, 1024
cb = synth.make_codebook(8, 256, 16, 100)
qs = synth.make_queries(nq, 128, 101)
tree = synth.synth_tree(n, 8, seed=102)
payload, nb = synth.encode_dtc(tree)
lib = _lib.load()
lib.dpq_debug_scan_time.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
with api.DeltaPQIndex.open_memory(payload, n, 8, 256) as idx:
    idx.set_codebook(cb)
    for topk in (100, 400, 1600):
        idx.query_batch(qs, topk)   # leaves the final thresholds of a top-`topk` search behind
        for nseg in (16, 106, 856, 3907):
            os.environ["DPQ_DEBUG_NSEG"] = str(nseg)
            for mode in (0, 2):
                ms = ctypes.c_float()
                rc = lib.dpq_debug_scan_time(idx._h, nq, mode, 20, 0, ms)
                assert rc == 0, lib.dpq_last_error()
                print("thresholds of top-%d, nseg=%d mode=%d: %.1f us" % (topk, nseg, mode, ms.value * 1e3), flush=True)
