import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deltapq_amd import synth, api, _lib
n, nq = 1000000, 1024
cb = synth.make_codebook(8, 256, 16, 100); qs = synth.make_queries(nq, 128, 101)
tree = synth.synth_tree(n, 8, seed=102); payload, nb = synth.encode_dtc(tree)
lib = _lib.load()
lib.dpq_debug_select_time.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
with api.DeltaPQIndex.open_memory(payload, n, 8, 256) as idx:
    idx.set_codebook(cb); idx.query_batch(qs, 100)
    for q in (256, 512, 768, 1000, 1024):
        ms = ctypes.c_float(); rc = lib.dpq_debug_select_time(idx._h, q, 100, 0, 5, ms); assert rc == 0
        print('level-0 select, %4d queries: %.1f us' % (q, ms.value * 1e3), flush=True)
