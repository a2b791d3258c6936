"""Developer probe: scan-kernel time with thresholds pinned (decode+ADC vs survivor cost)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deltapq_amd import synth, api, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
cb = synth.make_codebook(8, 256, 16, 100)
qs = synth.make_queries(nq, 128, 101)
tree = synth.synth_tree(n, 8, seed=102)
payload, nb = synth.encode_dtc(tree)
lib = _lib.load()
lib.dpq_debug_scan_time.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
for cps in [int(x) for x in (sys.argv[3].split(',') if len(sys.argv) > 3 else ['4'])]:
    with api.DeltaPQIndex.open_memory(payload, n, 8, 256, chunks_per_segment=cps) as idx:
        idx.set_codebook(cb)
        idx.query_batch(qs, 100)
        for thr in (0, 1):
            for splits in (0, 4, 8, 16, 32):
                ms = ctypes.c_float()
                rc = lib.dpq_debug_scan_time(idx._h, nq, thr, 5, splits, ms)
                assert rc == 0, lib.dpq_last_error()
                pairs = n * nq
                print("cps=%d thr=%s splits=%d: %.3f ms  %.3f T(code,q)/s  alg %.0f GB/s" % (
                    cps, thr, splits, ms.value, pairs / ms.value / 1e9, nq * nb / ms.value / 1e6), flush=True)
                if thr > 0: break
