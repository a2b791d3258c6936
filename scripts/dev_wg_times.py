"""When do the scan's workgroups start and end (GPU box)?  Bench workload, one launch with the last batch's thresholds."""
import ctypes, os, sys
os.environ["DPQ_DEBUG_WG_TIMES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deltapq_amd import _lib, api, synth
n, nq, k = 1_000_000, 1000, int(os.environ.get("TOPK", 100))
lib = _lib.load()
lib.dpq_debug_scan_time.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
base = synth.make_clustered_vectors(n, 128, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
queries = synth.make_clustered_vectors(nq, 128, seed=101, n_clusters=20000, spread=12.0, centre_seed=7)
cb = synth.kmeans_codebook(base, 8, 256, iters=6, seed=102)
codes = api.encode_pq(base, cb)
tree = api.DeltaTree(codes, codebook=cb, device=0)
qd = torch.from_numpy(queries).cuda()
with api.DeltaPQIndex.open_memory(tree.payload(), n, 8, 256) as idx:
    idx.set_codebook(cb)
    for _ in range(3):
        idx.query_batch_torch(qd, k)
    ms = ctypes.c_float()
    for mode, name in ((0, "nothing passes"), (2, "the batch's own thresholds (bootstrap)")):
        assert lib.dpq_debug_scan_time(idx._h, nq, mode, 10, 0, ms) == 0
        print("%s: %.1f us per launch" % (name, ms.value * 1e3), flush=True)
