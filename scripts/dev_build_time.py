"""End-to-end DeltaTree build time of 1 M SIFT-shaped codes (GPU box): edge search + layout on the GPU vs layout on the host."""
import os
import sys, time, subprocess
os.environ.setdefault("DPQ_DEV", "1")   # developer switches of the library are read only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deltapq_amd import api, synth
n = 1_000_000
base = synth.make_clustered_vectors(n, 128, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
cb = synth.kmeans_codebook(base, 8, 256, iters=6, seed=102)
codes = api.encode_pq(base, cb)
del base
api.DeltaTree(codes[:1000], codebook=cb, device=0)     # warm up (module load, hipCUB temp sizes)
for it in range(2):
    t0 = time.time(); t = api.DeltaTree(codes, codebook=cb, device=0); t1 = time.time() - t0
    print("DeltaTree(1M codes, codebook, device=0) [%s layout]: %.3f s, %.2f diffs/node, max depth %d" % (
        os.environ.get("DPQ_BUILD_LAYOUT", "gpu"), t1, t.stats["n_diffs"] / n, t.stats["max_depth"]), flush=True)
