"""Developer simulation (CPU, numpy): survivors per query of the scan's lower-bound filter for byte entries (QT 80 /
SAT 26, the product) against 4-bit entries (SAT 15) at several QT, one filter level, threshold = the bootstrap's
(rank ~RANK of the index).  Bench workload."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deltapq_amd import synth
from oracle import pq_encode_oracle

N, M, K, NQ = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 8, 256, 48
RANKS = [800, 400, 200]
cache = "/tmp/sim_filter_%d.npz" % N
if os.path.exists(cache):
    z = np.load(cache)
    cb, codes, queries = z["cb"], z["codes"], z["queries"]
else:
    base = synth.make_clustered_vectors(N, 128, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
    queries = synth.make_clustered_vectors(NQ, 128, seed=101, n_clusters=20000, spread=12.0, centre_seed=7)
    cb = synth.kmeans_codebook(base, M, 256, iters=6, seed=102)
    codes = pq_encode_oracle.encode_pq(base, cb)
    del base
    np.savez(cache, cb=cb, codes=codes, queries=queries)
Ds = 128 // M
CONFIGS = [(80, 26), (80, 24), (80, 23), (80, 22), (72, 23), (72, 22), (64, 24), (64, 23), (64, 22), (64, 21), (56, 23)]


def survivors(T, tau, qt, sat, codes):
    mn = T.min(axis=1)
    R = tau * (1 + 2.0 ** -20) - mn.sum()
    s = qt / R
    e = np.minimum(np.floor((T - mn[:, None]) * s), sat).astype(np.int32)
    sq = e[np.arange(M)[None, :], codes].sum(axis=1)
    return int((sq <= qt).sum())


res = {r: [] for r in RANKS}
for qi in range(min(NQ, len(queries))):
    q = queries[qi].reshape(M, Ds)
    T = ((cb - q[:, None, :]) ** 2).sum(axis=2).astype(np.float64)
    d = T[np.arange(M)[None, :], codes].sum(axis=1)
    for rank in RANKS:
        tau = np.partition(d, rank - 1)[rank - 1]
        res[rank].append([survivors(T, tau, qt, sat, codes) for qt, sat in CONFIGS])
print("configs (QT, SAT):", CONFIGS)
for rank in RANKS:
    r = np.array(res[rank], dtype=np.float64)
    print("threshold rank %4d: mean survivors " % rank + " ".join("%6.0f" % v for v in r.mean(axis=0)) + " | max " + " ".join("%6.0f" % v for v in r.max(axis=0)))
