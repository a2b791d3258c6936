"""Developer simulation (CPU, numpy): survivors per query of the scan's lower-bound filter
at 16-bit and 8-bit table entries, on the bench's default workload and cascade plan."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deltapq_amd import synth
from oracle import pq_encode_oracle

N, M, K, NQ, TOPK = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 8, 256, 24, 100
t0 = time.time()
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
print("data %.0f s" % (time.time() - t0), flush=True)
rng = np.random.default_rng(5)
perm = rng.permutation(N)
codes = codes[perm]
bounds = [4096, N // 256 * 8 // 8, N // 32, N // 4, N]  # level 0 | {8, 8, 4} from the top
bounds = [4096, max(4096, N // 256), N // 32, N // 4, N]
Ds = 128 // M


def quant(T, tau, qt, sat):
    mn = T.min(axis=1)
    R = tau * (1 + 2.0 ** -20) - mn.sum()
    s = qt / R
    e = np.minimum(np.floor((T - mn[:, None]) * s), sat).astype(np.int64)
    Q = int(np.ceil(R * s))
    return e, Q


CONFIGS = [(7500, 8191), (80, 26), (90, 27), (96, 28), (104, 29), (112, 30), (120, 31), (90, 31)]
tot = {}
for qi in range(NQ):
    q = queries[qi].reshape(M, Ds)
    T = ((cb - q[:, None, :]) ** 2).sum(axis=2).astype(np.float64)  # [M][K]
    d = T[np.arange(M)[None, :], codes].sum(axis=1)
    for l in range(1, len(bounds) - 1 + 1):
        lo, hi = bounds[l - 1], bounds[l]
        if hi <= lo:
            continue
        tau = np.partition(d[:lo], TOPK - 1)[TOPK - 1]
        seg = slice(lo, hi)
        exact = int((d[seg] <= tau).sum())
        row = [exact]
        for qt, sat in CONFIGS:
            e, Q = quant(T, tau, qt, sat)
            sq = e[np.arange(M)[None, :], codes[seg]].sum(axis=1)
            row.append(int((sq <= Q).sum()))
        tot.setdefault(l, []).append(row)
print("configs (QT, SAT):", CONFIGS)
for l, rows in tot.items():
    r = np.array(rows, dtype=np.float64)
    print("level %d (%d..%d): exact %.0f | " % (l, bounds[l - 1], bounds[l], r[:, 0].mean()) + " ".join("%.0f" % v for v in r[:, 1:].mean(axis=0)))
