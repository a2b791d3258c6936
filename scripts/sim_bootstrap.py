"""CPU simulation (numpy, no GPU): how tight a first threshold can be.

Compares, on the bench workload (SIFT-shaped mixture, k-means codebook, M=8, top-100), the rank among all N
distances of the threshold produced by
  (a) the cascade's level 0 + level 1 + level 2 (k-th of a spread sample of 3840 / 31 K / 250 K nodes), and
  (b) a TARGETED sample: the nodes of the c x c best cells of a two-sub-space inverted multi-index
      (cell = (code[m0], code[m1])), evaluated exactly.
rank(threshold) = nodes a single filter level over the whole index turns into candidates.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from deltapq_amd import synth

n = int(os.environ.get("N", 1_000_000))
nq, k = int(os.environ.get("NQ", 40)), int(os.environ.get("K", 100))
t0 = time.time()
base = synth.make_clustered_vectors(n, 128, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
queries = synth.make_clustered_vectors(nq, 128, seed=101, n_clusters=20000, spread=12.0, centre_seed=7)
cb = synth.kmeans_codebook(base, 8, 256, iters=6, seed=102)
codes = np.zeros((n, 8), dtype=np.uint8)
for m in range(8):
    c = cb[m].astype(np.float32)
    for lo in range(0, n, 200000):
        x = base[lo:lo + 200000, m * 16:(m + 1) * 16]
        d2 = (x * x).sum(1)[:, None] - 2.0 * x @ c.T + (c * c).sum(1)[None, :]
        codes[lo:lo + 200000, m] = d2.argmin(1)
del base
print("workload %.0f s" % (time.time() - t0), flush=True)
rng = np.random.default_rng(5)
perm = rng.permutation(n)          # stands in for the low-discrepancy visiting order of segments

cell = codes[:, 0].astype(np.int32) | (codes[:, 1].astype(np.int32) << 8)
order = np.argsort(cell, kind="stable")
cell_start = np.searchsorted(cell[order], np.arange(65537))


def targeted(lut, c, stride, cap):
    a = np.argsort(lut[0], kind="stable")[:c]
    b = np.argsort(lut[1], kind="stable")[:c]
    ids = []
    for bb in b:
        for aa in a:
            ce = int(aa) | (int(bb) << 8)
            ids.append(order[cell_start[ce]:cell_start[ce + 1]])
    ids = np.concatenate(ids) if ids else np.zeros(0, dtype=np.int64)
    if stride > 1:
        ids = ids[ids % stride == 0]
    return ids[:cap]


res = {}
for qi in range(nq):
    q = queries[qi]
    lut = np.stack([((cb[m] - q[m * 16:(m + 1) * 16][None, :]) ** 2).sum(1) for m in range(8)]).astype(np.float64)
    d = lut[np.arange(8)[None, :], codes].sum(1)
    ds = np.sort(d)

    def rank_of(t):
        return int(np.searchsorted(ds, t, side="right"))

    def add(name, val):
        res.setdefault(name, []).append(val)

    for s in (3840, 31000, 250000):
        sub = d[perm[:s]]
        add("spread sample %6d" % s, rank_of(np.partition(sub, k - 1)[k - 1]))
    for c, stride, cap in ((8, 1, 4096), (16, 1, 4096), (16, 1, 8192), (32, 4, 4096), (32, 1, 16384), (64, 16, 4096)):
        ids = targeted(lut, c, stride, cap)
        if len(ids) >= k:
            add("targeted c=%d stride=%d cap=%d" % (c, stride, cap), rank_of(np.partition(d[ids], k - 1)[k - 1]))
            add("   nodes evaluated c=%d stride=%d cap=%d" % (c, stride, cap), len(ids))
        else:
            add("targeted c=%d stride=%d cap=%d" % (c, stride, cap), n)
            add("   nodes evaluated c=%d stride=%d cap=%d" % (c, stride, cap), len(ids))
for name, v in res.items():
    v = np.array(v)
    print("%-45s median %8.0f  mean %9.0f  p90 %8.0f  max %8.0f" % (name, np.median(v), v.mean(), np.percentile(v, 90), v.max()))
