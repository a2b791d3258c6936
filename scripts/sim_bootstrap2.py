"""CPU simulation: tail of the bootstrap threshold rank for P multi-indexes over disjoint node classes
(node id mod P), sub-space pair (2p, 2p+1) each, cap/P nodes taken from each in shell order."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deltapq_amd import synth
n, nq, k = 1_000_000, int(os.environ.get("NQ", 300)), 100
base = synth.make_clustered_vectors(n, 128, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
queries = synth.make_clustered_vectors(nq, 128, seed=101, n_clusters=20000, spread=12.0, centre_seed=7)
cb = synth.kmeans_codebook(base, 8, 256, iters=6, seed=102)
codes = np.zeros((n, 8), dtype=np.uint8)
for m in range(8):
    c = cb[m].astype(np.float32)
    for lo in range(0, n, 200000):
        x = base[lo:lo + 200000, m * 16:(m + 1) * 16]
        codes[lo:lo + 200000, m] = ((x * x).sum(1)[:, None] - 2.0 * x @ c.T + (c * c).sum(1)[None, :]).argmin(1)
del base
ids_all = np.arange(n)

def build(P):
    out = []
    for p in range(P):
        sel = ids_all[ids_all % P == p]
        cell = codes[sel, 2 * p].astype(np.int32) | (codes[sel, 2 * p + 1].astype(np.int32) << 8)
        order = sel[np.argsort(cell, kind="stable")]
        start = np.searchsorted(np.sort(cell), np.arange(65537))
        out.append((order, start))
    return out

def shells(lut, p, order, start, cap):
    a = np.argsort(lut[2 * p], kind="stable"); b = np.argsort(lut[2 * p + 1], kind="stable")
    got = []; have = 0
    for t in range(256):
        for s in range(2 * t + 1):
            i, j = (t, s) if s <= t else (s - t - 1, t)
            ce = int(a[i]) | (int(b[j]) << 8)
            seg = order[start[ce]:start[ce + 1]]
            if len(seg):
                got.append(seg); have += len(seg)
            if have >= cap: break
        if have >= cap: break
    return np.concatenate(got)[:cap]

res = {}
for P, cap in ((1, 3072), (1, 4096), (2, 3072), (4, 3072), (2, 4096), (4, 4096)):
    mi = build(P)
    ranks = []
    for qi in range(nq):
        q = queries[qi]
        lut = np.stack([((cb[m] - q[m * 16:(m + 1) * 16][None, :]) ** 2).sum(1) for m in range(8)]).astype(np.float64)
        d = lut[np.arange(8)[None, :], codes].sum(1)
        ids = np.concatenate([shells(lut, p, mi[p][0], mi[p][1], cap // P) for p in range(P)])
        thr = np.partition(d[ids], k - 1)[k - 1]
        ranks.append(int((d <= thr).sum()))
    r = np.array(ranks)
    print("P=%d cap=%d: median %5.0f mean %6.0f p90 %6.0f p99 %6.0f max %6.0f" % (P, cap, np.median(r), r.mean(), np.percentile(r, 90), np.percentile(r, 99), r.max()), flush=True)
