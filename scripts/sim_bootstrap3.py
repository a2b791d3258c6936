"""CPU simulation: bootstrap threshold rank when the cells are visited in the order of a PRECOMPUTED neighbour list of
the query's nearest centroid (per sub-space: centroids sorted by distance to that centroid) instead of the exact order
of the query's own table row.  P = 4 classes, 3072 nodes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deltapq_amd import synth
n, nq, k, P, cap = 1_000_000, int(os.environ.get("NQ", 300)), 100, 4, 3072
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
nbr = np.zeros((8, 256, 256), dtype=np.int64)      # nbr[m][c] = centroids of sub-space m by distance to centroid c
for m in range(8):
    d2 = ((cb[m][:, None, :] - cb[m][None, :, :]) ** 2).sum(2)
    nbr[m] = np.argsort(d2, axis=1, kind="stable")
mi = []
for p in range(P):
    sel = ids_all[ids_all % P == p]
    cell = codes[sel, 2 * p].astype(np.int32) | (codes[sel, 2 * p + 1].astype(np.int32) << 8)
    mi.append((sel[np.argsort(cell, kind="stable")], np.searchsorted(np.sort(cell), np.arange(65537))))

def walk(a, b, order, start, want):
    got, have = [], 0
    for w in range(65536):
        t = int(np.sqrt(w))
        while t * t > w: t -= 1
        while (t + 1) * (t + 1) <= w: t += 1
        s = w - t * t
        i, j = (t, s) if s <= t else (s - t - 1, t)
        ce = int(a[i]) | (int(b[j]) << 8)
        seg = order[start[ce]:start[ce + 1]]
        if len(seg):
            got.append(seg); have += len(seg)
        if have >= want: break
    return np.concatenate(got)[:want]

res = {"exact order": [], "neighbour list of the nearest centroid": [], "exact order among the 32 nearest neighbours of the nearest centroid": [],
       "exact order among the 64 nearest neighbours": []}
for qi in range(nq):
    q = queries[qi]
    lut = np.stack([((cb[m] - q[m * 16:(m + 1) * 16][None, :]) ** 2).sum(1) for m in range(8)]).astype(np.float64)
    d = lut[np.arange(8)[None, :], codes].sum(1)
    for name in res:
        ids = []
        for p in range(P):
            if name == "exact order":
                a, b = np.argsort(lut[2 * p], kind="stable"), np.argsort(lut[2 * p + 1], kind="stable")
            elif name.startswith("neighbour"):
                a, b = nbr[2 * p][int(lut[2 * p].argmin())], nbr[2 * p + 1][int(lut[2 * p + 1].argmin())]
            else:
                R = 32 if "32" in name else 64
                ab = []
                for s_ in (2 * p, 2 * p + 1):
                    cand = nbr[s_][int(lut[s_].argmin())]
                    head = cand[:R][np.argsort(lut[s_][cand[:R]], kind="stable")]
                    ab.append(np.concatenate([head, cand[R:]]))
                a, b = ab
            ids.append(walk(a, b, mi[p][0], mi[p][1], cap // P))
        ids = np.concatenate(ids)
        thr = np.partition(d[ids], k - 1)[k - 1]
        res[name].append(int((d <= thr).sum()))
for name, r in res.items():
    r = np.array(r)
    print("%-40s median %5.0f mean %6.0f p90 %6.0f p99 %6.0f max %6.0f" % (name, np.median(r), r.mean(), np.percentile(r, 90), np.percentile(r, 99), r.max()))
