"""GPU probe: what a conflict-aware relabelling of the centroids (balanced 16-colouring of their co-occurrence inside
the ds_read_b128 lane groups, scripts/sim_relabel_conflicts.py) buys the filter scan.  Plain index over the bench codes
in DFS order (same lane neighbourhoods as the DTC scan), labels as they are vs relabelled (+ permuted codebook)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deltapq_amd import _lib, api, synth
n, nq, k, M = 1_000_000, 1000, 100, 8
lib = _lib.load()
lib.dpq_debug_scan_time.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
base = synth.make_clustered_vectors(n, 128, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
queries = synth.make_clustered_vectors(nq, 128, seed=101, n_clusters=20000, spread=12.0, centre_seed=7)
cb = synth.kmeans_codebook(base, 8, 256, iters=6, seed=102)
codes = api.encode_pq(base, cb)
del base
tree = api.DeltaTree(codes, codebook=cb, device=0)
dfs = np.ascontiguousarray(codes[tree.vec_id[:n].astype(np.int64)])
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
nchunk = n // 64
blk = dfs[: nchunk * 64].reshape(nchunk, 64, M)[::5]
labels = np.zeros((M, 256), dtype=np.int64)
t0 = time.time()
for m in range(M):
    Wm = np.zeros((256, 256), dtype=np.int64)
    for g in GROUPS:
        v = blk[:, g, m].astype(np.int64)
        for a in range(16):
            for b in range(a + 1, 16):
                d = v[:, a] != v[:, b]
                np.add.at(Wm, (v[d, a], v[d, b]), 1)
    Wm = Wm + Wm.T
    order = np.argsort(-Wm.sum(axis=1))
    colour = -np.ones(256, dtype=np.int64); load = np.zeros(16, dtype=np.int64); cost = np.zeros((256, 16), dtype=np.int64)
    for kk in order:
        c = min((c for c in range(16) if load[c] < 16), key=lambda c: (cost[kk, c], load[c]))
        colour[kk] = c; load[c] += 1; cost[:, c] += Wm[:, kk]
    rank = np.zeros(256, dtype=np.int64); seen = np.zeros(16, dtype=np.int64)
    for kk in range(256):
        rank[kk] = seen[colour[kk]]; seen[colour[kk]] += 1
    labels[m] = colour + 16 * rank
print("colouring %.0f s" % (time.time() - t0), flush=True)
re_codes = np.stack([labels[m][dfs[:, m]] for m in range(M)], axis=1).astype(np.uint8)
re_cb = np.zeros_like(cb)
for m in range(M):
    re_cb[m, labels[m]] = cb[m]
qd = torch.from_numpy(queries).cuda()
res = {}
for tag, cd, book in (("labels as built", dfs, cb), ("relabelled", re_codes, re_cb)):
    with api.DeltaPQIndex.open_plain(cd) as idx:
        idx.set_codebook(book)
        ids, dists = idx.query_batch_torch(qd, k)
        res[tag] = (ids.cpu().numpy(), dists.cpu().numpy())
        for _ in range(3): idx.query_batch_torch(qd, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): idx.query_batch_torch(qd, k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        out = []
        for mode in (0, 2):
            ms = ctypes.c_float(); assert lib.dpq_debug_scan_time(idx._h, nq, mode, 20, 0, ms) == 0; out.append(ms.value)
        print("%-16s step %.3f ms, filter alone %.4f ms, with final thresholds %.4f ms" % (tag, dt * 1e3, out[0], out[1]), flush=True)
a, b = res["labels as built"], res["relabelled"]
print("same answers:", np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)))
