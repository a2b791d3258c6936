"""Developer simulation (CPU): LDS cycles of the scan's ds_read_b128 gathers when the 64 nodes of a chunk are dealt to
the four 16-lane read groups differently (the per-batch plain-code scratch could hold a chunk's nodes in any order):
DFS order as it is, sorted by code inside the chunk, and a greedy grouping by shared bytes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deltapq_amd import api, synth

N, M = 1000000, 8
path = "/tmp/sim_filter_%d.npz" % N
if os.path.exists(path):
    z = np.load(path)
    cb, codes = z["cb"], z["codes"]
else:
    base = synth.make_clustered_vectors(N, 128, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
    cb = synth.kmeans_codebook(base, 8, 256, iters=6, seed=102)
    codes = synth.encode_pq_numpy(base, cb) if hasattr(synth, "encode_pq_numpy") else None
    assert codes is not None, "needs the cached codes"
tree = api.DeltaTree(codes, codebook=cb, device=None)
dfs = codes[tree.vec_id]
nchunk = N // 64
sample = np.arange(0, nchunk, 11)
blk = dfs[: nchunk * 64].reshape(nchunk, 64, M)[sample]   # [chunks][node][m]


def cycles(groups):
    """groups: [chunks][4][16] node indices -> mean LDS cycles per (group, m) read"""
    tot = 0.0
    for gi in range(4):
        v = np.take_along_axis(blk, groups[:, gi, :, None].repeat(M, axis=2), axis=1)      # [chunks][16][m]
        v = np.sort(v, axis=1)
        distinct = np.concatenate([np.ones_like(v[:, :1, :], dtype=bool), v[:, 1:, :] != v[:, :-1, :]], axis=1)
        quad = v % 16
        c = np.zeros((v.shape[0], 16, M), dtype=np.int32)
        for lane in range(16):
            np.add.at(c, (np.arange(v.shape[0])[:, None], quad[:, lane, :], np.arange(M)[None, :]), distinct[:, lane, :].astype(np.int32))
        tot += c.max(axis=1).sum()
    return tot / (blk.shape[0] * 4 * M)


HW = np.array([list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))])
C = blk.shape[0]
print("DFS order, hardware groups:        %.3f cycles per group read" % cycles(np.broadcast_to(HW, (C, 4, 16))), flush=True)
print("DFS order, 16 consecutive nodes:    %.3f" % cycles(np.broadcast_to(np.arange(64).reshape(4, 16), (C, 4, 16))), flush=True)
# lexicographic sort inside the chunk
key = np.zeros((C, 64), dtype=np.uint64)
for m in range(M):
    key = (key << np.uint64(8)) | blk[:, :, m].astype(np.uint64)
order = np.argsort(key, axis=1, kind="stable")
print("sorted by code (m = 0 first):       %.3f" % cycles(order.reshape(C, 4, 16)), flush=True)
key = np.zeros((C, 64), dtype=np.uint64)
for m in range(M - 1, -1, -1):
    key = (key << np.uint64(8)) | blk[:, :, m].astype(np.uint64)
order = np.argsort(key, axis=1, kind="stable")
print("sorted by code (m = 7 first):       %.3f" % cycles(order.reshape(C, 4, 16)), flush=True)
# greedy: seed each group with the node farthest (Hamming) from the seeds so far, then add the node closest to the group
t0 = time.time()
sub = min(C, 400)
groups = np.zeros((sub, 4, 16), dtype=np.int64)
for ci in range(sub):
    b = blk[ci]
    ham = (b[:, None, :] != b[None, :, :]).sum(axis=2)
    left = set(range(64))
    for gi in range(4):
        if gi == 0:
            seed = 0
        else:
            seed = max(left, key=lambda j: min(ham[j, groups[ci, g2, 0]] for g2 in range(gi)))
        members = [seed]
        left.discard(seed)
        while len(members) < 16:
            j = min(left, key=lambda j: ham[j, members].sum())
            members.append(j)
            left.discard(j)
        groups[ci, gi] = members
blk_full = blk
blk = blk[:sub]
print("greedy Hamming grouping (%d chunks): %.3f   (DFS on the same chunks: %.3f)" % (sub, cycles(groups), cycles(np.broadcast_to(HW, (sub, 4, 16)))), flush=True)
