import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deltapq_amd import synth, api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
for ncl, spread in [(2000, 9.0), (20000, 12.0), (100000, 14.0)]:
    t = time.time()
    base = synth.make_clustered_vectors(n, 128, seed=11, n_clusters=ncl, spread=spread)
    cb = synth.kmeans_codebook(base, 8, 256, iters=6, seed=13)
    t1 = time.time()
    codes = api.encode_pq(base, cb)
    t2 = time.time()
    uniq = len(np.unique(codes.view('u8')))
    tree = api.DeltaTree(codes, codebook=cb)
    t3 = time.time()
    print("clusters=%d spread=%.0f: gen+kmeans %.1fs encode %.1fs build %.1fs unique %.3f B/code %.2f diffs/node %.2f hist %s" % (
        ncl, spread, t1 - t, t2 - t1, t3 - t2, uniq / n, tree.stats['n_bytes'] / n, tree.stats['n_diffs'] / n, tree.stats['depth_hist'][:8]), flush=True)
