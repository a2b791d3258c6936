"""Small-batch latency of the query path (GPU box): nq = 1..64 on the 1M-code pipeline index."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deltapq_amd import synth, api
n, k = 1_000_000, 100
base = synth.make_clustered_vectors(n, 128, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
queries = synth.make_clustered_vectors(256, 128, seed=101, n_clusters=20000, spread=12.0, centre_seed=7)
cb = synth.kmeans_codebook(base, 8, 256, iters=6, seed=102)
codes = api.encode_pq(base, cb)
tree = api.DeltaTree(codes, codebook=cb, device=0)
idx = api.DeltaPQIndex.open_memory(tree.payload(), n, 8, 256)
idx.set_codebook(cb)
qd = torch.from_numpy(queries).cuda()
for nq in (1, 8, 32, 64, 128, 256):
    q = qd[:nq].contiguous()
    for _ in range(5): idx.query_batch_torch(q, k)
    torch.cuda.synchronize()
    idx.profile_enable(1 if os.environ.get('LAT_PROFILE') else 0); idx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(50): idx.query_batch_torch(q, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    p = idx.profile_read(); idx.profile_enable(False)
    print("nq=%2d: %.1f us per call (%.0f q/s)  scan %.1f us select %.1f us lut %.1f us  reruns %d  launches %d  checks/q %.0f cands/q %.0f" % (
        nq, dt * 1e6, nq / dt, p['scan_ms'] / 50 * 1e3, p['select_ms'] / 50 * 1e3, p['lut_ms'] / 50 * 1e3,
        p['overflow_reruns'], p['scan_launches'], p['exact_checks'] / 50 / nq, p['candidates'] / 50 / nq), flush=True)
