"""Compressed (DTC) vs plain scan on the same pipeline-built 1M index (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deltapq_amd import synth, api
n, nq, k = 1_000_000, 1000, 100
base = synth.make_clustered_vectors(n, 128, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
queries = synth.make_clustered_vectors(nq, 128, seed=101, n_clusters=20000, spread=12.0, centre_seed=7)
cb = synth.kmeans_codebook(base, 8, 256, iters=6, seed=102)
codes = api.encode_pq(base, cb)
tree = api.DeltaTree(codes, codebook=cb)
payload = tree.payload()
qd = torch.from_numpy(queries).cuda()
for name, idx in [("dtc  ", api.DeltaPQIndex.open_memory(payload, n, 8, 256)), ("plain", api.DeltaPQIndex.open_plain(codes))]:
    idx.set_codebook(cb)
    for _ in range(3): idx.query_batch_torch(qd, k)
    torch.cuda.synchronize()
    idx.profile_enable(True); idx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(20): idx.query_batch_torch(qd, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    p = idx.profile_read(); info = idx.info()
    print("%s: %.3f ms/step  %.0f q/s  scan %.3f ms select %.3f ms  bytes/code %.2f  device MB %.1f" % (
        name, dt * 1e3, nq / dt, p['scan_ms'] / 20, p['select_ms'] / 20, info['algorithmic_bytes'] / n, info['device_bytes'] / 1e6), flush=True)
    idx.close()
