"""Single-call latency of the stream pass on a big shard (GPU box): synchronous calls, one at a time.
usage: python scripts/dev_latency_big.py [n_codes]   (DPQ_DEV=1 DPQ_STRANDS=0/1/2 to pick the pass)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deltapq_amd import synth, api
n, k = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000, 100
tree = synth.synth_tree_large(n, 8, seed=7, mean_diffs=3.0)
payload, _ = synth.encode_dtc(tree)
del tree
cb = synth.make_codebook(8, 256, 16, seed=3)
qs = torch.from_numpy(synth.make_queries(16, 128, seed=5)).cuda()
with api.DeltaPQIndex.open_memory(payload, n, 8, 256) as idx:
    idx.set_codebook(cb)
    for nq in (1, 2, 4):
        q = qs[:nq].contiguous()
        for _ in range(5):
            idx.query_batch_torch(q, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            idx.query_batch_torch(q, k)
        torch.cuda.synchronize()
        print("n=%d nq=%d: %.1f us per synchronous call (DPQ_STRANDS=%s)" % (n, nq, (time.perf_counter() - t0) / 50 * 1e6, os.environ.get("DPQ_STRANDS", "default")), flush=True)
