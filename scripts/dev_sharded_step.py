"""What one rank of an index-sharded run spends per step (GPU box): a 1/8 shard of the bench index, 1000 queries, the
stream-ordered entry followed by pack + (emulated all-gather: the own list 8 times) + merge, on one torch stream or
alternating between two."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deltapq_amd import api, synth, dist as dpq_dist
n, nq, k, world = int(os.environ.get("N", 125_000)), 1000, 100, 8
base = synth.make_clustered_vectors(n, 128, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
cb = synth.kmeans_codebook(base, 8, 256, iters=6, seed=102)
codes = api.encode_pq(base, cb)
tree = api.DeltaTree(codes, codebook=cb, device=0)
batches = [torch.from_numpy(synth.make_clustered_vectors(nq, 128, seed=101 + b, n_clusters=20000, spread=12.0, centre_seed=7)).cuda() for b in range(4)]
with api.DeltaPQIndex.open_memory(tree.payload(), n, 8, 256) as idx:
    idx.set_codebook(cb)
    for n_streams in (1, 2, 1, 2):
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        out = [(torch.empty((nq, k), dtype=torch.int32, device="cuda"), torch.empty((nq, k), dtype=torch.float32, device="cuda")) for _ in range(n_streams)]

        def step(i):
            j = i % n_streams
            with torch.cuda.stream(streams[j]):
                idx.query_batch_torch(batches[i % 4], k, out[j][0], out[j][1], wait=False, ordered=True)
                packed = dpq_dist.pack_lists(out[j][0], out[j][1])
                gathered = packed.unsqueeze(0).expand(world, -1, -1).contiguous()   # stands in for the all-gather
                return api.merge_topk_packed_torch(gathered, k)
        for i in range(6):
            step(i)
        idx.finish(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            step(i)
        idx.finish(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40
        print("N=%d codes, %d stream(s): %.1f us per step (%.2f M queries/s per rank-step)" % (n, n_streams, dt * 1e6, nq / dt / 1e6), flush=True)
