"""Why is a step slow?  Bench workload; prints kernel times, overflow reruns and finish() rerun counts (GPU box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from deltapq_amd import api, synth

n, nq, k = int(os.environ.get("N", 1_000_000)), int(os.environ.get("NQ", 1000)), int(os.environ.get("K", 100))
base = synth.make_clustered_vectors(n, 128, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
queries = synth.make_clustered_vectors(nq, 128, seed=101, n_clusters=20000, spread=12.0, centre_seed=7)
cb = synth.kmeans_codebook(base, 8, 256, iters=6, seed=102)
codes = api.encode_pq(base, cb)
del base
tree = api.DeltaTree(codes, codebook=cb, device=0)
payload = tree.payload()
qd = torch.from_numpy(queries).cuda()
for cap in [int(c) for c in os.environ.get("CAPS", "0").split(",")]:
  print("cand_capacity", cap, flush=True)
  with api.DeltaPQIndex.open_memory(payload, n, 8, 256, cand_capacity=cap) as idx:
      idx.set_codebook(cb)
      for _ in range(3):
          idx.query_batch_torch(qd, k)
      torch.cuda.synchronize()
      idx.profile_enable(1)
      idx.profile_reset()
      t0 = time.perf_counter()
      for _ in range(10):
          idx.query_batch_torch(qd, k)
      torch.cuda.synchronize()
      dt = (time.perf_counter() - t0) / 10
      p = idx.profile_read()
      print("sync step %.4f ms; per step: scan %.4f select %.4f check %.4f boot %.4f lut %.4f decode %.4f quantise %.4f; launches scan %d select %d; "
            "overflow reruns %d; survivors/q %.0f candidates/q %.0f" % (
                dt * 1e3, p["scan_ms"] / 10, p["select_ms"] / 10, p.get("check_ms", 0) / 10, p["bootstrap_ms"] / 10, p["lut_ms"] / 10, p["decode_ms"] / 10,
                p["quantise_ms"] / 10, p["scan_launches"], p["select_launches"], p["overflow_reruns"], p["exact_checks"] / (10 * nq),
                p["candidates"] / (10 * nq)), flush=True)
      idx.profile_enable(0)
      outs = [(torch.empty((nq, k), dtype=torch.int32, device="cuda"), torch.empty((nq, k), dtype=torch.float32, device="cuda")) for _ in range(2)]
      for rep in range(3):
          t0 = time.perf_counter()
          for i in range(20):
              idx.query_batch_torch(qd, k, outs[i & 1][0], outs[i & 1][1], wait=False)
          reruns = idx.finish()
          torch.cuda.synchronize()
          print("pipelined: %.4f ms per step, %d batches answered again by finish()" % ((time.perf_counter() - t0) / 20 * 1e3, reruns), flush=True)
