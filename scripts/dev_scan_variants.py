"""Scan-kernel A/B on the bench workload (GPU box): one product batch, then the first filter level's scan launch timed
alone -- nothing passes / final thresholds / the bootstrap's thresholds (what the product launch sees) -- plus the
pipelined step.  Run once per library variant: DPQ_LIB_PATH=variants/lib_x.so python scripts/dev_scan_variants.py"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from deltapq_amd import _lib, api, synth

n, nq, k, M = int(os.environ.get("N", 1_000_000)), int(os.environ.get("NQ", 1000)), int(os.environ.get("K", 100)), int(os.environ.get("M", 8))
cache = "/tmp/dpq_variant_index_%d_%d.npz" % (n, M)
if os.path.exists(cache):
    z = np.load(cache)
    payload, cb, queries = z["payload"], z["cb"], z["queries"]
else:
    base = synth.make_clustered_vectors(n, 128, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
    queries = synth.make_clustered_vectors(nq, 128, seed=101, n_clusters=20000, spread=12.0, centre_seed=7)
    cb = synth.kmeans_codebook(base, M, 256, iters=6, seed=102)
    codes = api.encode_pq(base, cb)
    del base
    tree = api.DeltaTree(codes, codebook=cb, device=0)
    payload = tree.payload()
    np.savez(cache, payload=payload, cb=cb, queries=queries)
lib = _lib.load()
lib.dpq_debug_scan_time.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
qd = torch.from_numpy(queries[:nq]).cuda()
with api.DeltaPQIndex.open_memory(payload, n, M, 256) as idx:
    idx.set_codebook(cb)
    for _ in range(3):
        idx.query_batch_torch(qd, k)
    torch.cuda.synchronize()
    idx.profile_enable(1)
    idx.profile_reset()
    for _ in range(10):
        idx.query_batch_torch(qd, k)
    torch.cuda.synchronize()
    p = idx.profile_read()
    if os.environ.get("DPQ_PROGRESS"): print("product batches done", p, flush=True)
    idx.profile_enable(0)
    idx.query_batch_torch(qd, k)   # the launch the timings below repeat: as the product runs it, without statistics
    torch.cuda.synchronize()
    res = {}
    for mode, name in ((0, "none"), (2, "final"), (3, "boot")):
        if mode == 3 and "r02" in os.environ.get("DPQ_LIB_PATH", ""):   # the round-2 library has no such mode
            res[name] = float("nan")
            continue
        ms = ctypes.c_float()
        rc = lib.dpq_debug_scan_time(idx._h, nq, mode, 20, 0, ms)
        assert rc == 0, lib.dpq_last_error()
        res[name] = ms.value
        if os.environ.get("DPQ_PROGRESS"): print("mode", name, ms.value, flush=True)
    outs = [(torch.empty((nq, k), dtype=torch.int32, device="cuda"), torch.empty((nq, k), dtype=torch.float32, device="cuda")) for _ in range(2)]
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        for i in range(20):
            idx.query_batch_torch(qd, k, outs[i & 1][0], outs[i & 1][1], wait=False)
        idx.finish()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    print("%-28s scan alone: none %.4f final %.4f boot %.4f | product: scan %.4f select %.4f check %.4f boot %.4f lut %.4f dec %.4f | pipelined step %.4f ms | surv/q %.0f cand/q %.0f reruns %d" % (
        os.path.basename(os.environ.get("DPQ_LIB_PATH", "in-tree")), res["none"], res["final"], res["boot"], p["scan_ms"] / 10, p["select_ms"] / 10, p.get("check_ms", 0) / 10,
        p["bootstrap_ms"] / 10, p["lut_ms"] / 10, p["decode_ms"] / 10, best, p["exact_checks"] / (10 * nq), p["candidates"] / (10 * nq), p["overflow_reruns"]), flush=True)
