"""Where the scan kernel's cycles go (GPU box).  Bench workload (pipeline-built 1 M index, 1000 queries, top-100).

1. STAMPS build of scan_kernel: s_memtime brackets around prologue / segment begin / decode / ADC gathers /
   mask fold / queue push / refine, summed over all wavefronts of ONE full-index launch with the final thresholds.
2. The product kernel on the same launch: nothing passes (pure filter) vs final thresholds.
3. Compressed vs plain scan with the plain codes in DFS order (codes[vec_id]: identical lane neighbourhoods,
   so the difference is the decode) and in file order (the difference to DFS order is lane locality).
"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from deltapq_amd import _lib, api, synth

n, nq, k = int(os.environ.get("N", 1_000_000)), 1000, 100
STAMP_NAMES = ["prologue", "segment_begin", "decode", "adc_gather", "fold", "push", "refine", "total", "steps",
               "refines", "waves", "pairs_checked"]
lib = _lib.load()
lib.dpq_debug_scan_time.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.POINTER(ctypes.c_float)]
lib.dpq_debug_scan_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong),
                                      ctypes.c_int, ctypes.POINTER(ctypes.c_float)]

base = synth.make_clustered_vectors(n, 128, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
queries = synth.make_clustered_vectors(nq, 128, seed=101, n_clusters=20000, spread=12.0, centre_seed=7)
cb = synth.kmeans_codebook(base, 8, 256, iters=6, seed=102)
codes = api.encode_pq(base, cb)
del base
tree = api.DeltaTree(codes, codebook=cb, device=0)
payload = tree.payload()
vec_id = tree.vec_id[:n].astype(np.int64)
codes_dfs = np.ascontiguousarray(codes[vec_id])
qd = torch.from_numpy(queries).cuda()
out = {}


def stamps(idx, tag):
    buf = (ctypes.c_ulonglong * 16)()
    ms = ctypes.c_float()
    rc = lib.dpq_debug_scan_stamps(idx._h, nq, 0, buf, 16, ms)
    assert rc == 0, lib.dpq_last_error()
    v = dict(zip(STAMP_NAMES, list(buf)[:len(STAMP_NAMES)]))
    steps, waves = max(1, v["steps"]), max(1, v["waves"])
    res = {"launch_ms": ms.value, "wave_steps": v["steps"], "waves": v["waves"], "refines": v["refines"],
           "pairs_checked": v["pairs_checked"],
           "cycles_per_wave_step": {s: v[s] / steps for s in ("segment_begin", "decode", "adc_gather", "fold", "push", "refine")},
           "prologue_cycles_per_wave": v["prologue"] / waves, "total_cycles_per_wave": v["total"] / waves,
           "share_of_wave_time": {s: v[s] / max(1, v["total"]) for s in ("prologue", "segment_begin", "decode", "adc_gather", "fold", "push", "refine")}}
    out["stamps_" + tag] = res
    print(tag, json.dumps(res), flush=True)


def timed(idx, tag):
    idx.set_codebook(cb)
    for _ in range(3):
        idx.query_batch_torch(qd, k)
    torch.cuda.synchronize()
    idx.profile_enable(1)
    idx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(20):
        idx.query_batch_torch(qd, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    p = idx.profile_read()
    idx.profile_enable(0)
    r = {"ms_per_step": dt * 1e3, "scan_ms": p["scan_ms"] / 20, "select_ms": p["select_ms"] / 20,
         "quantise_ms": p["quantise_ms"] / 20, "lut_ms": p["lut_ms"] / 20,
         "exact_checks_per_query": p["exact_checks"] / (20 * nq), "candidates_per_query": p["candidates"] / (20 * nq)}
    for mode, name in ((0, "filter_nothing_passes_ms"), (2, "filter_final_thresholds_ms")):
        ms = ctypes.c_float()
        rc = lib.dpq_debug_scan_time(idx._h, nq, mode, 20, 0, ms)
        assert rc == 0, lib.dpq_last_error()
        r[name] = ms.value
    out[tag] = r
    print(tag, json.dumps(r), flush=True)


with api.DeltaPQIndex.open_memory(payload, n, 8, 256) as idx:
    timed(idx, "dtc")
    stamps(idx, "dtc")
with api.DeltaPQIndex.open_plain(codes_dfs) as idx:
    timed(idx, "plain_dfs_order")
    stamps(idx, "plain_dfs_order")
with api.DeltaPQIndex.open_plain(codes) as idx:
    timed(idx, "plain_file_order")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/limiter_%s.json" % os.environ.get("TAG", "r02"), "w"), indent=1)
