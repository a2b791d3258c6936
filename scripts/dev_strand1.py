"""One query per call on a big stream-synthesised shard (GPU box): the stream pass's kernel time (HIP events), wall time
per synchronous call, candidates / exact checks per call, and a check of the first queries against the oracle.
usage: python scripts/dev_strand1.py [--codes N] [--calls C] [--check Q] [--flags F] [--queries q]
The payload is cached in /tmp (the library variants of one gpurun call share it): DPQ_LIB_PATH picks the library."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deltapq_amd import synth, api

ap = argparse.ArgumentParser()
ap.add_argument("--codes", type=int, default=125_000_000)
ap.add_argument("--calls", type=int, default=30)
ap.add_argument("--check", type=int, default=0)
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--queries", type=int, default=1)
ap.add_argument("--topk", type=int, default=100)
ap.add_argument("--tag", default="")
a = ap.parse_args()
n, k = a.codes, a.topk
cache = "/tmp/dpq_stream_%d.npy" % n
t0 = time.time()
if os.path.exists(cache):
    payload = np.load(cache, mmap_mode="r")
else:
    tree = synth.synth_tree_large(n, 8, seed=102, mean_diffs=3.0)
    payload, _ = synth.encode_dtc(tree)
    del tree
    np.save(cache, payload)
cb = synth.make_codebook(8, 256, 16, seed=100)
qs_np = synth.make_queries(64, 128, seed=101)
qs = torch.from_numpy(qs_np).cuda()
t1 = time.time()
with api.DeltaPQIndex.open_memory(np.asarray(payload), n, 8, 256, flags=a.flags) as idx:
    idx.set_codebook(cb)
    info = idx.info()
    t2 = time.time()
    nq = a.queries
    ids0, d0 = idx.query_batch(qs_np[:nq], k)
    for i in range(3):
        idx.query_batch_torch(qs[i * nq:(i + 1) * nq].contiguous(), k)
    torch.cuda.synchronize()
    idx.profile_enable(1)
    idx.profile_reset()
    tw = time.perf_counter()
    for i in range(a.calls):
        j = (i % (64 // nq)) * nq
        idx.query_batch_torch(qs[j:j + nq].contiguous(), k)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - tw) / a.calls
    p = idx.profile_read()
    c = a.calls
    pay = info["algorithmic_bytes"]
    scan = p["scan_ms"] / c
    print("%s n=%d nq=%d flags=%d: scan %.1f us/call (%d launches: stream %d strand %d strand1 %d) = %.2f TB/s of payload = %.3f of 8 TB/s | "
          "call %.1f us wall (%.3f), bootstrap %.1f select %.1f lut %.1f us | candidates %.0f exact checks %.0f per call, reruns %d | "
          "image %.0f MB strand %.0f MB payload %.0f MB | gen %.0f s open %.0f s"
          % (a.tag, n, nq, a.flags, scan * 1e3, p["scan_launches"] / c, p["stream_launches"] / c, p["strand_launches"] / c, p["strand1_launches"] / c,
             pay / (scan * 1e-3) / 1e12, pay / (scan * 1e-3) / 8e12, wall * 1e6, pay / wall / 8e12, p["bootstrap_ms"] / c * 1e3,
             p["select_ms"] / c * 1e3, p["lut_ms"] / c * 1e3, p["candidates"] / c, p["exact_checks"] / c, p["overflow_reruns"],
             info["device_bytes"] / 1e6, info["strand_bytes"] / 1e6, pay / 1e6, t1 - t0, t2 - t1), flush=True)
    if a.check > 0:
        from oracle import dtc_oracle as O
        orc = O.Oracle()
        for i in range(min(a.check, nq * 1 if nq > 1 else a.check)):
            ids_i, d_i = (ids0[i], d0[i]) if i < nq else idx.query_batch(qs_np[i:i + 1], k)
            if i >= nq:
                ids_i, d_i = ids_i[0], d_i[0]
            lut = orc.build_lut(cb, qs_np[i])
            oi, od, alld, _ = orc.scan_lut(np.asarray(payload), n, lut, k, want_all=True)
            ok, msg = O.tie_aware_equal(ids_i, d_i, oi, od, alld, n)
            print("  oracle check query %d: %s %s" % (i, "OK" if ok else "MISMATCH", msg if not ok else ""), flush=True)
