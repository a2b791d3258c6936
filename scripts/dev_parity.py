"""Developer parity probe: HIP path vs oracle on seeded synthetic indexes (GPU box)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deltapq_amd import synth, api
from oracle import dtc_oracle as O

orc = O.Oracle()
ok_all = True
CASES = [(8, 1, 3, 1, 4), (8, 2, 3, 2, 4), (8, 65, 5, 10, 1), (8, 1000, 20, 10, 4), (8, 10000, 100, 10, 4),
         (8, 9999, 33, 100, 2), (8, 100000, 64, 100, 4), (8, 300001, 40, 100, 4),
         (16, 1, 3, 1, 4), (16, 2, 3, 2, 4), (16, 65, 5, 10, 1), (16, 1000, 20, 10, 4), (16, 10000, 50, 100, 4),
         (16, 100001, 40, 1000, 4), (16, 300000, 33, 100, 2)]
for (M, n, nq, k, cps) in CASES:
    cb = synth.make_codebook(M, 256, 128 // M, 0)
    tree = synth.synth_tree(n, M, seed=n, mean_diffs=3.0 if M == 8 else 5.0)
    payload, nb = synth.encode_dtc(tree)
    qs = synth.make_queries(nq, 128, seed=n + 1)
    t0 = time.time()
    with api.DeltaPQIndex.open_memory(payload, n, M, 256, chunks_per_segment=cps) as idx:
        idx.set_codebook(cb)
        idx.profile_enable(True)
        ids, dists = idx.query_batch(qs, k)
        prof = idx.profile_read()
    t1 = time.time()
    bad = 0
    for i in range(nq):
        lut = orc.build_lut(cb, qs[i])
        oi, od, alld, _ = orc.scan_lut(payload, n, lut, k, want_all=True)
        ok, msg = O.tie_aware_equal(ids[i], dists[i], oi, od, alld, n)
        if not ok:
            bad += 1
            if bad <= 3:
                print("  MISMATCH q%d: %s" % (i, msg)); print("   gpu", ids[i][:8], dists[i][:4]); print("   ref", oi[:8], od[:4])
    print("M=%d n=%d nq=%d k=%d cps=%d: %s (%d bad) gpu %.3fs scan_ms=%.3f launches=%d reruns=%d" % (
        M, n, nq, k, cps, "OK" if bad == 0 else "FAIL", bad, t1 - t0, prof['scan_ms'], prof['scan_launches'], prof['overflow_reruns']), flush=True)
    ok_all &= bad == 0
print("ALL OK" if ok_all else "SOME FAILED")
sys.exit(0 if ok_all else 1)
