"""Developer parity probe for the corners of the query path (GPU box): extreme top_k, batch sizes around
the group size, one-segment-sized level 0, tiny and huge candidate buffers, pipelined calls."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deltapq_amd import synth, api
from oracle import dtc_oracle as O

orc = O.Oracle()
ok_all = True
# (M, n, nq, k, cps, cand_capacity, sample of queries to check)
CASES = [(8, 2048, 5, 2048, 2, 0, 5), (8, 5000, 3, 2047, 2, 0, 3), (8, 100000, 65, 1, 2, 0, 20), (8, 100000, 63, 100, 64, 0, 10),
         (8, 100000, 129, 100, 1, 0, 10), (8, 200000, 2049, 10, 2, 0, 12), (8, 150000, 70, 500, 2, 600, 10),
         (8, 150000, 70, 100, 2, 100, 10), (8, 150000, 10, 2048, 4, 0, 4), (16, 50000, 17, 2048, 2, 0, 4),
         (16, 120000, 70, 200, 2, 256, 8), (8, 3, 4, 3, 2, 0, 4), (8, 129, 64, 129, 2, 0, 8)]
for (M, n, nq, k, cps, cap, ncheck) in CASES:
    cb = synth.make_codebook(M, 256, 128 // M, 1)
    tree = synth.synth_tree(n, M, seed=n + k, mean_diffs=3.0 if M == 8 else 5.0)
    payload, nb = synth.encode_dtc(tree)
    qs = synth.make_queries(nq, 128, seed=n + 2)
    with api.DeltaPQIndex.open_memory(payload, n, M, 256, chunks_per_segment=cps, cand_capacity=cap) as idx:
        idx.set_codebook(cb)
        idx.profile_enable(True)
        ids, dists = idx.query_batch(qs, k)
        prof = idx.profile_read()
    bad = 0
    for i in np.linspace(0, nq - 1, ncheck).astype(int):
        lut = orc.build_lut(cb, qs[i])
        oi, od, alld, _ = orc.scan_lut(payload, n, lut, k, want_all=True)
        ok, msg = O.tie_aware_equal(ids[i], dists[i], oi, od, alld, n)
        if not ok:
            bad += 1
            if bad <= 2:
                print("  MISMATCH q%d: %s" % (i, msg))
    print("M=%d n=%d nq=%d k=%d cps=%d cap=%d: %s (%d bad) launches=%d reruns=%d" % (
        M, n, nq, k, cps, cap, "OK" if bad == 0 else "FAIL", bad, prof['scan_launches'], prof['overflow_reruns']), flush=True)
    ok_all &= bad == 0
print("ALL OK" if ok_all else "SOME FAILED")
sys.exit(0 if ok_all else 1)
