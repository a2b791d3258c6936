"""Randomised parity stress: HIP path vs oracle over random shapes/options (GPU box).
usage: python scripts/fuzz_parity.py [seconds] [seed]     (DPQ_FUZZ_BIG=1: also shards of up to 400 K nodes)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deltapq_amd import synth, api
from oracle import dtc_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
orc = O.Oracle()
t_end = time.time() + budget
cases = bad = 0
while time.time() < t_end:
    M = int(rng.choice([8, 8, 8, 16]))
    n = int(rng.choice([1, 2, 3, 63, 64, 65, 255, 257, int(rng.integers(300, 3000)), int(rng.integers(3000, 60000))]))
    if os.environ.get("DPQ_FUZZ_BIG") == "1" and rng.random() < 0.3:
        n = int(rng.integers(17000, 400000))  # bootstrap shards
    cps = int(rng.choice([1, 2, 4, 4, 8, 16, 64]))
    k = int(min(n, rng.choice([1, 2, 10, 100, 100, 1000, 2048])))
    nq = int(rng.choice([1, 2, 3, 4, 5, 8, 31, 32, 33, 70, 129, 200, 500, 700]))  # 1-4 (8 with stream_max 8): stream pass; >= 450: in-scan tightening
    # DPQ_OPT_NO_TIGHTEN 16, DPQ_OPT_FORCE_STRANDS 64 (one query per call on a bootstrap shard: strand1_kernel), 64 | 16 (its
    # multi-level plan without in-kernel tightening), 64 | 128 (DPQ_OPT_NO_STRAND1: the exact-table kernel)
    flags = int(rng.choice([0, 0, 16, 64, 64, 64, 80, 192]))
    boot = int(rng.choice([0, 0, 1]))           # 1: the threshold bootstrap (and with it the strand image) from 16 K nodes
    smax = int(rng.choice([0, 0, 8, -1]))
    bd = int(rng.choice([-1, 0, 0, 1, 1, 2, 5, 37, 300]))  # dpq_open_opts.batch_decode (>= 2: scratch tiles of that many segments)
    cap = int(rng.choice([0, 0, 0, 64, 300]))
    shards = int(rng.choice([1, 1, 1, 2, 5]))
    K = int(rng.choice([256, 256, 256, 17, 100]))
    md = float(rng.choice([0.2, 1.0, 3.0, 6.0]))
    seed = int(rng.integers(1 << 30))
    cb = synth.make_codebook(M, K, 128 // M, seed)
    tree = synth.synth_tree(n, M, seed=seed + 1, mean_diffs=md)
    tree["deltas"] = (tree["deltas"].astype(np.int64) % K).astype(np.uint8)
    tree["root"] = (tree["root"].astype(np.int64) % K).astype(np.uint8)
    payload, nb = synth.encode_dtc(tree)
    qs = synth.make_queries(nq, 128, seed + 2)
    desc = "M=%d n=%d cps=%d k=%d nq=%d cap=%d shards=%d K=%d md=%.1f seed=%d batch_decode=%d flags=%d stream_max=%d bootstrap=%d" % (
        M, n, cps, k, nq, cap, shards, K, md, seed, bd, flags, smax, boot)
    try:
        parts = []
        for r in range(shards):
            with api.DeltaPQIndex.open_memory(payload, n, M, K, chunks_per_segment=cps, cand_capacity=cap,
                                              shard_rank=r, shard_count=shards, batch_decode=bd, flags=flags,
                                              stream_max_queries=smax, bootstrap=boot) as idx:
                idx.set_codebook(cb)
                parts.append(idx.query_batch(qs, k))
        if shards > 1:
            ids, dists = api.merge_topk_host(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
        else:
            ids, dists = parts[0]
        check = range(nq) if nq <= 40 else sorted(set(range(12)) | set(int(v) for v in rng.integers(0, nq, 24)))
        for i in check:
            lut = orc.build_lut(cb, qs[i])
            oi, od, alld, _ = orc.scan_lut(payload, n, lut, k, want_all=True)
            ok, msg = O.tie_aware_equal(ids[i], dists[i], oi, od, alld, n)
            if not ok:
                bad += 1
                print("MISMATCH %s q%d: %s" % (desc, i, msg), flush=True)
                break
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("ERROR %s: %r" % (desc, e), flush=True)
    cases += 1
    if cases % 25 == 0:
        print("%d cases, %d bad" % (cases, bad), flush=True)
print("FUZZ DONE: %d cases, %d bad" % (cases, bad))
sys.exit(1 if bad else 0)
