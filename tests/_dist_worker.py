"""Worker of tests/test_dist_gloo.py: one rank of the sharded query path on CPU.

The GPU scan is replaced by the oracle's per-node distances restricted to this
rank's shard (test infrastructure); everything else is the product's N > 1
path: byte-balanced shard ranges from the C-ABI transcoder, ONE all-gather of
the partial top-k lists, merge by (distance, id) through dpq_merge_topk_host.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from deltapq_amd import api, synth                 # noqa: E402
from deltapq_amd import dist as dpq_dist           # noqa: E402
from oracle import dtc_oracle as O                 # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, nq, k = int(sys.argv[1]), 6, int(sys.argv[2])
    cb = synth.make_codebook(8, 256, 16, seed=0)
    qs = synth.make_queries(nq, 128, seed=1)
    tree = synth.synth_tree(n, 8, seed=2, mean_diffs=0.6)          # duplicate-heavy: ties cross shard borders
    payload, _ = synth.encode_dtc(tree)
    soa = api.HostSoA(payload, n, 8, shard_rank=rank, shard_count=world)
    lo, hi = soa.info["node_lo"], soa.info["node_hi"]
    orc = O.Oracle()
    ids = np.full((nq, k), -1, np.int32)
    dists = np.full((nq, k), np.inf, np.float32)
    ref = []
    for i in range(nq):
        lut = orc.build_lut(cb, qs[i])
        oi, od, alld, _ = orc.scan_lut(payload, n, lut, k, want_all=True)
        ref.append((oi, od, alld))
        pos = np.arange(lo, hi, dtype=np.int32)
        rep = pos.copy()
        if n % 2 == 0:
            rep[pos == n - 1] = n                                   # even-N quirk is applied by each shard
        ids[i], dists[i] = dpq_dist.partial_topk_rows(rep, alld[lo:hi], k)
    mi, md = dpq_dist.gather_and_merge(torch.from_numpy(ids), torch.from_numpy(dists))
    ok = True
    for i in range(nq):
        good, msg = O.tie_aware_equal(mi[i].numpy(), md[i].numpy(), ref[i][0], ref[i][1], ref[i][2], n)
        if not good:
            ok = False
            print("rank %d query %d: %s" % (rank, i, msg), flush=True)
    covered = torch.tensor([hi - lo], dtype=torch.int64)
    dist.all_reduce(covered)
    ok = ok and int(covered.item()) == n
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("DIST_OK" if int(flag.item()) == 1 else "DIST_FAIL", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
