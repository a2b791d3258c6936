"""Worker of test_two_process_sharded_run_on_one_gpu: one rank of the sharded
GPU query path.  Both ranks use GPU 0 (a 1-GPU box), so the exchange goes over
gloo with CPU tensors; on an 8-GPU node bench.py does the same over RCCL."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deltapq_amd import api, synth                 # noqa: E402
from deltapq_amd import dist as dpq_dist           # noqa: E402
from oracle import dtc_oracle as O                 # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, nq, k = 200_001, 24, 100
    cb = synth.make_codebook(8, 256, 16, seed=0)
    qs = synth.make_queries(nq, 128, seed=1)
    tree = synth.synth_tree(n, 8, seed=2)
    payload, _ = synth.encode_dtc(tree)
    with api.DeltaPQIndex.open_memory(payload, n, 8, 256, device=0, shard_rank=rank, shard_count=world) as idx:
        idx.set_codebook(cb)
        ids, dists = idx.query_batch(qs, k)
    mi, md = dpq_dist.gather_and_merge(torch.from_numpy(ids), torch.from_numpy(dists))
    ok = True
    if rank == 0:
        orc = O.Oracle()
        for i in range(nq):
            lut = orc.build_lut(cb, qs[i])
            oi, od, alld, _ = orc.scan_lut(payload, n, lut, k, want_all=True)
            good, msg = O.tie_aware_equal(mi[i].numpy(), md[i].numpy(), oi, od, alld, n)
            if not good:
                ok = False
                print("query %d: %s" % (i, msg), flush=True)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("DIST_GPU_OK" if int(flag.item()) == 1 else "DIST_GPU_FAIL", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
