"""Phase times of the bootstrap kernel on the bench workload (GPU box).  argv: bootstrap variants to compare (default: 0 7)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DPQ_DEV"] = "1"
import numpy as np, torch
from deltapq_amd import _lib, api, synth
n, nq, k = 1_000_000, 1000, int(os.environ.get("K", "100"))
M = int(os.environ.get("M", "8"))
variants = sys.argv[1:] or ["0", "1"]   # variant[:target[:cap[:select_fast]]]
lib = _lib.load()
lib.dpq_debug_boot_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
base = synth.make_clustered_vectors(n, 128, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
queries = synth.make_clustered_vectors(nq, 128, seed=101, n_clusters=20000, spread=12.0, centre_seed=7)
cb = synth.kmeans_codebook(base, M, 256, iters=6, seed=102)
codes = api.encode_pq(base, cb)
tree = api.DeltaTree(codes, codebook=cb, device=0)
qd = torch.from_numpy(queries).cuda()
ref = None
for v in variants:
    f = (v.split(":") + ["0", "0", "1"])[:4]
    f = [x or "0" for x in f]
    os.environ["DPQ_BOOT_VARIANT"], os.environ["DPQ_BOOT_TARGET"], os.environ["DPQ_BOOT_CAP"], os.environ["DPQ_SELECT_FAST"] = f   # read once per dpq_open_* (developer mode)
    with api.DeltaPQIndex.open_memory(tree.payload(), n, M, 256) as idx:
        idx.set_codebook(cb)
        out = (ctypes.c_double * 8)()
        ids, dists = idx.query_batch_torch(qd, k)
        if ref is None:
            ref = (ids.clone(), dists.clone())
        else:
            assert torch.equal(ids, ref[0]) and torch.equal(dists.view(torch.int32), ref[1].view(torch.int32)), "variant %s changes the answer" % v
        assert lib.dpq_debug_boot_stamps(idx._h, nq, out) == 0
        for _ in range(3):
            idx.query_batch_torch(qd, k)
        print("== bootstrap variant[:target[:cap[:select_fast]]] %s" % v, flush=True)
        sys.stderr.flush()
        assert lib.dpq_debug_boot_stamps(idx._h, nq, out) == 0
        sys.stderr.flush()
        print("bootstrap phases, mean cycles per block: rank %.0f  cells %.0f  evaluate %.0f  select %.0f" % tuple(out[:4]))
        print("final select phases, mean cycles per block: gather %.0f  kth %.0f  winners %.0f  sort+out %.0f" % tuple(out[4:]), flush=True)
