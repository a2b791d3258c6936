"""World-size-2 (and 3) run of the sharded path on CPU over gloo."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n,k", [(2, 6000, 50), (3, 5001, 10), (2, 300, 100)])
def test_sharded_gather_and_merge(built, world, n, k):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py"), str(n), str(k)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_packed_exchange_round_trip():
    """The RCCL path sends ids and distance bits in one tensor: packing must be lossless (inf, -1 padding)."""
    import numpy as np
    import torch
    from deltapq_amd import dist as dpq_dist
    rng = np.random.default_rng(3)
    nq, k, world = 7, 5, 3
    ids = [torch.from_numpy(rng.integers(-1, 1 << 30, size=(nq, k)).astype(np.int32)) for _ in range(world)]
    dists = [torch.from_numpy(rng.random((nq, k)).astype(np.float32)) for _ in range(world)]
    dists[1][2, 3:] = float("inf")
    gathered = torch.stack([dpq_dist.pack_lists(i, d) for i, d in zip(ids, dists)])
    assert gathered.shape == (world, nq, 2 * k) and gathered.dtype == torch.int32
    gi, gd = dpq_dist.unpack_lists(gathered, k)
    for r in range(world):
        assert torch.equal(gi[r], ids[r])
        assert torch.equal(gd[r].view(torch.int32), dists[r].view(torch.int32))
