"""GPU parity tests: the HIP path, called through the C-ABI, against the oracle
on the same seeded inputs.  Bar: ids bit-exact up to exact-distance ties
(tie-aware comparator), fp32 distances BIT-equal (the north star
allows 1e-5 relative; we hold the stronger bar)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import assert_parity, make_case, oracle_topk

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def gpu(lib):
    from deltapq_amd import api
    if api.device_count() < 1:
        pytest.fail("no GPU visible: the HIP path is the product and must be what runs here")
    return api


def run(gpu, payload, n, cb, qs, k, M=8, **kw):
    with gpu.DeltaPQIndex.open_memory(payload, n, M, 256, **kw) as idx:
        idx.set_codebook(cb)
        idx.profile_enable(True)
        ids, dists = idx.query_batch(qs, k)
        prof = idx.profile_read()
        info = idx.info()
    return ids, dists, prof, info


@pytest.mark.parametrize("name", ["small_even", "small_odd", "dup_heavy"])
def test_golden_vectors(gpu, name):
    """Committed fixtures (tests/golden/make_golden.py): ids as multisets per tie group, distance bits exact."""
    from oracle.dtc_oracle import tie_aware_equal
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n, k = int(g["n_codes"]), int(g["top_k"])
    ids, dists, _, _ = run(gpu, g["payload"], n, g["codebook"], g["queries"], k)
    for i in range(len(g["queries"])):
        ok, msg = tie_aware_equal(ids[i], dists[i], g["ids"][i], g["dist_bits"][i].view(np.float32))
        assert ok, msg


# (n_codes, n_queries, top_k, chunks_per_segment)
SHAPES = [
    (1, 3, 1, 4), (2, 3, 2, 4), (3, 2, 3, 1), (63, 4, 10, 1), (64, 4, 64, 1), (65, 5, 10, 1),
    (1000, 20, 10, 4), (10000, 100, 10, 4),          # BASELINE configs[0] shape: siftsmall, 100 queries, topk 10
    (9999, 33, 100, 2), (10000, 17, 1000, 4), (4097, 5, 2048, 4), (100000, 64, 100, 4), (300001, 40, 100, 16),
]


@pytest.mark.parametrize("n,nq,k,cps", SHAPES)
def test_parity_with_oracle(gpu, oracle, codebook, n, nq, k, cps):
    from deltapq_amd import synth
    tree, payload, nb = make_case(n, seed=n + 7)
    qs = synth.make_queries(nq, 128, seed=n + 8)
    ids, dists, prof, info = run(gpu, payload, n, codebook, qs, k, chunks_per_segment=cps)
    assert info["algorithmic_bytes"] == nb and info["node_hi"] == n
    assert_parity(ids, dists, oracle_topk(oracle, payload, n, codebook, qs, k), n)
    if n % 2 == 0 and k == n:
        assert n in ids[0] and (n - 1) not in ids[0]              # even-N quirk (h:2949, 2970)


def test_duplicate_heavy_ties(gpu, oracle, codebook):
    """Zero-diff children are exact duplicates: equal-distance groups everywhere, also at the k-th boundary."""
    from deltapq_amd import synth
    n = 50000
    tree, payload, _ = make_case(n, seed=5, dup_heavy=True)
    codes = synth.decode_tree_codes(tree)
    assert len(np.unique(codes, axis=0)) < 0.8 * n
    qs = synth.make_queries(24, 128, seed=6)
    for k in (1, 10, 100):
        ids, dists, _, _ = run(gpu, payload, n, codebook, qs, k)
        assert_parity(ids, dists, oracle_topk(oracle, payload, n, codebook, qs, k), n)
        # canonical tie order of this implementation: ascending id inside equal distances
        for r in range(len(qs)):
            key = dists[r].view(np.uint32).astype(np.uint64) << np.uint64(32) | ids[r].astype(np.uint64)
            assert np.all(np.diff(key.astype(np.float64)) >= 0) and len(set(ids[r].tolist())) == k


def test_query_equal_to_a_centroid_gives_zero_entries(gpu, oracle, codebook):
    """LUT arithmetic at the exact-zero corner and with fractional codebooks (a3)."""
    rng = np.random.default_rng(3)
    cb = rng.normal(40, 25, size=(8, 256, 16)).astype(np.float32)
    n = 4000
    tree, payload, _ = make_case(n, seed=12)
    qs = np.stack([cb[np.arange(8), rng.integers(0, 256, 8)].reshape(-1) for _ in range(6)]).astype(np.float32)
    qs[3:] += rng.normal(0, 1e-3, size=qs[3:].shape).astype(np.float32)
    ids, dists, _, _ = run(gpu, payload, n, cb, qs, 20)
    assert_parity(ids, dists, oracle_topk(oracle, payload, n, cb, qs, 20), n)


def test_candidate_overflow_is_recovered(gpu, oracle, codebook):
    """A tiny candidate buffer forces the overflow rerun path; results must not change."""
    from deltapq_amd import synth
    n = 60000
    tree, payload, _ = make_case(n, seed=21)
    qs = synth.make_queries(40, 128, seed=22)
    ids, dists, prof, _ = run(gpu, payload, n, codebook, qs, 50, cand_capacity=64)
    assert prof["overflow_reruns"] > 0
    assert_parity(ids, dists, oracle_topk(oracle, payload, n, codebook, qs, 50), n)


def test_profile_counts_filter_survivors_and_candidates(gpu, oracle, codebook):
    """dpq_profile.exact_checks / .candidates: what the 8-bit lower-bound filter lets through is checked
    exactly in the scan; every final result must have been a candidate, and the filter may only err on the
    side of letting too much through."""
    from deltapq_amd import synth
    n, nq, k = 200000, 100, 20
    tree, payload, _ = make_case(n, seed=41)
    qs = synth.make_queries(nq, 128, seed=42)
    ids, dists, prof, info = run(gpu, payload, n, codebook, qs, k)
    assert_parity(ids, dists, oracle_topk(oracle, payload, n, codebook, qs, k), n)
    assert prof["scan_launches"] >= 1 and prof["overflow_reruns"] == 0
    assert prof["exact_checks"] >= prof["candidates"] > 0
    assert prof["exact_checks"] < 0.05 * prof["scan_node_query_pairs"]      # the filter does filter


def test_pipelined_batches_match_synchronous_ones(gpu, oracle, codebook):
    """dpq_query_batch_device_async + dpq_finish: several batches in flight on one index, with the
    default buffers and with tiny ones (every batch overflows and is answered again at dpq_finish)."""
    import torch
    from deltapq_amd import synth
    n, k = 60000, 50
    tree, payload, _ = make_case(n, seed=51)
    batches = [synth.make_queries(nq, 128, seed=60 + i) for i, nq in enumerate((40, 7, 130))]
    for kw in ({}, {"cand_capacity": 64}):
        with gpu.DeltaPQIndex.open_memory(payload, n, 8, 256, **kw) as idx:
            idx.set_codebook(codebook)
            idx.profile_enable(True)
            qd = [torch.from_numpy(q).cuda() for q in batches]
            outs = [idx.query_batch_torch(q, k, wait=False) for q in qd]
            idx.finish()
            reruns = idx.profile_read()["overflow_reruns"]
            assert (reruns > 0) == bool(kw)
            for q, (ids, dists) in zip(batches, outs):
                assert_parity(ids.cpu().numpy(), dists.cpu().numpy(), oracle_topk(oracle, payload, n, codebook, q, k), n)
            # a synchronous call after pending work, and an empty finish, are fine too
            idx.query_batch_torch(qd[0], k, wait=False)
            ids2, dists2 = idx.query_batch_torch(qd[1], k)
            assert_parity(ids2.cpu().numpy(), dists2.cpu().numpy(), oracle_topk(oracle, payload, n, codebook, batches[1], k), n)
            idx.finish()


def test_pipelined_batches_on_two_lanes_and_streams(gpu, oracle, codebook):
    """Pipelined batches alternate between two lanes (workspaces + internal streams): many batches
    in flight on an index with the bootstrap and the fused first-level tables, a batch larger than the internal
    2048-query split, a second user stream in the middle (settles what is in flight first), ragged sizes so that
    every batch has padding slots in its last query group."""
    import torch
    from deltapq_amd import synth
    n, k = 300000, 20
    tree, payload, _ = make_case(n, seed=151)
    sizes = (70, 1, 64, 2500, 33, 129, 5)
    batches = [synth.make_queries(nq, 128, seed=160 + i) for i, nq in enumerate(sizes)]
    with gpu.DeltaPQIndex.open_memory(payload, n, 8, 256) as idx:
        idx.set_codebook(codebook)
        assert idx.info()["bootstrap_bytes"] > 0
        qd = [torch.from_numpy(q).cuda() for q in batches]
        other = torch.cuda.Stream()
        outs = []
        for i, q in enumerate(qd):
            if i == 4:
                other.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(other):
                    outs.append(idx.query_batch_torch(q, k, wait=False))
            else:
                outs.append(idx.query_batch_torch(q, k, wait=False))
        idx.finish()
        torch.cuda.synchronize()
        for q, (ids, dists) in zip(batches, outs):
            rows = list(range(0, len(q), max(1, len(q) // 6)))
            assert_parity(ids.cpu().numpy()[rows], dists.cpu().numpy()[rows], oracle_topk(oracle, payload, n, codebook, q[rows], k), n)
            assert np.all(np.diff(dists.cpu().numpy(), axis=1) >= 0) and ids.min().item() >= 0


def test_host_to_host_batches_in_flight(gpu, oracle, codebook):
    """dpq_query_batch_host_async (the reference's interface: host query vectors in, host result lists out, h:2805-2810;
    its per-query loop main:328-339 as batches in flight): seven batches of different sizes enqueued back to back from page-locked memory (result lists written by the select
    kernel into the mapped buffers) and from pageable memory (staging copies), then twenty more than the sixteen slots, on
    both decode placements; a batch that overflows its candidate buffers is answered again by finish() and goes down a
    second time.  Every list equals the synchronous call's and the oracle's."""
    from deltapq_amd import synth
    n, k = 120_000, 40
    tree, payload, _ = make_case(n, seed=411)
    qs = synth.make_queries(1500, 128, seed=412)
    cuts = [(0, 300), (300, 301), (301, 700), (700, 704), (704, 1100), (1100, 1163), (1163, 1500)]
    for pinned, kw in ((True, {}), (False, {}), (True, dict(cand_capacity=64))):
        with gpu.DeltaPQIndex.open_memory(payload, n, 8, 256, **kw) as idx:
            idx.set_codebook(codebook)
            want_i, want_d = idx.query_batch(qs, k)
            q_in = [np.ascontiguousarray(qs[lo:hi]) for lo, hi in cuts]
            out_i = [np.full((hi - lo, k), -7, dtype=np.int32) for lo, hi in cuts]
            out_d = [np.full((hi - lo, k), -7.0, dtype=np.float32) for lo, hi in cuts]
            if pinned:
                for a in q_in + out_i + out_d:
                    gpu.pin_host(a)
            for q, oi, od in zip(q_in, out_i, out_d):
                idx.query_batch_host_async(q, k, oi, od)
            reruns = idx.finish()
            if not kw:   # more calls than slots: the seventeenth settles the earlier ones
                many = [(np.ascontiguousarray(qs[j:j + 3]), np.empty((3, k), np.int32), np.empty((3, k), np.float32)) for j in range(0, 60, 3)]
                for q, oi, od in many:
                    idx.query_batch_host_async(q, k, oi, od)
                idx.finish()
                for j, (q, oi, od) in enumerate(many):
                    assert np.array_equal(oi, want_i[3 * j:3 * j + 3]) and np.array_equal(od.view(np.uint32), want_d[3 * j:3 * j + 3].view(np.uint32)), j
            if pinned:
                for a in q_in + out_i + out_d:
                    gpu.unpin_host(a)
        assert (reruns > 0) == ("cand_capacity" in kw), (kw, reruns)
        got_i, got_d = np.concatenate(out_i), np.concatenate(out_d)
        assert np.array_equal(got_i, want_i) and np.array_equal(got_d.view(np.uint32), want_d.view(np.uint32)), (pinned, kw)
    pick = [0, 300, 702, 1499]
    assert_parity(want_i[pick], want_d[pick], oracle_topk(oracle, payload, n, codebook, qs[pick], k), n)


def test_stream_ordered_batches_feed_device_consumers(gpu, oracle, codebook):
    """dpq_query_batch_device_ordered: the result is consumed on the device in stream order (here: copied by a torch
    op enqueued right behind it, as the sharded driver's pack + all-gather are), no host round trip; finish() reports
    how many batches had to be answered again (none with the default buffers, every one with tiny buffers)."""
    import torch
    from deltapq_amd import synth
    n, k = 200000, 30
    tree, payload, _ = make_case(n, seed=171)
    batches = [synth.make_queries(nq, 128, seed=180 + i) for i, nq in enumerate((90, 64, 200))]
    for kw, expect_reruns in (({}, False), ({"cand_capacity": 64, "bootstrap": -1}, True)):
        with gpu.DeltaPQIndex.open_memory(payload, n, 8, 256, **kw) as idx:
            idx.set_codebook(codebook)
            ids = torch.empty((200, k), dtype=torch.int32, device="cuda")
            dists = torch.empty((200, k), dtype=torch.float32, device="cuda")
            copies = []
            for q in batches:
                qd = torch.from_numpy(q).cuda()
                idx.query_batch_torch(qd, k, ids[:len(q)], dists[:len(q)], wait=False, ordered=True)
                copies.append((qd, ids[:len(q)].clone(), dists[:len(q)].clone()))     # consumers in stream order; buffers reused
            reruns = idx.finish()
            torch.cuda.synchronize()
            assert (reruns > 0) == expect_reruns
            if not expect_reruns:
                for q, (_, ci, cd) in zip(batches, copies):
                    rows = list(range(0, len(q), 17))
                    assert_parity(ci.cpu().numpy()[rows], cd.cpu().numpy()[rows], oracle_topk(oracle, payload, n, codebook, q[rows], k), n)


def test_stream_ordered_batches_on_two_alternating_streams(gpu, codebook):
    """Stream-ordered batches of two caller streams in flight together: each stream is given one of the two
    workspaces, no host round trip when the caller switches streams (the sharded driver alternates its steps like
    this).  Eight batches, consumed by a copy enqueued behind each on its own stream, against synchronous calls; then
    a third stream and a laned batch settle what is in flight and still answer right."""
    import torch
    from deltapq_amd import synth
    n, k = 150000, 25
    tree, payload, _ = make_case(n, seed=173)
    qs = [torch.from_numpy(synth.make_queries(200, 128, seed=190 + i)).cuda() for i in range(8)]
    with gpu.DeltaPQIndex.open_memory(payload, n, 8, 256) as idx:
        idx.set_codebook(codebook)
        want = [tuple(t.clone() for t in idx.query_batch_torch(q, k)) for q in qs]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        out = [(torch.empty((200, k), dtype=torch.int32, device="cuda"), torch.empty((200, k), dtype=torch.float32, device="cuda"))
               for _ in range(2)]
        got = []
        for i, q in enumerate(qs):
            with torch.cuda.stream(streams[i & 1]):
                o_i, o_d = out[i & 1]
                idx.query_batch_torch(q, k, o_i, o_d, wait=False, ordered=True)
                got.append((o_i.clone(), o_d.clone()))      # consumed in stream order; the buffer is reused two steps later
        assert idx.finish() == 0
        torch.cuda.synchronize()
        for (wi, wd), (gi, gd) in zip(want, got):
            assert torch.equal(wi, gi) and torch.equal(wd.view(torch.int32), gd.view(torch.int32))
        third = torch.cuda.Stream()
        third.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(streams[0]):
            idx.query_batch_torch(qs[0], k, out[0][0], out[0][1], wait=False, ordered=True)
        with torch.cuda.stream(streams[1]):
            idx.query_batch_torch(qs[1], k, out[1][0], out[1][1], wait=False, ordered=True)
        with torch.cuda.stream(third):
            t_i, t_d = idx.query_batch_torch(qs[2], k, wait=False, ordered=True)    # settles the two above first
        l_i, l_d = idx.query_batch_torch(qs[3], k, wait=False)                          # laned: settles again
        assert idx.finish() == 0
        torch.cuda.synchronize()
        for j, (gi, gd) in enumerate((out[0], out[1], (t_i, t_d), (l_i, l_d))):
            assert torch.equal(want[j][0], gi) and torch.equal(want[j][1].view(torch.int32), gd.view(torch.int32))


def test_large_batch_is_split_internally(gpu, oracle, codebook):
    from deltapq_amd import synth
    n = 3000
    tree, payload, _ = make_case(n, seed=31)
    qs = synth.make_queries(2500, 128, seed=32)                  # > 2048: two internal sub-batches
    ids, dists, _, _ = run(gpu, payload, n, codebook, qs, 5)
    sel = [0, 1, 2047, 2048, 2049, 2499]
    ref = oracle_topk(oracle, payload, n, codebook, qs[sel], 5)
    assert_parity(ids[sel], dists[sel], ref, n)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_index_merges_to_the_same_answer(gpu, oracle, codebook, world):
    """8e: shards are independent (own checkpoints), report global DFS positions, host merge."""
    from deltapq_amd import synth
    n, nq, k = 40000, 12, 100
    tree, payload, _ = make_case(n, seed=41)
    qs = synth.make_queries(nq, 128, seed=42)
    pi, pd, covered = [], [], 0
    for r in range(world):
        ids, dists, _, info = run(gpu, payload, n, codebook, qs, k, shard_rank=r, shard_count=world)
        ok = ids >= 0
        rep_lo, rep_hi = info["node_lo"], info["node_hi"] + (1 if info["node_hi"] == n and n % 2 == 0 else 0)
        assert np.all((ids[ok] >= rep_lo) & (ids[ok] < rep_hi))
        covered += info["node_hi"] - info["node_lo"]
        pi.append(ids)
        pd.append(dists)
    assert covered == n
    mi, md = gpu.merge_topk_host(np.stack(pi), np.stack(pd))
    assert_parity(mi, md, oracle_topk(oracle, payload, n, codebook, qs, k), n)


def test_shard_smaller_than_topk_pads(gpu, oracle, codebook):
    from deltapq_amd import synth
    n, k = 300, 200                                               # 2 segments, 4 shards: two are empty
    tree, payload, _ = make_case(n, seed=51)
    qs = synth.make_queries(3, 128, seed=52)
    pi, pd = [], []
    for r in range(4):
        ids, dists, _, info = run(gpu, payload, n, codebook, qs, k, shard_rank=r, shard_count=4)
        held = info["node_hi"] - info["node_lo"]
        assert np.all((ids >= 0).sum(axis=1) == min(k, held))
        assert np.all(np.isinf(dists[ids < 0]))
        pi.append(ids)
        pd.append(dists)
    mi, md = gpu.merge_topk_host(np.stack(pi), np.stack(pd))
    assert_parity(mi, md, oracle_topk(oracle, payload, n, codebook, qs, k), n)


def test_device_pointer_entry_and_device_merge(gpu, oracle, codebook):
    import torch
    from deltapq_amd import synth
    n, nq, k = 30000, 20, 64
    tree, payload, _ = make_case(n, seed=61)
    qs = synth.make_queries(nq, 128, seed=62)
    qd = torch.from_numpy(qs).cuda()
    torch.cuda.synchronize()
    parts_i, parts_d = [], []
    side = torch.cuda.Stream()                                     # a non-default stream of torch's runtime:
    with torch.cuda.stream(side):                                  # library and torch must share ONE HIP runtime
        for r in range(2):
            with gpu.DeltaPQIndex.open_memory(payload, n, 8, 256, shard_rank=r, shard_count=2) as idx:
                idx.set_codebook(codebook)
                i, d = idx.query_batch_torch(qd, k)
                parts_i.append(i.clone())
                parts_d.append(d.clone())
        mi, md = gpu.merge_topk_torch(torch.stack(parts_i).contiguous(), torch.stack(parts_d).contiguous())
    side.synchronize()
    assert_parity(mi.cpu().numpy(), md.cpu().numpy(), oracle_topk(oracle, payload, n, codebook, qs, k), n)


def test_error_codes(gpu, codebook):
    from deltapq_amd import synth
    tree, payload, _ = make_case(500, seed=71)
    qs = synth.make_queries(2, 128, seed=72)
    with gpu.DeltaPQIndex.open_memory(payload, 500, 8, 256) as idx:
        with pytest.raises(gpu.DpqError) as e:
            idx.query_batch(qs, 5)
        assert e.value.status == -7                                # DPQ_ERR_STATE: no codebook yet
        idx.set_codebook(codebook)
        with pytest.raises(gpu.DpqError) as e:
            idx.query_batch(qs, 501)
        assert e.value.status == -8                                # DPQ_ERR_TOPK (reference: empty-heap pop)
        with pytest.raises(gpu.DpqError) as e:
            idx.query_batch(qs, 0)
        assert e.value.status == -1
        ids, _ = idx.query_batch(qs, 500)
        assert sorted(ids[0].tolist()) == list(range(499)) + [500]   # even N
    with pytest.raises(gpu.DpqError) as e:
        gpu.DeltaPQIndex.open_memory(payload[:-1], 500, 8, 256)
    assert e.value.status == -3
    # a part of a larger index must say how large the whole is (else the even-N rule would hit the wrong node), and its
    # global positions must fit the int32 ids of the result
    for kw in (dict(global_offset=1000), dict(global_offset=2**31 - 400, global_n_codes=2**31 - 2),
               dict(global_offset=10, global_n_codes=505), dict(global_offset=-1, global_n_codes=1000)):
        with pytest.raises(gpu.DpqError) as e:
            gpu.DeltaPQIndex.open_memory(payload, 500, 8, 256, **kw)
        assert e.value.status == -1, kw
    with pytest.raises(TypeError):
        gpu.DeltaPQIndex.open_memory(payload, 500, 8, 256, no_such_knob=1)


def test_reference_named_entry_points(gpu, oracle, codebook, tmp_path):
    """One call per query with the reference's argument lists (h:2805-2810, 3731-3736)."""
    from deltapq_amd import synth
    d = str(tmp_path)
    tree, cb, queries = synth.make_dataset_dir(d, 2001, 3, seed=81)
    n_codes, payload = gpu.read_dtc_file(synth.dtc_file_name(d, 8, 256, 2001))
    for q in queries:
        ref = oracle_topk(oracle, payload, 2001, cb, [q], 10)
        a = gpu.query_processing_scan_compressed_codes_opt_in_memory(payload, len(payload), q, 10, 8, 256, 16, 2001, cb)
        b = gpu.query_processing_scan_compressed_codes_opt_o_direct(d, q, 10, 8, 256, 16, 2001, cb)
        for res in (a, b):
            ids = np.array([[r[0] for r in res]], np.int32)
            dd = np.array([[r[1] for r in res]], np.float32)
            assert_parity(ids, dd, ref, 2001)


def test_cli_query_matches_oracle(gpu, oracle, tmp_path):
    """`deltapq -task query` with the reference's flags on a reference-style dataset directory."""
    from deltapq_amd import synth
    d = str(tmp_path)
    n, nq, k = 10000, 100, 10                                     # BASELINE configs[0]
    tree, cb, queries = synth.make_dataset_dir(d, n, nq, seed=91)
    n_codes, payload = gpu.read_dtc_file(synth.dtc_file_name(d, 8, 256, n))
    exe = os.path.join(ROOT, "deltapq_amd", "csrc", "deltapq")
    out = os.path.join(d, "results.bin")
    for task in ("query", "query_im"):
        r = subprocess.run([exe, "-dataset", d, "-task", task, "-m", "8", "-k", "256", "-h", "1", "-diff", "8",
                            "-N", str(n), "-query_size", "50", "-topk", str(k), "-debug", "-out", out],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "[msec/query]" in r.stdout and "50 queries run" in r.stdout
        raw = np.fromfile(out, dtype=np.uint8)
        nq_out, k_out = np.frombuffer(raw[:16], np.int64)
        assert (nq_out, k_out) == (50, k)
        ids = np.frombuffer(raw[16:16 + 50 * k * 4], np.int32).reshape(50, k)
        dists = np.frombuffer(raw[16 + 50 * k * 4:], np.float32).reshape(50, k)
        assert_parity(ids, dists, oracle_topk(oracle, payload, n, cb, queries[:50], k), n)
        # -debug prints "<top1 id> <top1 dist>" per query (main:340-343)
        top1 = [l.split() for l in r.stdout.splitlines() if len(l.split()) == 2 and l.split()[0].lstrip("-").isdigit()]
        assert [int(t[0]) for t in top1[-50:]] == ids[:, 0].tolist()


def test_full_size_sift1m_shape(gpu, oracle):
    """BASELINE configs[1] at full size, on the index the bench measures: 1 M SIFT-shaped vectors -> k-means codebook -> GPU
    PQ encode -> GPU DeltaTree build -> DTC, one batch of 1000 queries, top-100.  Full oracle parity on 32 sampled queries,
    size-independent properties on all 1000: ascending distances, k distinct ids in range, every distance = the fp64 sum of
    the query's table entries over the decoded code of its id (rounded once), every node filtered exactly once per query."""
    from deltapq_amd import synth
    n, nq, k, M = 1_000_000, 1000, 100, 8
    base = synth.make_clustered_vectors(n, 128, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
    cb = synth.kmeans_codebook(base, M, 256, iters=6, seed=102)
    codes = gpu.encode_pq(base, cb)
    del base
    tree = gpu.DeltaTree(codes, codebook=cb, device=0)
    payload = tree.payload()
    tree.close()
    nb = len(payload)
    qs = synth.make_clustered_vectors(nq, 128, seed=101, n_clusters=20000, spread=12.0, centre_seed=7)
    ids, dists, prof, info = run(gpu, payload, n, cb, qs, k)
    assert info["algorithmic_bytes"] == nb and 3.5 < nb / n < 5.5                 # ~4.3 B per code
    sample = list(range(0, nq, 32))[:32]
    assert_parity(ids[sample], dists[sample], oracle_topk(oracle, payload, n, cb, qs[sample], k), n)
    lut0 = oracle.build_lut(cb, qs[1])
    _, _, _, dcodes = oracle.scan_lut(payload, n, lut0, 1, want_all=True)
    assert np.all(np.diff(dists, axis=1) >= 0)
    for r in range(nq):
        assert len(set(ids[r].tolist())) == k and ids[r].min() >= 0 and ids[r].max() <= n
        lut = oracle.build_lut(cb, qs[r])
        pos = np.where(ids[r] == n, n - 1, ids[r])
        s = sum(lut[m, dcodes[pos, m]].astype(np.float64) for m in range(M)).astype(np.float32)
        assert np.array_equal(s.view(np.uint32), dists[r].view(np.uint32))
    # every node scanned exactly once per query by the plan's levels
    assert prof["scan_node_query_pairs"] == info["n_segments"] * 64 * info["chunks_per_segment"] * nq


def test_large_shard_config3_per_gpu_share(gpu, oracle, codebook):
    """BASELINE configs[3]: 100M codes over 8 GPUs = 12.5M codes per GPU.  One such
    shard-sized index on one GPU: oracle parity on a query sample + properties."""
    from deltapq_amd import synth
    n, nq, k = 12_500_000, 64, 100
    tree = synth.synth_tree_large(n, 8, seed=7, mean_diffs=3.0)
    payload, nb = synth.encode_dtc(tree)
    del tree
    qs = synth.make_queries(nq, 128, seed=8)
    ids, dists, prof, info = run(gpu, payload, n, codebook, qs, k)
    S = 64 * info["chunks_per_segment"]
    assert info["algorithmic_bytes"] == nb and info["n_segments"] == (n + S - 1) // S
    sample = [0, 21, 42, 63]
    assert_parity(ids[sample], dists[sample], oracle_topk(oracle, payload, n, codebook, qs[sample], k), n)
    assert np.all(np.diff(dists, axis=1) >= 0)
    for r in range(nq):
        assert len(set(ids[r].tolist())) == k and ids[r].min() >= 0 and ids[r].max() <= n
    assert prof["scan_node_query_pairs"] == info["n_segments"] * S * nq
    # calls of one, two and four queries on this shard size take the stream pass as the library routes it by default: the
    # small first level through the chunk-per-wavefront pass over its strips' segments, the big one through the strand
    # pass -- same lists as inside the 64-query batch, bit for bit
    with gpu.DeltaPQIndex.open_memory(payload, n, 8, 256) as idx:
        idx.set_codebook(codebook)
        for lo_q, hi_q in ((0, 1), (20, 22), (40, 44)):
            ids_s, d_s = idx.query_batch(qs[lo_q:hi_q], k)
            assert np.array_equal(ids_s, ids[lo_q:hi_q]) and np.array_equal(d_s.view(np.uint32), dists[lo_q:hi_q].view(np.uint32)), lo_q
    # the same index as shard 5 of 8 (what one rank of the 8-GPU run holds)
    ids5, dists5, _, info5 = run(gpu, payload, n, codebook, qs[:8], k, shard_rank=5, shard_count=8)
    lut = oracle.build_lut(codebook, qs[3])
    _, _, alld, _ = oracle.scan_lut(payload, n, lut, 1, want_all=True)
    lo, hi = info5["node_lo"], info5["node_hi"]
    assert abs((hi - lo) - n / 8) < 0.05 * n / 8
    order = np.lexsort((np.arange(lo, hi), alld[lo:hi].view(np.uint32)))[:k] + lo
    assert np.array_equal(dists5[3].view(np.uint32), alld[order].view(np.uint32))
    assert set(ids5[3].tolist()) == set(order.tolist()) or np.array_equal(np.sort(alld[ids5[3]]), np.sort(alld[order]))


def test_two_process_sharded_run_on_one_gpu(gpu, built):
    """8e end to end with real processes: 2 ranks (both on GPU 0, gloo for the
    exchange because RCCL wants one device per rank) shard the index, query on
    the GPU through the C-ABI, all-gather the partial lists and merge."""
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_dist_gpu_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "DIST_GPU_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def _run_bench(extra, nproc=2, timeout=900, env=None):
    import json
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    # exactly the driver's command line for N > 1 (torch.distributed.run, one rank per GPU), plus the test's sizes
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nproc)] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, OMP_NUM_THREADS="1", **(env or {})), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("mode", ["index_stream_parts", "index_pipeline", "query"])
def test_bench_two_ranks_rehearsal(gpu, built, mode):
    """bench.py's N > 1 paths end to end, 2 ranks on GPU 0 (gloo instead of RCCL, ranks share the GPU):
    index shards where every rank synthesises only its own DFS range (configs[3]/[4] scaled down; the default
    decomposition, strong scaling), index shards of a pipeline-built index, and query replicas (--shard query)."""
    common = ["--backend", "gloo", "--queries", "100", "--steps", "3", "--warmup", "1", "--reps", "2", "--check", "3",
              "--no-cpu-baseline"]
    if mode == "index_stream_parts":
        line = _run_bench(common + ["--data", "stream", "--codes", "700001"])
        assert "own DFS range" in line["config"]["workload"] and line["index"]["codes_rank0"] == 350000
    elif mode == "index_pipeline":
        line = _run_bench(common + ["--data", "pipeline", "--codes", "120000"])
        assert line["query_replicas"]["value"] > 0 and line["query_replicas"]["scaling"] == "weak"
    else:
        line = _run_bench(common + ["--data", "stream", "--codes", "200000", "--shard", "query"])
    assert line["n_gpus"] == 2 and line["parity_checked_queries"] == 3 and line["value"] > 0
    assert 0 < line["roofline"]["frac"] <= 1 and line["roofline"]["bound"] == "lds" and line["roofline"]["launches_per_step"] >= 1
    assert line["repetitions"]["count"] == 2 and line["repetitions"]["ms_per_step_min"] <= line["ms_per_step"] <= line["repetitions"]["ms_per_step_max"]
    if mode == "query":
        assert line["scaling"] == "weak" and line["config"]["global_queries_per_step"] == 200
    else:
        assert line["scaling"] == "strong" and line["config"]["global_queries_per_step"] == 100
        assert "index shards x2" in line["config"]["sharding"]


@pytest.mark.parametrize("mode", ["index_pipeline", "index_stream_parts", "query"])
def test_bench_two_gpus_over_rccl(gpu, built, mode):
    """The RCCL branch of the exchange (all_gather_into_tensor on device tensors + device merge): only where two
    GPUs are visible (the driver's multi-GPU node); the 1-GPU box skips it.  Three shapes: a pipeline-built index cut
    into DFS-range shards, every rank synthesising only its own part of a larger index (BASELINE configs[3]/[4]:
    --data stream), and query replicas (no collective on the data path)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 visible GPUs (RCCL wants one device per rank)")
    args = {"index_pipeline": ["--data", "pipeline", "--codes", "300000"],
            "index_stream_parts": ["--data", "stream", "--codes", "600000"],
            "query": ["--data", "pipeline", "--codes", "300000", "--shard", "query"]}[mode]
    line = _run_bench(["--backend", "nccl"] + args + ["--queries", "200", "--steps", "3", "--warmup", "1", "--reps", "2",
                                                      "--check", "8", "--no-cpu-baseline"])
    assert line["n_gpus"] == 2 and line["parity_checked_queries"] == 8 and line["value"] > 0
    assert line["scaling"] == ("weak" if mode == "query" else "strong")
    if mode != "query":
        assert line["config"]["collectives"].startswith("RCCL (backend nccl)"), line["config"]["collectives"]


def test_cli_two_gpus_shards_and_merges(gpu, oracle, tmp_path):
    """`deltapq -task query -gpus 2`: one process, a host thread per device, each device a DFS-range shard, partial
    lists merged on the host (dpq_merge_topk_host) -- the north star's host-merged decomposition.  Needs two visible
    devices; with one the tool must refuse and say why."""
    import torch
    from deltapq_amd import synth
    d = str(tmp_path)
    n, k = 60_000, 20
    tree, cb, queries = synth.make_dataset_dir(d, n, 40, seed=77)
    n_codes, payload = gpu.read_dtc_file(synth.dtc_file_name(d, 8, 256, n))
    exe = os.path.join(ROOT, "deltapq_amd", "csrc", "deltapq")
    out = os.path.join(d, "results2.bin")
    r = subprocess.run([exe, "-dataset", d, "-task", "query", "-m", "8", "-k", "256", "-h", "1", "-diff", "8", "-N", str(n),
                        "-query_size", "40", "-topk", str(k), "-gpus", "2", "-out", out], capture_output=True, text=True, timeout=300)
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "only 1 device(s) visible" in r.stdout, r.stdout + r.stderr
        pytest.skip("needs 2 visible GPUs for the sharded run itself")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "index resident on 2 GPU(s)" in r.stdout
    raw = np.fromfile(out, dtype=np.uint8)
    ids = np.frombuffer(raw[16:16 + 40 * k * 4], np.int32).reshape(40, k)
    dists = np.frombuffer(raw[16 + 40 * k * 4:], np.float32).reshape(40, k)
    assert_parity(ids, dists, oracle_topk(oracle, payload, n, cb, queries[:40], k), n)


def test_bench_survives_an_unusable_rccl(gpu, built):
    """The driver's N > 1 command line with the default backend where RCCL cannot work (two ranks on ONE device:
    ncclInvalidUsage): bench.py proves the collective layer with one all-reduce, falls back to gloo with host staging
    and says so on its line.  (With two devices visible RCCL works and the line says that.)"""
    line = _run_bench(["--data", "pipeline", "--codes", "150000", "--queries", "100", "--steps", "3", "--warmup", "1", "--reps", "2",
                       "--check", "4", "--no-cpu-baseline", "--no-replicas"], timeout=600, env={"DPQ_BENCH_SHARE_GPUS": "1"})
    import torch
    note = line["config"]["collectives"]
    assert note.startswith("RCCL unusable" if torch.cuda.device_count() < 2 else "RCCL (backend nccl)"), note
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["parity_checked_queries"] == 4 and line["value"] > 0


@pytest.mark.parametrize("n,cps", [(200_000, 0), (70_001, 4), (9_000, 1)])
def test_batch_decode_modes_agree(gpu, oracle, codebook, n, cps):
    """dpq_open_opts.batch_decode: decoding the shard once per batch into the plain-code scratch (decode_list_kernel +
    the scan's plain-code instantiation with the DTC distance rule) and decoding inside the scan give identical lists;
    automatic = scratch from three query groups on.  Shapes: bootstrap shard, odd N with 4-chunk segments, small
    cascade shard (level 0 included)."""
    from deltapq_amd import synth
    k = 40
    tree, payload, _ = make_case(n, seed=41 + cps)
    qs = synth.make_queries(200, 128, seed=42)
    ids_f, d_f, prof_f, info_f = run(gpu, payload, n, codebook, qs, k, chunks_per_segment=cps, batch_decode=-1)
    ids_b, d_b, prof_b, info_b = run(gpu, payload, n, codebook, qs, k, chunks_per_segment=cps, batch_decode=1)
    assert np.array_equal(ids_f, ids_b) and np.array_equal(d_f.view(np.uint32), d_b.view(np.uint32))
    assert prof_f["decode_ms"] == 0.0 and prof_b["decode_ms"] > 0.0
    assert info_f["batch_decode_mb"] == 0 and info_b["batch_decode_mb"] == -(-info_b["n_segments"] * 64 * info_b["chunks_per_segment"] * 8 // (1 << 20))
    ids_a, d_a, prof_a, _ = run(gpu, payload, n, codebook, qs, k, chunks_per_segment=cps)              # 200 queries: 4 groups
    assert prof_a["decode_ms"] > 0.0 and np.array_equal(ids_a, ids_b) and np.array_equal(d_a.view(np.uint32), d_b.view(np.uint32))
    ids_s, d_s, prof_s, _ = run(gpu, payload, n, codebook, qs[:128], k, chunks_per_segment=cps)        # 2 groups: inside the scan
    assert prof_s["decode_ms"] == 0.0 and np.array_equal(ids_s, ids_b[:128]) and np.array_equal(d_s.view(np.uint32), d_b[:128].view(np.uint32))
    assert_parity(ids_b[:12], d_b[:12], oracle_topk(oracle, payload, n, codebook, qs[:12], k), n)


@pytest.mark.parametrize("n,tile,k", [(300_000, 37, 60), (2_600_000, 3000, 100)])
def test_batch_decode_in_tiles(gpu, oracle, codebook, n, tile, k):
    """The scratch one TILE of a level's segment list at a time (dpq_open_opts.batch_decode = tile segments): the
    scan addresses the scratch by list position and a level's launches append to its candidate regions.  A
    single-level shard in 64 tiles, and a shard beyond 2 M nodes (two filter levels in the golden-ratio visiting
    order, tiles that straddle nothing: each level is tiled on its own)."""
    from deltapq_amd import synth
    tree, payload, _ = make_case(n, seed=91)
    qs = synth.make_queries(200, 128, seed=92)
    ids_f, d_f, prof_f, _ = run(gpu, payload, n, codebook, qs, k, batch_decode=-1)
    ids_t, d_t, prof_t, info_t = run(gpu, payload, n, codebook, qs, k, batch_decode=tile)
    assert np.array_equal(ids_f, ids_t) and np.array_equal(d_f.view(np.uint32), d_t.view(np.uint32))
    assert prof_t["decode_ms"] > 0.0 and prof_t["scan_launches"] >= -(-info_t["n_segments"] // tile)
    assert info_t["batch_decode_mb"] == -(-tile * 64 * info_t["chunks_per_segment"] * 8 // (1 << 20))
    assert prof_t["exact_checks"] == prof_f["exact_checks"] and prof_t["candidates"] == prof_f["candidates"]
    assert_parity(ids_t[:6], d_t[:6], oracle_topk(oracle, payload, n, codebook, qs[:6], k), n)


@pytest.mark.parametrize("n,M,k", [(200_000, 8, 50), (9_000, 8, 10), (2_300_000, 8, 100), (150_000, 16, 30)])
def test_one_query_per_pass_matches_the_batched_path(gpu, oracle, n, M, k):
    """Small batches run stream_kernel (1, 2 or 4 queries per pass over the compressed image, every decoded node against
    the queries' exact tables in LDS); the same queries inside a larger batch run the 64-query filter path.  Same lists,
    bit for bit; and against the oracle.  The index is opened with stream_max_queries = 8 (the default switch-over is
    4) so that multi-pass batches are covered too: 1, 2, 3 (a pass of four with an unused slot), 4, 5 (4 + 1), 8
    (two passes of four) and 9 (back on the filter path).  Shapes: bootstrap shard, small cascade shard, a shard beyond
    2 M nodes (two levels), M = 16; also as shard 1 of 2."""
    from deltapq_amd import synth
    cb = synth.make_codebook(M, 256, 128 // M, seed=3)
    tree = synth.synth_tree(n, M, seed=n + 5, mean_diffs=3.0 if M == 8 else 5.0)
    payload, _ = synth.encode_dtc(tree)
    qs = synth.make_queries(70, 128, seed=n + 6)
    cuts = [(0, 1), (5, 7), (10, 13), (20, 24), (30, 35), (40, 48), (50, 59)]
    for kw in ({}, {"shard_rank": 1, "shard_count": 2}):
        # flags = 64: DPQ_OPT_FORCE_STRANDS, the lane-per-run stream pass (strand_kernel) on shards far below the size from
        # which it is the default (M = 8 shards with a bootstrap have the image; the others run stream_kernel)
        with gpu.DeltaPQIndex.open_memory(payload, n, M, 256, stream_max_queries=8, flags=gpu.OPT_FORCE_STRANDS, **kw) as idx:
            idx.set_codebook(cb)
            ids_b, d_b = idx.query_batch(qs, k)                # 70 queries: filter path
            got = [idx.query_batch(qs[lo:hi], k) for lo, hi in cuts]
        with gpu.DeltaPQIndex.open_memory(payload, n, M, 256, stream_max_queries=8, **kw) as idx:   # stream_kernel at these sizes
            idx.set_codebook(cb)
            for (lo, hi), (ids_s, d_s) in zip(cuts[:4], got[:4]):
                ids_c, d_c = idx.query_batch(qs[lo:hi], k)
                assert np.array_equal(ids_c, ids_s) and np.array_equal(d_c.view(np.uint32), d_s.view(np.uint32)), (lo, hi, kw)
            idx.profile_enable(1)
            idx.profile_reset()
            idx.query_batch(qs[20:24], k)
            prof = idx.profile_read()
        for (lo, hi), (ids_s, d_s) in zip(cuts, got):
            assert np.array_equal(ids_s, ids_b[lo:hi]), (lo, hi, kw)
            assert np.array_equal(d_s.view(np.uint32), d_b[lo:hi].view(np.uint32)), (lo, hi, kw)
        # the stream path visits every (node, query) pair once and checks nothing through the filter
        assert prof["exact_checks"] == 0 and prof["scan_launches"] >= 1
        if not kw:
            pick = [0, 5, 6, 10, 12, 23, 34, 47]
            sel = {q: (i, q - lo) for i, (lo, hi) in enumerate(cuts) for q in range(lo, hi)}
            ids_p = np.stack([got[sel[q][0]][0][sel[q][1]] for q in pick])
            d_p = np.stack([got[sel[q][0]][1][sel[q][1]] for q in pick])
            assert_parity(ids_p, d_p, oracle_topk(oracle, payload, n, cb, qs[pick], k), n)


@pytest.mark.parametrize("n,k,diffs,kw", [
    (20_000, 10, 3.0, dict(bootstrap=1)),                                   # the smallest shard that can carry the strand image
    (200_000, 50, 3.0, {}),
    (200_000, 100, 0.4, {}),                                                # duplicate-heavy: thousands of equal keys at the cut
    (2_300_000, 100, 3.0, {}),
    (200_001, 20, 3.0, dict(num_codes=123_457)),                            # a prefix scan (odd: exactly the reference's `-N`; even prefixes: test_stream_pass_on_a_prefix...)
    (200_000, 20, 3.0, dict(shard_rank=1, shard_count=3)),
    (200_000, 20, 3.0, dict(global_offset=12_345_678, global_n_codes=1_000_000_000)),   # a part of a larger index
])
def test_one_query_strand_pass_with_the_bound_table(gpu, oracle, codebook, n, k, diffs, kw):
    """strand1_kernel (one query per pass over the strand image): per-bank 8-bit bound rows, exact check of
    what they let through, one level with in-kernel threshold tightening.  Every single-query call must return the list
    the same query gets inside a 70-query batch (the filter path), bit for bit; also with the tightening off (the
    multi-level plan) and through the exact-table kernel it replaces (DPQ_OPT_NO_STRAND1); and the oracle's list."""
    from deltapq_amd import synth
    tree = synth.synth_tree(n, 8, seed=n + k, mean_diffs=diffs)
    payload, _ = synth.encode_dtc(tree)
    qs = synth.make_queries(70, 128, seed=n + k + 1)
    F = gpu.OPT_FORCE_STRANDS
    with gpu.DeltaPQIndex.open_memory(payload, n, 8, 256, flags=F, **kw) as idx:
        idx.set_codebook(codebook)
        ids_b, d_b = idx.query_batch(qs, k)
        idx.profile_enable(1)
        idx.profile_reset()
        singles = [idx.query_batch(qs[i:i + 1], k) for i in range(12)]
        prof = idx.profile_read()
    assert prof["strand1_launches"] == 12 and prof["strand_launches"] == 0 and prof["stream_launches"] == 0   # one level each
    assert prof["overflow_reruns"] == 0
    for i, (ids_s, d_s) in enumerate(singles):
        assert np.array_equal(ids_s[0], ids_b[i]) and np.array_equal(d_s[0].view(np.uint32), d_b[i].view(np.uint32)), i
    for flags in (F | gpu.OPT_NO_TIGHTEN, F | gpu.OPT_NO_STRAND1):
        with gpu.DeltaPQIndex.open_memory(payload, n, 8, 256, flags=flags, **kw) as idx:
            idx.set_codebook(codebook)
            for i in (0, 5, 11):
                ids_s, d_s = idx.query_batch(qs[i:i + 1], k)
                assert np.array_equal(ids_s[0], ids_b[i]) and np.array_equal(d_s[0].view(np.uint32), d_b[i].view(np.uint32)), (flags, i)
    if not kw or "bootstrap" in kw:
        pick = [0, 3, 7, 11]
        assert_parity(ids_b[pick], d_b[pick], oracle_topk(oracle, payload, n, codebook, qs[pick], k), n)
    elif "num_codes" in kw:
        n_scan = kw["num_codes"]
        assert_parity(ids_b[:2], d_b[:2], oracle_topk(oracle, payload, n_scan, codebook, qs[:2], k), n_scan)


def test_one_query_strand_pass_tightens_in_the_kernel(gpu, oracle, codebook):
    """One level over the whole shard at the bootstrap's threshold would admit every node below it; the kernel lowers its
    cut from the candidates all wavefronts have found: far fewer candidates than nodes under the first threshold, none
    lost (the list equals the filter path's), no overflow rerun."""
    from deltapq_amd import synth
    n, k = 3_000_000, 100
    tree = synth.synth_tree_large(n, 8, seed=77, mean_diffs=3.0)
    payload, _ = synth.encode_dtc(tree)
    qs = synth.make_queries(70, 128, seed=78)
    res = {}
    for flags in (gpu.OPT_FORCE_STRANDS, gpu.OPT_FORCE_STRANDS | gpu.OPT_NO_TIGHTEN):
        with gpu.DeltaPQIndex.open_memory(payload, n, 8, 256, flags=flags, plan_ratios=[1, 0, 0] if flags & gpu.OPT_NO_TIGHTEN else [0, 0, 0]) as idx:
            idx.set_codebook(codebook)
            ids_b, d_b = idx.query_batch(qs, k)
            idx.profile_enable(1)
            idx.profile_reset()
            got = [idx.query_batch(qs[i:i + 1], k) for i in range(8)]
            res[flags] = (got, idx.profile_read())
            for i, (ids_s, d_s) in enumerate(got):
                assert np.array_equal(ids_s[0], ids_b[i]) and np.array_equal(d_s[0].view(np.uint32), d_b[i].view(np.uint32)), (flags, i)
    on, off = res[gpu.OPT_FORCE_STRANDS][1], res[gpu.OPT_FORCE_STRANDS | gpu.OPT_NO_TIGHTEN][1]
    assert on["strand1_launches"] == 8 and off["strand1_launches"] == 8          # one level either way (forced for `off`)
    assert on["overflow_reruns"] == 0
    assert on["candidates"] < 0.5 * off["candidates"], (on["candidates"], off["candidates"])
    assert_parity(ids_b[:3], d_b[:3], oracle_topk(oracle, payload, n, codebook, qs[:3], k), n)


@pytest.mark.parametrize("n,M,k,dup", [(300_000, 8, 100, False), (300_000, 8, 10, True), (150_000, 16, 50, False)])
def test_in_scan_tightening_changes_nothing_but_the_work(gpu, oracle, n, M, k, dup):
    """dpq_open_opts.flags & DPQ_OPT_NO_TIGHTEN (16): a one-level filter scan that keeps the bootstrap's thresholds and
    one that lowers them from the candidates found so far (scan_kernel's helper wavefront) return the
    same lists bit for bit -- every tightened threshold is an upper bound of the final k-th key -- while the tightened
    scan checks fewer pairs and keeps fewer candidates.  Also on a duplicate-heavy index (thousands of equal keys at the
    cut) and at M = 16 (the cut lives in the 16-bit accumulators' start values)."""
    from deltapq_amd import synth
    cb = synth.make_codebook(M, 256, 128 // M, seed=5)
    tree = synth.synth_tree(n, M, seed=n + M, mean_diffs=(0.4 if dup else 3.0) if M == 8 else 5.0)
    payload, _ = synth.encode_dtc(tree)
    qs = synth.make_queries(640, 128, seed=n + 1)
    res = {}
    for flags in (0, gpu.OPT_NO_TIGHTEN):
        with gpu.DeltaPQIndex.open_memory(payload, n, M, 256, flags=flags) as idx:
            idx.set_codebook(cb)
            idx.profile_enable(1)
            idx.profile_reset()
            ids, d = idx.query_batch(qs, k)
            res[flags] = (ids, d, idx.profile_read())
    assert np.array_equal(res[0][0], res[16][0]) and np.array_equal(res[0][1].view(np.uint32), res[16][1].view(np.uint32))
    on, off = res[0][2], res[16][2]
    assert on["scan_node_query_pairs"] == off["scan_node_query_pairs"]          # every pair is still visited
    assert on["candidates"] <= off["candidates"] and on["exact_checks"] <= off["exact_checks"]
    if not dup:
        assert on["candidates"] < 0.9 * off["candidates"]                       # it really tightened
    sample = [0, 63, 64, 300, 639]
    assert_parity(res[0][0][sample], res[0][1][sample], oracle_topk(oracle, payload, n, cb, qs[sample], k), n)


@pytest.mark.parametrize("M,k", [(8, 1000), (16, 1000), (8, 300)])
def test_large_topk_takes_one_level_when_the_scan_tightens(gpu, oracle, M, k):
    """top_k > 512: with the in-scan tightening the plan keeps ONE filter level (ensure_plan); without it
    (DPQ_OPT_NO_TIGHTEN) two short levels go in front.  Both answer bit for bit the same, and as the oracle.  The final
    order of more than 256 winners comes from the bitonic network (select_kernel: 512-thread blocks beyond top-512)."""
    from deltapq_amd import synth
    n = 200_000
    cb = synth.make_codebook(M, 256, 128 // M, seed=15)
    tree = synth.synth_tree(n, M, seed=k + M, mean_diffs=3.0 if M == 8 else 5.0)
    payload, _ = synth.encode_dtc(tree)
    qs = synth.make_queries(640, 128, seed=k + 1)
    res = {}
    for flags in (0, gpu.OPT_NO_TIGHTEN):
        with gpu.DeltaPQIndex.open_memory(payload, n, M, 256, flags=flags) as idx:
            idx.set_codebook(cb)
            idx.profile_enable(1)
            idx.profile_reset()
            ids, d = idx.query_batch(qs, k)
            res[flags] = (ids, d, idx.profile_read())
    assert np.array_equal(res[0][0], res[16][0]) and np.array_equal(res[0][1].view(np.uint32), res[16][1].view(np.uint32))
    on, off = res[0][2], res[16][2]
    assert on["overflow_reruns"] == 0 and off["overflow_reruns"] == 0
    assert on["scan_launches"] == 1 and off["scan_launches"] == (3 if k > 512 else 1)
    assert on["scan_node_query_pairs"] == off["scan_node_query_pairs"]          # every pair is visited either way
    sample = [0, 63, 64, 639]
    assert_parity(res[0][0][sample], res[0][1][sample], oracle_topk(oracle, payload, n, cb, qs[sample], k), n)


@pytest.mark.parametrize("M,k,dup", [(8, 1, False), (8, 100, False), (8, 100, True), (8, 256, False), (8, 300, False),
                                     (8, 1000, False), (8, 2048, False), (16, 100, False), (16, 1000, False)])
def test_histogram_selection_rules_change_nothing_but_the_work(gpu, oracle, monkeypatch, M, k, dup):
    """Round 4's one-histogram-pass rules against the exact ones they replace, bit for bit, over top_k: the bootstrap's
    threshold as the upper edge of the k-th key's bin (bootstrap_kernel<M, 1>; 0 = radix select) and the select kernel's last
    level as a bucket sort (SelectArgs.fast_final; 0 = radix select + rank count / bitonic network).  Any valid upper bound of
    the k-th key leaves the lists identical, and a bucket sort orders like any other sort; also on a duplicate-heavy index,
    where crowded bins send the select down the exact way.  The developer switches are read at dpq_open_* under DPQ_DEV=1."""
    from deltapq_amd import synth
    n = 250_000
    cb = synth.make_codebook(M, 256, 128 // M, seed=25)
    tree = synth.synth_tree(n, M, seed=k + M + 3, mean_diffs=(0.4 if dup else 3.0) if M == 8 else 5.0)
    payload, _ = synth.encode_dtc(tree)
    qs = synth.make_queries(200, 128, seed=k + 7)
    monkeypatch.setenv("DPQ_DEV", "1")
    res = {}
    for variant, fast in ((1, 1), (0, 0), (1, 0), (0, 1)):
        monkeypatch.setenv("DPQ_BOOT_VARIANT", str(variant))
        monkeypatch.setenv("DPQ_SELECT_FAST", str(fast))
        with gpu.DeltaPQIndex.open_memory(payload, n, M, 256) as idx:
            idx.set_codebook(cb)
            idx.profile_enable(1)
            idx.profile_reset()
            ids, d = idx.query_batch(qs, k)
            res[variant, fast] = (ids, d, idx.profile_read())
    base = res[1, 1]
    assert base[2]["bootstrap_launches"] >= 1
    for key, (ids, d, prof) in res.items():
        assert np.array_equal(ids, base[0]) and np.array_equal(d.view(np.uint32), base[1].view(np.uint32)), key
        assert prof["scan_node_query_pairs"] == base[2]["scan_node_query_pairs"]
    # the bound is at most one bin (2^-9 of the keys' span) above the exact k-th key: the work barely moves
    assert res[1, 1][2]["exact_checks"] <= 1.1 * res[0, 0][2]["exact_checks"] + 64
    sample = [0, 63, 64, 199]
    assert_parity(base[0][sample], base[1][sample], oracle_topk(oracle, payload, n, cb, qs[sample], k), n)


def test_codebook_can_be_set_again_between_scratch_batches(gpu, oracle, codebook):
    """dpq_set_codebook after batches that used the plain-code scratch and the relabelled tables (it once freed them):
    the same index answers for a second codebook and again for the first."""
    from deltapq_amd import synth
    n, k = 120_000, 20
    tree, payload, _ = make_case(n, seed=61)
    qs = synth.make_queries(200, 128, seed=62)
    cb2 = synth.make_codebook(8, 256, 16, seed=99)
    with gpu.DeltaPQIndex.open_memory(payload, n, 8, 256) as idx:
        idx.set_codebook(codebook)
        ids_a, d_a = idx.query_batch(qs, k)
        idx.set_codebook(cb2)
        ids_b, d_b = idx.query_batch(qs, k)
        idx.set_codebook(codebook)
        ids_c, d_c = idx.query_batch(qs, k)
    assert np.array_equal(ids_a, ids_c) and np.array_equal(d_a.view(np.uint32), d_c.view(np.uint32))
    assert_parity(ids_a[:5], d_a[:5], oracle_topk(oracle, payload, n, codebook, qs[:5], k), n)
    assert_parity(ids_b[:5], d_b[:5], oracle_topk(oracle, payload, n, cb2, qs[:5], k), n)


def test_batch_decode_m16(gpu):
    """The same for the M = 16 format extension (4-dword codes)."""
    from deltapq_amd import synth
    n, k = 150_000, 30
    codebook16 = synth.make_codebook(16, 256, 8, seed=3)
    tree = synth.synth_tree(n, 16, seed=77, mean_diffs=5.0)
    payload, _ = synth.encode_dtc(tree)
    qs = synth.make_queries(130, 128, seed=78)
    ids_f, d_f, _, _ = run(gpu, payload, n, codebook16, qs, k, M=16, batch_decode=-1)
    ids_b, d_b, prof_b, _ = run(gpu, payload, n, codebook16, qs, k, M=16, batch_decode=1)
    assert prof_b["decode_ms"] > 0.0
    assert np.array_equal(ids_f, ids_b) and np.array_equal(d_f.view(np.uint32), d_b.view(np.uint32))


def test_part_of_a_larger_index_reports_global_positions(gpu, oracle, codebook):
    """dpq_open_opts.global_offset / global_n_codes: a self-contained part of a larger index (what a rank of the
    100 M / 1 B-code runs holds).  ids = offset + local position; the even-N rule applies to the global tail only."""
    from deltapq_amd import synth
    n, nq, k, off = 300000, 16, 50, 1_000_000_000 - 300000
    tree, payload, _ = make_case(n, seed=31)
    qs = synth.make_queries(nq, 128, seed=32)
    ids0, d0, _, _ = run(gpu, payload, n, codebook, qs, k)
    ids_mid, d_mid, _, info = run(gpu, payload, n, codebook, qs, k, global_offset=12_500_000, global_n_codes=1_000_000_000)
    assert info["node_lo"] == 12_500_000 and info["n_codes_total"] == 1_000_000_000
    assert np.array_equal(d0.view(np.uint32), d_mid.view(np.uint32))
    assert np.array_equal(np.where(ids0 == n, n - 1, ids0) + 12_500_000, ids_mid)        # not the global tail: no id N
    ids_tail, d_tail, _, _ = run(gpu, payload, n, codebook, qs, k, global_offset=off, global_n_codes=1_000_000_000)
    assert np.array_equal(ids0.astype(np.int64) + off, ids_tail.astype(np.int64))          # local id n <-> global id N
    assert_parity(ids0, d0, oracle_topk(oracle, payload, n, codebook, qs, k), n)


def test_config4_per_gpu_share_125m_codes_bvecs_queries(gpu, oracle, tmp_path):
    """BASELINE configs[4]: 1 B bvecs-shaped codes over 8 GPUs = 125 M codes per GPU.  One rank's share on one
    GPU -- a self-contained part with global_offset = 3 x 125 M (rank 3 of 8), u8 queries read through
    dpq_read_vecs(.bvecs) -- oracle parity on 4 queries, size-independent properties on the whole batch."""
    import time
    from deltapq_amd import synth
    n, nq, k, rank, N = 125_000_000, 128, 100, 3, 1_000_000_000
    t0 = time.time()
    tree = synth.synth_tree_large(n, 8, seed=102 + 1000 * rank, mean_diffs=3.0)
    payload, nb = synth.encode_dtc(tree)
    del tree
    cb = synth.make_codebook(8, 256, 16, seed=100)
    qpath = str(tmp_path / "query.bvecs")
    synth.write_bvecs(qpath, synth.make_queries(nq, 128, seed=5))
    qs = gpu.read_vecs(qpath, ext="bvecs")
    assert qs.shape == (nq, 128) and qs.dtype == np.float32 and np.all(qs == np.rint(qs))
    with gpu.DeltaPQIndex.open_memory(payload, n, 8, 256, global_offset=rank * n, global_n_codes=N) as idx:
        idx.set_codebook(cb)
        idx.profile_enable(True)
        ids, dists = idx.query_batch(qs, k)
        prof, info = idx.profile_read(), idx.info()
    assert info["node_lo"] == rank * n and info["node_hi"] == (rank + 1) * n and info["algorithmic_bytes"] == nb
    assert info["bootstrap_stride"] >= 16 and info["device_bytes"] > 256 << 20          # beyond L2 + Infinity Cache
    S = 64 * info["chunks_per_segment"]
    assert prof["scan_node_query_pairs"] == info["n_segments"] * S * nq               # every node filtered once per query
    assert np.all(np.diff(dists, axis=1) >= 0)
    for r in range(nq):
        assert len(set(ids[r].tolist())) == k and rank * n <= ids[r].min() and ids[r].max() < (rank + 1) * n
    sample = [0, 41, 83, 127]
    local = ids[sample].astype(np.int64) - rank * n
    ref = oracle_topk(oracle, payload, n, cb, qs[sample], k)
    ref = [(np.where(oi == n, n - 1, oi), od, alld) for oi, od, alld in ref]             # a part, not the tail: no id-N rule
    assert_parity(local, dists[sample], ref, n | 1)
    print("config4 share: %.0f s" % (time.time() - t0))


def _eight_way_decomposition(gpu, oracle, N, nq, k, checked, queries, world=8):
    """BASELINE configs[3]/[4] as what they name, share after share on ONE GPU: rank r's DFS range synthesised exactly
    as bench.py does (seed 102 + 1000 r), opened as a self-contained part (global_offset / global_n_codes), the whole
    batch answered on it; the `checked` queries are verified against the oracle run on that share; the 8 partial
    lists are merged on the host (dpq_merge_topk_host) AND on the device from the packed [8][nq][2k] tensor the
    all-gather would deliver (dpq_merge_topk_device_packed); the merged lists of the checked queries must be the top-k
    of the union of the oracle's partial lists.  Properties on the whole batch; the even-N id rule on the global tail only."""
    import time
    import torch
    from deltapq_amd import dist as dpq_dist, synth
    from oracle.dtc_oracle import tie_aware_equal
    cb = synth.make_codebook(8, 256, 16, seed=100)
    per = N // world
    part_ids, part_d, ora = [], [], {q: [] for q in checked}
    pairs = 0
    for r in range(world):
        t0 = time.time()
        n = per if r + 1 < world else N - per * (world - 1)
        tree = synth.synth_tree_large(n, 8, seed=102 + 1000 * r, mean_diffs=3.0)
        payload, nb = synth.encode_dtc(tree)
        del tree
        with gpu.DeltaPQIndex.open_memory(payload, n, 8, 256, global_offset=r * per, global_n_codes=N) as idx:
            idx.set_codebook(cb)
            idx.profile_enable(True)
            ids, dists = idx.query_batch(queries, k)
            prof, info = idx.profile_read(), idx.info()
        assert info["node_lo"] == r * per and info["node_hi"] == r * per + n and info["algorithmic_bytes"] == nb
        S = 64 * info["chunks_per_segment"]
        assert prof["scan_node_query_pairs"] == info["n_segments"] * S * nq      # every node filtered once per query
        pairs += n * nq
        assert np.all(np.diff(dists, axis=1) >= 0)
        tail = r + 1 == world and N % 2 == 0
        for q in range(nq):
            row = ids[q].astype(np.int64)
            assert len(set(row.tolist())) == k and row.min() >= r * per
            assert row.max() < r * per + n or (tail and row.max() == N)          # id N: the global tail only (h:2949, 2970)
            assert (N - 1) not in row.tolist() if tail else True
        for q in checked:
            lut = oracle.build_lut(cb, queries[q])
            oi, od, alld, _ = oracle.scan_lut(payload, n, lut, k, want_all=True)
            if n % 2 == 0:    # the oracle sees a whole index of n codes: fold its id n back, then the global tail's forward
                oi = np.where(oi == n, n - 1, oi)
            mine = ids[q].astype(np.int64) - r * per
            if tail:
                mine = np.where(mine == n, n - 1, mine)
            ok, msg = tie_aware_equal(mine, dists[q], oi, od, alld, n | 1)
            assert ok, "share %d query %d: %s" % (r, q, msg)
            g = oi.astype(np.int64) + r * per
            ora[q].append((np.where(g == N - 1, N, g) if tail else g, od))
        part_ids.append(ids)
        part_d.append(dists)
        del payload
        print("share %d of %d (%d codes): %.0f s" % (r, world, n, time.time() - t0), flush=True)
    pi, pd = np.stack(part_ids), np.stack(part_d)
    mi, md = gpu.merge_topk_host(pi, pd)
    packed = torch.stack([dpq_dist.pack_lists(torch.from_numpy(part_ids[r]).cuda(), torch.from_numpy(part_d[r]).cuda())
                          for r in range(world)]).contiguous()                   # [8][nq][2k]: the all-gather's layout
    di, dd = gpu.merge_topk_packed_torch(packed, k)
    torch.cuda.synchronize()
    assert np.array_equal(di.cpu().numpy(), mi) and np.array_equal(dd.cpu().numpy().view(np.uint32), md.view(np.uint32))
    # every merged row = the k smallest (distance, id) keys of the union of the 8 partial rows
    keys = (pd.view(np.uint32).astype(np.uint64) << np.uint64(32)) | pi.astype(np.uint32).astype(np.uint64)
    want = np.sort(keys.transpose(1, 0, 2).reshape(nq, -1), axis=1)[:, :k]
    got = (md.view(np.uint32).astype(np.uint64) << np.uint64(32)) | mi.astype(np.uint32).astype(np.uint64)
    assert np.array_equal(got, want)
    for q in checked:   # ... and, for the checked queries, of the union of the ORACLE's partial lists (tie-aware at the boundary)
        oi = np.concatenate([a for a, _ in ora[q]])
        od = np.concatenate([b for _, b in ora[q]])
        order = np.lexsort((oi, od.view(np.uint32)))[:k]
        assert np.array_equal(md[q].view(np.uint32), od[order].view(np.uint32))
        below = od[order] < od[order][-1]
        assert set(mi[q][below[:k]].tolist()) == set(oi[order][below].tolist())
    return pairs


def test_config3_100m_codes_as_eight_shares_and_merge(gpu, oracle):
    """BASELINE configs[3] -- 100 M codes over 8 GPUs, per-GPU partial top-k host-merged -- executed as the 8-way
    decomposition it names, one share after the other on this GPU."""
    from deltapq_amd import synth
    nq = 64
    qs = synth.make_queries(nq, 128, seed=8)
    _eight_way_decomposition(gpu, oracle, 100_000_000, nq, 100, [0, 31, 63], qs)


def test_config4_1b_codes_as_eight_parts_and_packed_gather_merge(gpu, oracle, tmp_path):
    """BASELINE configs[4] -- 1 B bvecs-shaped codes over 8 GPUs with a gather of the candidate lists -- executed as
    its 8 parts of 125 M codes, one after the other on this GPU; u8 queries through dpq_read_vecs(.bvecs); the merge
    consumes the packed [8][nq][2k] tensor the RCCL all-gather delivers."""
    from deltapq_amd import synth
    nq = 32
    qpath = str(tmp_path / "query.bvecs")
    synth.write_bvecs(qpath, synth.make_queries(nq, 128, seed=5))
    qs = gpu.read_vecs(qpath, ext="bvecs")
    assert qs.shape == (nq, 128) and np.all(qs == np.rint(qs))
    _eight_way_decomposition(gpu, oracle, 1_000_000_000, nq, 100, [0, 17], qs)


M16_SHAPES = [(1, 3, 1), (2, 2, 2), (65, 5, 10), (1000, 20, 10), (10000, 50, 100), (100001, 40, 1000), (300000, 33, 100)]


@pytest.mark.parametrize("n,nq,k", M16_SHAPES)
def test_m16_parity(gpu, oracle, n, nq, k):
    """BASELINE configs[2] family (16-byte codes).  The reference has no M = 16 query
    (format hard-wired to M <= 8), so the oracle here is the same stack machine on
    this build's format extension: a consistency check, plus losslessness."""
    from deltapq_amd import synth
    cb = synth.make_codebook(16, 256, 8, seed=3)
    tree = synth.synth_tree(n, 16, seed=n + 1, mean_diffs=5.0)
    payload, nb = synth.encode_dtc(tree)
    qs = synth.make_queries(nq, 128, seed=n + 2)
    ids, dists, prof, info = run(gpu, payload, n, cb, qs, k, M=16)
    assert info["algorithmic_bytes"] == nb and info["M"] == 16
    assert_parity(ids, dists, oracle_topk(oracle, payload, n, cb, qs, k), n)


def test_m16_sift1m_shape_top1000(gpu, oracle):
    """BASELINE configs[2] at full size: 1M 16-byte codes, top-1000, sharded 2 ways as well."""
    from deltapq_amd import synth
    n, nq, k = 1_000_000, 48, 1000
    cb = synth.make_codebook(16, 256, 8, seed=5)
    tree = synth.synth_tree(n, 16, seed=6, mean_diffs=5.0)
    payload, nb = synth.encode_dtc(tree)
    codes = synth.decode_tree_codes(tree)
    del tree
    qs = synth.make_queries(nq, 128, seed=7)
    ids, dists, prof, info = run(gpu, payload, n, cb, qs, k, M=16)
    sample = [0, 17, 47]
    assert_parity(ids[sample], dists[sample], oracle_topk(oracle, payload, n, cb, qs[sample], k), n)
    assert np.all(np.diff(dists, axis=1) >= 0)
    for r in range(nq):
        assert len(set(ids[r].tolist())) == k
        lut = oracle.build_lut(cb, qs[r])
        pos = np.where(ids[r] == n, n - 1, ids[r])
        s = sum(lut[m, codes[pos, m]].astype(np.float64) for m in range(16)).astype(np.float32)
        assert np.array_equal(s.view(np.uint32), dists[r].view(np.uint32))
    parts = [run(gpu, payload, n, cb, qs[sample], k, M=16, shard_rank=r, shard_count=2)[:2] for r in range(2)]
    mi, md = gpu.merge_topk_host(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
    assert np.array_equal(md.view(np.uint32), dists[sample].view(np.uint32))


def test_tie_explosion_thousands_of_duplicates_of_the_nearest_code(gpu, oracle, codebook):
    """Pathological ties: 9000 exact copies of the code nearest to the query.  The
    lower-bound filter cannot separate equal distances, so every copy reaches the
    scan's exact check; that check compares whole (distance, id) keys, so only
    copies below the threshold id become candidates and the canonical answer
    (lowest ids of the tie group) comes out without a rerun."""
    from deltapq_amd import api, synth
    rng = np.random.default_rng(5)
    n = 40000
    codes = rng.integers(0, 256, size=(n, 8), dtype=np.uint8)
    q = synth.make_queries(3, 128, seed=9)
    lut = oracle.build_lut(codebook, q[0])
    best = np.array([int(np.argmin(lut[m])) for m in range(8)], dtype=np.uint8)   # the nearest possible code
    dup_at = rng.choice(n, 9000, replace=False)
    codes[dup_at] = best
    tree = api.DeltaTree(codes)
    payload = tree.payload()
    ids, dists, prof, _ = run(gpu, payload, n, codebook, q, 100)
    assert prof["overflow_reruns"] == 0
    assert_parity(ids, dists, oracle_topk(oracle, payload, n, codebook, q, 100), n)
    # query 0: all 100 results are copies of `best`, and they are the 100 lowest DFS positions holding it
    pos_of_best = np.flatnonzero((codes[tree.vec_id] == best).all(1))
    assert np.array_equal(np.sort(ids[0]), pos_of_best[:100]) and len(set(dists[0].tolist())) == 1


def plain_dists(lut, codes):
    """h:2658-2662: float dist = 0; for m: dist += lut[m][code]  (fp32, m ascending)."""
    d = np.zeros(len(codes), dtype=np.float32)
    for m in range(codes.shape[1]):
        d = (d + lut[m, codes[:, m]]).astype(np.float32)
    return d


@pytest.mark.parametrize("n,nq,k,M", [(1, 2, 1, 8), (300, 5, 50, 8), (20000, 40, 10, 8), (200000, 33, 100, 8),
                                       (50000, 10, 100, 16)])
def test_plain_pqscan_comparator(gpu, oracle, n, nq, k, M):
    """`-task pqscan` (h:2590-2678): raw codes, fp32 accumulation, ids = file positions."""
    from oracle.dtc_oracle import tie_aware_equal
    from deltapq_amd import synth
    rng = np.random.default_rng(n)
    protos = rng.integers(0, 256, size=(max(2, n // 20), M), dtype=np.uint8)
    codes = protos[rng.integers(0, len(protos), size=n)].copy()                 # many exact duplicates -> ties
    codes[np.arange(n), rng.integers(0, M, size=n)] = rng.integers(0, 256, size=n)
    cb = synth.make_codebook(M, 256, 128 // M, seed=1)
    qs = synth.make_queries(nq, 128, seed=2)
    with gpu.DeltaPQIndex.open_plain(codes) as idx:
        idx.set_codebook(cb)
        ids, dists = idx.query_batch(qs, k)
        info = idx.info()
    assert info["algorithmic_bytes"] == n * M
    for i in range(nq):
        lut = oracle.build_lut(cb, qs[i])
        oi, od = oracle.pqscan_plain(codes, lut, k)
        ok, msg = tie_aware_equal(ids[i], dists[i], oi, od, plain_dists(lut, codes), n + 1)   # n+1: odd/even quirk off
        assert ok, "query %d: %s" % (i, msg)
    # sharded: positions stay global
    parts = []
    for r in range(3):
        with gpu.DeltaPQIndex.open_plain(codes, shard_rank=r, shard_count=3) as idx:
            idx.set_codebook(cb)
            parts.append(idx.query_batch(qs, min(k, n)))
    mi, md = gpu.merge_topk_host(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
    assert np.array_equal(md.view(np.uint32), dists.view(np.uint32)) and np.array_equal(mi, ids)


def test_cli_pqscan_and_compressed_query_agree_through_vec_id(gpu, oracle, tmp_path):
    """The compressed scan (DFS positions, fp64 rule) and the plain scan (file positions, fp32 rule) find the
    same vectors: map the former through QNode.vec_id (TreeNodesDFS file) and compare as sets."""
    from deltapq_amd import api, synth
    from oracle import pq_encode_oracle
    d = str(tmp_path)
    n, nq, k = 20001, 16, 10
    base = synth.make_clustered_vectors(n, 128, seed=3, n_clusters=500)
    queries = synth.make_clustered_vectors(nq, 128, seed=4, n_clusters=500, centre_seed=3)
    cb = synth.kmeans_codebook(base, 8, 256, iters=3, seed=5)
    synth.write_codewords_txt(os.path.join(d, "M8K256codewords.txt"), cb)
    cb = synth.read_codewords_txt(os.path.join(d, "M8K256codewords.txt"))
    synth.write_fvecs(os.path.join(d, "query.fvecs"), queries)
    synth.write_fvecs(os.path.join(d, "base.fvecs"), base)
    exe = os.path.join(ROOT, "deltapq_amd", "csrc", "deltapq")
    common = ["-dataset", d, "-m", "8", "-k", "256", "-N", str(n), "-topk", str(k), "-query_size", str(nq)]
    r = subprocess.run([exe, "-task", "encode"] + common, capture_output=True, text=True, timeout=300)   # base.fvecs -> codes
    assert r.returncode == 0, r.stdout + r.stderr
    codes = api.read_codes_plain(os.path.join(d, "codes.bin.plain.M8K256N%d" % n), 8)
    assert np.array_equal(codes, pq_encode_oracle.encode_pq(base, cb))
    r = subprocess.run([exe, "-task", "approx_tree"] + common, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    res = {}
    for task in ("query", "pqscan"):
        out = os.path.join(d, task + ".bin")
        r = subprocess.run([exe, "-task", task, "-out", out] + common, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "[msec/query]" in r.stdout, r.stdout + r.stderr
        raw = np.fromfile(out, dtype=np.uint8)
        res[task] = (np.frombuffer(raw[16:16 + nq * k * 4], np.int32).reshape(nq, k),
                     np.frombuffer(raw[16 + nq * k * 4:], np.float32).reshape(nq, k))
    vec_id = api.read_qnode_ids(os.path.join(d, "M8K256_Approx_TreeNodesDFS_N%d" % n), n)
    for i in range(nq):
        lut = oracle.build_lut(cb, queries[i])
        d32 = plain_dists(lut, codes)
        got_plain, got_tree = res["pqscan"][0][i], vec_id[res["query"][0][i]]
        assert np.array_equal(d32[got_plain].view(np.uint32), res["pqscan"][1][i].view(np.uint32))
        # same k-th distance up to fp32-vs-fp64 rounding; the id sets may differ only inside that band
        kth = res["pqscan"][1][i][-1]
        assert abs(res["query"][1][i][-1] - kth) <= 2e-6 * kth
        strictly_inside = d32 < kth * (1 - 2e-6)
        assert set(np.flatnonzero(strictly_inside).tolist()) <= set(got_tree.tolist()) & set(got_plain.tolist())


@pytest.mark.parametrize("K,Ds", [(64, 4), (200, 16), (1, 2), (256, 32)])
def test_other_codebook_shapes(gpu, oracle, K, Ds):
    """K < 256 (codes only use k < K) and other sub-space widths."""
    from deltapq_amd import synth
    n, nq, k = 5000, 9, 25
    rng = np.random.default_rng(K)
    cb = rng.normal(10, 4, size=(8, K, Ds)).astype(np.float32)
    qs = rng.normal(10, 4, size=(nq, 8 * Ds)).astype(np.float32)
    tree = synth.synth_tree(n, 8, seed=K + 1)
    tree["deltas"] = (tree["deltas"].astype(np.int64) % K).astype(np.uint8)
    tree["root"] = (tree["root"].astype(np.int64) % K).astype(np.uint8)
    payload, _ = synth.encode_dtc(tree)
    with gpu.DeltaPQIndex.open_memory(payload, n, 8, K) as idx:
        idx.set_codebook(cb)
        ids, dists = idx.query_batch(qs, k)
    assert_parity(ids, dists, oracle_topk(oracle, payload, n, cb, qs, k), n)


def test_adversarial_table_magnitudes(gpu, oracle):
    """`fp64 sum of the fp32 entries` equals the reference's incremental fp64 stack
    bit for bit while partial sums are exact in fp64 (entries within ~2^29 of each other).  Here sub-spaces
    differ by 1e5 in scale (1e10 in squared distance): the two summation orders may round differently, and
    the contract falls back to the north-star tolerance: 1e-5 relative on distances, same ids where the
    distances are separated by more than that."""
    from deltapq_amd import synth
    rng = np.random.default_rng(0)
    n, nq, k = 30000, 12, 50
    scale = np.array([1e5, 1e-2, 3.0, 1e2, 1e-3, 7.0, 1e4, 1.0], dtype=np.float64)
    cb = (rng.normal(0, 1, size=(8, 256, 16)) * scale[:, None, None]).astype(np.float32)
    qs = (rng.normal(0, 1, size=(nq, 8, 16)) * scale[None, :, None]).reshape(nq, 128).astype(np.float32)
    tree, payload, _ = make_case(n, seed=3)
    ids, dists, _, _ = run(gpu, payload, n, cb, qs, k)
    exact_rows = 0
    for i, (oi, od, alld) in enumerate(oracle_topk(oracle, payload, n, cb, qs, k)):
        assert np.allclose(dists[i], od, rtol=1e-5, atol=0.0), i
        pos = np.where(ids[i] == n, n - 1, ids[i])
        assert np.allclose(alld[pos], dists[i], rtol=1e-5, atol=0.0)        # every returned id carries its distance
        exact_rows += int(np.array_equal(dists[i].view(np.uint32), od.view(np.uint32)) and set(ids[i]) == set(oi))
    assert exact_rows >= nq - 2          # in practice still bit-identical almost always


# (n_codes, n_queries, top_k, bootstrap option, dup_heavy)
BOOT_CASES = [(300001, 40, 100, 0, False), (20000, 33, 10, 1, False), (70000, 20, 100, 0, True), (130000, 24, 100, 0, False),
              (400000, 24, 1000, 0, False), (262144, 16, 2048, 0, False)]


@pytest.mark.parametrize("n,nq,k,boot,dup", BOOT_CASES)
def test_threshold_bootstrap_gives_the_same_answer(gpu, oracle, codebook, n, nq, k, boot, dup):
    """The first threshold comes from the query's best multi-index cells (bootstrap_kernel) instead of
    the spread sample: identical results to the oracle and to the spread-sample cascade (bootstrap = -1)."""
    from deltapq_amd import synth
    tree, payload, nb = make_case(n, seed=n + 3, dup_heavy=dup)
    qs = synth.make_queries(nq, 128, seed=n + 4)
    ids, dists, prof, info = run(gpu, payload, n, codebook, qs, k, bootstrap=boot)
    classes = 4 if n >= 4 * 65536 else 1
    assert info["bootstrap_stride"] == 1 and info["bootstrap_bytes"] == 4 * (classes * 65537 + 3 * n)
    ids0, dists0, _, info0 = run(gpu, payload, n, codebook, qs, k, bootstrap=-1)
    assert info0["bootstrap_bytes"] == 0
    assert np.array_equal(dists.view(np.uint32), dists0.view(np.uint32))
    if not dup:
        assert np.array_equal(ids, ids0)
    sample = list(range(0, nq, 4))
    assert_parity(ids[sample], dists[sample], oracle_topk(oracle, payload, n, codebook, qs[sample], k), n)
    S = 64 * info["chunks_per_segment"]
    assert prof["scan_node_query_pairs"] == info["n_segments"] * S * nq      # every node filtered exactly once


def test_threshold_bootstrap_on_shards_and_plain_index(gpu, oracle, codebook):
    """Shards carry their own multi-index with global positions; the plain comparator index builds it from the raw codes."""
    from deltapq_amd import synth
    n, nq, k = 700000, 16, 100
    tree, payload, nb = make_case(n, seed=55)
    qs = synth.make_queries(nq, 128, seed=56)
    parts = []
    for r in range(2):
        ids, dists, _, info = run(gpu, payload, n, codebook, qs, k, shard_rank=r, shard_count=2)
        assert info["bootstrap_stride"] == 1 and info["node_hi"] - info["node_lo"] >= 262144
        parts.append((ids, dists))
    mi, md = gpu.merge_topk_host(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
    assert_parity(mi, md, oracle_topk(oracle, payload, n, codebook, qs, k), n)
    codes = synth.decode_tree_codes(tree)
    outs = []
    for boot in (0, -1):
        with gpu.DeltaPQIndex.open_plain(codes, bootstrap=boot) as idx:
            idx.set_codebook(codebook)
            outs.append(idx.query_batch(qs, k) + (idx.info()["bootstrap_bytes"],))
    assert outs[0][2] > 0 and outs[1][2] == 0
    assert np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32)) and np.array_equal(outs[0][0], outs[1][0])
    for i in range(0, nq, 5):
        lut = oracle.build_lut(codebook, qs[i])
        oi, od = oracle.pqscan_plain(codes, lut, k)
        assert np.array_equal(np.sort(outs[0][1][i]).view(np.uint32), np.sort(od).view(np.uint32))


@pytest.mark.parametrize("n_scan", [1, 2, 129, 4095, 4096, 20001, 29998])
def test_prefix_scan_minus_N_below_n_codes(gpu, oracle, codebook, n_scan):
    """`-N` smaller than the header's n_codes (h:2825-2829).  Odd N: the reference's own result (the oracle
    restates the loop bounds).  Even N: the reference reads a pair byte as a whole-byte depth (undefined);
    this build answers like an index holding exactly the first N codes (trailing rule, id N)."""
    from deltapq_amd import synth
    n, nq = 30000, 12
    k = min(10, n_scan)
    tree, payload, _ = make_case(n, seed=77)
    qs = synth.make_queries(nq, 128, seed=78)
    ids, dists, _, info = run(gpu, payload, n, codebook, qs, k, num_codes=n_scan)
    assert info["node_hi"] == n_scan and info["n_codes_total"] == n_scan
    if n_scan % 2 == 1:
        ref = oracle_topk(oracle, payload, n_scan, codebook, qs, k)          # scans the first n_scan codes of the long stream
    else:
        sub = dict(root=tree["root"], depths=tree["depths"][:n_scan], masks=tree["masks"][:n_scan], M=8,
                   deltas=tree["deltas"][:int(sum(bin(int(m)).count("1") for m in tree["masks"][1:n_scan]))])
        p2, _ = synth.encode_dtc(sub)
        ref = oracle_topk(oracle, p2, n_scan, codebook, qs, k)
    assert_parity(ids, dists, ref, n_scan)
    # two shards of the prefix
    parts = [run(gpu, payload, n, codebook, qs, k, num_codes=n_scan, shard_rank=r, shard_count=2)[:2] for r in range(2)]
    mi, md = gpu.merge_topk_host(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
    assert_parity(mi, md, ref, n_scan)


@pytest.mark.parametrize("n_scan", [70_001, 123_458, 200_000])
def test_stream_pass_on_a_prefix_and_on_a_part_of_a_larger_index(gpu, oracle, codebook, n_scan):
    """The strand image (forced: flags = 64) built for a prefix scan (`-N` below n_codes: odd, even -> the trailing id rule)
    and for a part of a larger index (global positions), on a whole shard and on shard 1 of 3: one-, two- and four-query
    calls against the same queries inside a 70-query batch (filter path), and against the oracle."""
    from deltapq_amd import synth
    n, k = 200_000, 20
    tree, payload, _ = make_case(n, seed=301)
    qs = synth.make_queries(70, 128, seed=302)
    for kw in (dict(num_codes=n_scan), dict(num_codes=n_scan, shard_rank=1, shard_count=3),
               dict(global_offset=1_000_000_000 - n if n_scan == n else 12_345_678, global_n_codes=1_000_000_000)):
        if "global_offset" in kw and n_scan != n and n_scan != 70_001:
            continue
        with gpu.DeltaPQIndex.open_memory(payload, n, 8, 256, flags=gpu.OPT_FORCE_STRANDS, bootstrap=1, **kw) as idx:
            idx.set_codebook(codebook)
            ids_b, d_b = idx.query_batch(qs, k)
            for lo, hi in ((0, 1), (3, 5), (10, 14)):
                ids_s, d_s = idx.query_batch(qs[lo:hi], k)
                assert np.array_equal(ids_s, ids_b[lo:hi]) and np.array_equal(d_s.view(np.uint32), d_b[lo:hi].view(np.uint32)), (kw, lo)
        if kw == dict(num_codes=n_scan) and n_scan % 2 == 1:
            assert_parity(ids_b[:4], d_b[:4], oracle_topk(oracle, payload, n_scan, codebook, qs[:4], k), n_scan)
