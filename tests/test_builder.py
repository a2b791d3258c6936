"""DeltaTree builder (host), codes.bin.plain / TreeNodesDFS files,
PQ encoder (GPU) -- the callers either side of the query path."""
import os

import numpy as np
import pytest


def clustered_codes(n, M=8, seed=0, n_protos=None):
    """PQ-code-like data: prototypes + few-position perturbations + exact duplicates."""
    rng = np.random.default_rng(seed)
    n_protos = n_protos or max(4, n // 50)
    protos = rng.integers(0, 256, size=(n_protos, M), dtype=np.uint8)
    codes = protos[rng.integers(0, n_protos, size=n)].copy()
    nchg = rng.integers(0, 4, size=n)
    for i in range(n):
        for pos in rng.choice(M, nchg[i], replace=False):
            codes[i, pos] = rng.integers(0, 256)
    return codes


def tree_decode(tree):
    """Codes of every DFS position from the builder's arrays."""
    from deltapq_amd import synth
    t = dict(root=tree.root, depths=tree.depth, masks=tree.mask, deltas=tree.deltas, M=tree.M)
    return synth.decode_tree_codes(t)


@pytest.mark.parametrize("n,M", [(1, 8), (2, 8), (3, 8), (500, 8), (20000, 8), (3000, 16)])
def test_builder_is_lossless_and_valid(lib, oracle, n, M):
    from deltapq_amd import api
    codes = clustered_codes(n, M, seed=n)
    tree = api.DeltaTree(codes, K=256, max_height_folds=1)
    assert sorted(tree.vec_id.tolist()) == list(range(n))             # a permutation: every code exactly once
    assert np.array_equal(tree_decode(tree), codes[tree.vec_id])      # lossless
    assert tree.depth[0] == 0 and (n == 1 or tree.depth[1:].min() >= 1)
    assert tree.stats["max_depth"] <= (7 if M <= 8 else 15)           # height cap M*h keeps the 3-bit depth field valid
    assert np.all(tree.depth[1:].astype(int) <= tree.depth[:-1].astype(int) + 1)   # DFS layout
    assert len(tree.edges) == n - 1
    if n > 1:
        par = tree.parent_pos[1:]
        assert np.all(par < np.arange(1, n)) and np.all(tree.depth[1:] == tree.depth[par] + 1)
    # masks are exactly the differing positions w.r.t. the parent
    if n > 1:
        child, parent = codes[tree.vec_id[1:]], codes[tree.vec_id[tree.parent_pos[1:]]]
        bits = ((child != parent) * (1 << np.arange(M))).sum(1)
        assert np.array_equal(bits.astype(np.uint16), tree.mask[1:])
    payload = tree.payload()
    st = api.dtc_validate(payload, n, M)
    assert st["n_diffs"] == tree.stats["n_diffs"] and len(payload) == tree.stats["n_bytes"]
    lut = np.random.default_rng(1).random((M, 256)).astype(np.float32)
    _, _, _, allc = oracle.scan_lut(payload, n, lut, 1, want_all=True)
    assert np.array_equal(allc, codes[tree.vec_id])                   # the oracle's scan decodes the same codes


def test_builder_exploits_similarity(lib):
    """Duplicates become 0-diff children and near-duplicates 1-3-diff children: far fewer
    changed bytes than M per code (the point of DeltaPQ)."""
    from deltapq_amd import api
    n = 30000
    codes = clustered_codes(n, 8, seed=3, n_protos=300)
    tree = api.DeltaTree(codes)
    assert tree.stats["n_diffs"] < 2.5 * n
    assert tree.stats["n_bytes"] < 0.55 * 8 * n                       # < 55 % of raw PQ
    assert (tree.mask[1:] == 0).sum() >= n - len(np.unique(codes, axis=0)) - 1
    rnd = np.random.default_rng(0).integers(0, 256, size=(n, 8), dtype=np.uint8)
    assert api.DeltaTree(rnd).stats["n_diffs"] > tree.stats["n_diffs"] * 2   # random codes compress worse


def test_sibling_order_uses_codebook(lib, codebook):
    from deltapq_amd import api
    codes = clustered_codes(5000, 8, seed=5)
    a, b = api.DeltaTree(codes), api.DeltaTree(codes, codebook=codebook)
    assert np.array_equal(np.sort(a.edges.view([("p", "u4"), ("c", "u4")]).ravel()),
                          np.sort(b.edges.view([("p", "u4"), ("c", "u4")]).ravel()))   # same tree ...
    assert not np.array_equal(a.vec_id, b.vec_id)                                      # ... other sibling order
    assert np.array_equal(tree_decode(b), codes[b.vec_id])


def test_reference_artefact_files(lib, tmp_path, codebook):
    from deltapq_amd import api, synth
    d = str(tmp_path)
    n = 4001
    codes = clustered_codes(n, 8, seed=7)
    path = os.path.join(d, "codes.bin.plain.M8K256N%d" % n)
    api.write_codes_plain(path, codes)
    assert os.path.getsize(path) == 8 + n * 8                          # int64 N + N*M bytes (pq_tree.cpp:1011-1031)
    assert np.array_equal(api.read_codes_plain(path, 8), codes)
    tree = api.DeltaTree(codes, codebook=codebook)
    tree.write_files(d)
    nodes = os.path.join(d, "M8K256_Approx_TreeNodesDFS_N%d" % n)
    assert os.path.getsize(nodes) == 60 * (n + 1)                      # QNode is 60 bytes, N+1 records (h:1484)
    assert np.array_equal(api.read_qnode_ids(nodes, n), tree.vec_id)
    rec = np.fromfile(nodes, dtype=np.uint8).reshape(n + 1, 60)
    assert np.array_equal(rec[:n, 33], tree.depth) and rec[0, 32] == 8
    assert np.array_equal(rec[0, 36:58:3], codes[tree.vec_id[0]])      # root diffs[m].to = root code (h:1437-1441)
    n_codes, payload = api.read_dtc_file(synth.dtc_file_name(d, 8, 256, n))
    assert n_codes == n and np.array_equal(payload, tree.payload())
    edges = np.fromfile(os.path.join(d, "M8K256H1_Approx_Edges_N%d" % n), dtype=np.uint32)
    assert edges[0] == tree.vec_id[0] and len(edges) == 1 + 2 * (n - 1)


def test_numpy_encoder_reference_is_nearest_centroid():
    from deltapq_amd import synth
    from oracle import pq_encode_oracle
    v = synth.make_clustered_vectors(300, 128, seed=1, n_clusters=20)
    cb = synth.kmeans_codebook(v, 8, 16, iters=3, seed=2)
    codes = pq_encode_oracle.encode_pq(v, cb)
    d = ((v.reshape(300, 8, 1, 16).astype(np.float64) - cb[None].astype(np.float64)) ** 2).sum(-1)
    assert (codes == d.argmin(-1)).mean() > 0.99                       # fp32 vs fp64 may differ only on near ties


@pytest.mark.gpu
def test_gpu_pq_encoder_matches_fp32_reference(lib):
    from deltapq_amd import api, synth
    from oracle import pq_encode_oracle
    if api.device_count() < 1:
        pytest.fail("no GPU")
    v = synth.make_clustered_vectors(20000, 128, seed=3, n_clusters=400)
    cb = synth.kmeans_codebook(v, 8, 256, iters=4, seed=4)
    assert np.array_equal(api.encode_pq(v, cb), pq_encode_oracle.encode_pq(v, cb))     # bit-for-bit the same argmin
    cb16 = synth.kmeans_codebook(v, 16, 64, iters=2, seed=5)
    assert np.array_equal(api.encode_pq(v[:3000], cb16), pq_encode_oracle.encode_pq(v[:3000], cb16))


@pytest.mark.gpu
def test_end_to_end_vectors_to_query(lib, oracle):
    """learn -> encode (GPU) -> build tree (host) -> query (GPU): results, mapped back through
    vec_id, are the true PQ nearest neighbours of the raw codes."""
    from conftest import assert_parity, oracle_topk
    from deltapq_amd import api, synth
    if api.device_count() < 1:
        pytest.fail("no GPU")
    n, nq, k = 60000, 32, 20
    base = synth.make_clustered_vectors(n, 128, seed=11, n_clusters=1500)
    queries = synth.make_clustered_vectors(nq, 128, seed=12, n_clusters=1500)
    cb = synth.kmeans_codebook(base, 8, 256, iters=4, seed=13)
    codes = api.encode_pq(base, cb)
    tree = api.DeltaTree(codes, codebook=cb)
    payload = tree.payload()
    assert len(payload) < 0.8 * n * 8
    with api.DeltaPQIndex.open_memory(payload, n, 8, 256) as idx:
        idx.set_codebook(cb)
        ids, dists = idx.query_batch(queries, k)
    assert_parity(ids, dists, oracle_topk(oracle, payload, n, cb, queries, k), n)
    pos = np.where(ids == n, n - 1, ids)                                # even-N quirk
    orig = tree.vec_id[pos]
    for i in range(nq):
        lut = oracle.build_lut(cb, queries[i])
        alld = sum(lut[m, codes[:, m]].astype(np.float64) for m in range(8)).astype(np.float32)
        assert np.array_equal(alld[orig[i]].view(np.uint32), dists[i].view(np.uint32))
        assert np.sort(alld)[k - 1] == dists[i][-1]                     # really the k best of the raw codes


@pytest.mark.parametrize("n,M,with_cb", [(1, 8, False), (2, 8, True), (3, 8, True), (700, 8, True), (6000, 8, True),
                                          (6000, 8, False), (2500, 16, True)])
def test_host_builder_matches_the_oracle_restatement(lib, n, M, with_cb):
    """The product's host builder against oracle/builder_oracle.py (an independent numpy restatement of
    h:445-627, h:1207-1313, h:1334-1487, h:1156-1183): same edges in the same order, same DFS layout, same stream."""
    from deltapq_amd import api, synth
    from oracle import builder_oracle
    codes = clustered_codes(n, M, seed=3 * n + M)
    cb = synth.make_codebook(M, 256, 4, seed=n) if with_cb else None
    ref = builder_oracle.build(codes, cb)
    t = api.DeltaTree(codes, codebook=cb)
    assert np.array_equal(t.edges, ref["edges"].reshape(-1, 2))
    assert np.array_equal(t.vec_id, ref["vec_id"]) and np.array_equal(t.depth, ref["depths"])
    assert np.array_equal(t.mask, ref["masks"]) and np.array_equal(t.deltas, ref["deltas"])
    assert np.array_equal(t.payload(), synth.encode_dtc(ref)[0])


@pytest.mark.gpu
@pytest.mark.parametrize("n,M", [(1, 8), (2, 8), (777, 8), (20000, 8), (8000, 16)])
def test_gpu_builder_matches_the_oracle_restatement(lib, n, M):
    """dpq_tree_build_gpu (edge search and tree layout on the device) against oracle/builder_oracle.py."""
    from deltapq_amd import api, synth
    from oracle import builder_oracle
    if api.device_count() < 1:
        pytest.fail("no GPU")
    codes = clustered_codes(n, M, seed=n + 11)
    cb = synth.make_codebook(M, 256, 4, seed=n)
    ref = builder_oracle.build(codes, cb)
    t = api.DeltaTree(codes, codebook=cb, device=0)
    assert np.array_equal(t.edges, ref["edges"].reshape(-1, 2))
    assert np.array_equal(t.vec_id, ref["vec_id"]) and np.array_equal(t.payload(), synth.encode_dtc(ref)[0])


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 5, 30000])
def test_gpu_layout_writes_the_same_artefact_files(lib, tmp_path, n):
    """The tree laid out on the GPU (adjacency, max_dist2p sibling order, DFS numbering, masks, changed bytes,
    sub-tree sizes, sqrt'ed max distances) against the host layout: the three artefact files byte for byte
    (the 60-byte QNode records carry every field of the layout)."""
    import filecmp
    from deltapq_amd import api, synth
    codes = clustered_codes(n, 8, seed=n + 5)
    cb = synth.make_codebook(8, 256, 16, seed=n)
    dh, dd = tmp_path / "host", tmp_path / "dev"
    dh.mkdir()
    dd.mkdir()
    host, dev = api.DeltaTree(codes, codebook=cb), api.DeltaTree(codes, codebook=cb, device=0)
    host.write_files(str(dh))
    dev.write_files(str(dd))
    names = sorted(os.listdir(str(dh)))
    assert len(names) == 3 and names == sorted(os.listdir(str(dd)))
    for f in names:
        assert filecmp.cmp(str(dh / f), str(dd / f), shallow=False), f
    assert host.stats == dev.stats and np.array_equal(host.parent_pos, dev.parent_pos)


@pytest.mark.gpu
@pytest.mark.parametrize("n,M", [(1, 8), (2, 8), (777, 8), (50000, 8), (20000, 16)])
def test_gpu_edge_search_builds_the_identical_tree(lib, n, M):
    """SURVEY.md 8f row 1 on the GPU: the sort/group passes over every position subset run on the device
    (hipCUB stable radix sort + grouping kernels) and must reproduce the host builder exactly."""
    from deltapq_amd import api
    if api.device_count() < 1:
        pytest.fail("no GPU")
    codes = clustered_codes(n, M, seed=n + 1)
    host = api.DeltaTree(codes)
    dev = api.DeltaTree(codes, device=0)
    assert np.array_equal(host.edges, dev.edges)
    assert np.array_equal(host.vec_id, dev.vec_id) and np.array_equal(host.payload(), dev.payload())
    assert host.stats == dev.stats


@pytest.mark.gpu
def test_gpu_edge_search_speed_and_structure_1m(lib):
    import time
    from deltapq_amd import api, synth
    n = 1_000_000
    tree = synth.synth_tree(n, 8, seed=102, mean_diffs=3.0)
    codes = synth.decode_tree_codes(tree)[np.random.default_rng(0).permutation(n)]
    t0 = time.time()
    dev = api.DeltaTree(codes, device=0)
    t_gpu = time.time() - t0
    assert dev.stats["n_diffs"] < 3.4 * n and dev.stats["max_depth"] <= 7
    assert np.array_equal(tree_decode(dev), codes[dev.vec_id])
    print("GPU-assisted build of 1M codes: %.2f s, %.2f diffs/node" % (t_gpu, dev.stats["n_diffs"] / n))
    assert t_gpu < 20


@pytest.mark.parametrize("seed", range(4))
def test_subset_prefilter_rule_keeps_every_member_of_a_group(seed):
    """find_edges_gpu's per-subset pre-filter (dpq_build_gpu.hip, keys_mark_kernel / hash_flag_kernel), restated in numpy: an
    open-addressing table of epoch | pair | fingerprint words, never cleared.  A node walks at most kProbes words from its
    slot: a word of an older epoch is free and is claimed; a word with the node's fingerprint gets its pair bit set; any
    other word is skipped.  A node is kept iff the word with its fingerprint carries the pair bit, or it found no word (then
    neither did any node with its key).  Whatever the order of arrival and whatever older epochs left in the table, every
    member of a group of >= 2 equal keys is kept, and a node alone under its key only by a fingerprint collision; the kept
    nodes in list order sort like the whole list does (the grouping returns on groups of one).  CPU only -- the kernels are
    checked by test_gpu_edge_search_builds_the_identical_tree."""
    rng = np.random.default_rng(seed)
    n, slots, probes = 5000, 1 << 13, 8                       # load ~0.6: probing and the not-placed case on purpose
    ep_of = np.zeros(slots, dtype=np.int64)                   # epoch 0 = never used
    pair = np.zeros(slots, dtype=bool)
    fp_of = np.zeros(slots, dtype=np.uint32)

    def h64(k):
        x = (k * np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        x ^= x >> np.uint64(29)
        x = (x * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
        return x ^ (x >> np.uint64(32))

    for epoch in (1, 2, 7):
        keys = rng.integers(0, 3000, size=n).astype(np.uint64)  # many groups of equal keys, many singletons
        with np.errstate(over="ignore"):
            h = h64(keys)
        fp = (h >> np.uint64(32)).astype(np.uint32)
        s0 = (h & np.uint64(slots - 1)).astype(np.int64)
        for i in rng.permutation(n):                            # CAS / OR semantics, arbitrary arrival order
            for p in range(probes):
                s = (s0[i] + p) & (slots - 1)
                if ep_of[s] != epoch:                           # free: claim
                    ep_of[s], pair[s], fp_of[s] = epoch, False, fp[i]
                    break
                if fp_of[s] == fp[i]:                           # this key is here already
                    pair[s] = True
                    break
        keep = np.ones(n, dtype=bool)                           # no word within `probes` steps: kept
        for i in range(n):
            for p in range(probes):
                s = (s0[i] + p) & (slots - 1)
                if ep_of[s] != epoch:
                    break
                if fp_of[s] == fp[i]:
                    keep[i] = pair[s]
                    break
        uniq, counts = np.unique(keys, return_counts=True)
        in_group = np.isin(keys, uniq[counts >= 2])
        assert keep[in_group].all()                             # no member of a clique is ever dropped
        assert (keep & ~in_group).sum() <= 0.02 * n + 8         # singles: fingerprint collisions / unplaced only
        order_all = np.argsort(keys, kind="stable")
        order_kept = np.flatnonzero(keep)[np.argsort(keys[keep], kind="stable")]
        grp = np.flatnonzero(in_group)
        assert np.array_equal(order_all[np.isin(order_all, grp)], order_kept[np.isin(order_kept, grp)])
