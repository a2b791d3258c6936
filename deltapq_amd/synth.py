"""Seeded synthetic inputs for the DeltaPQ query path.

No real SIFT data exists offline, so every workload is generated:

* SIFT-shaped codebooks and queries (non-negative, 0..218 value range),
* DeltaTree streams emitted directly as valid (depth, mask, changed-bytes)
  triples in DFS order -- the "streaming DTC synthesiser" of SURVEY.md 8d.4 --
  and serialised in the reference's on-disk format
  (qnodes_to_compressed_codes_opt, /root/reference/deltapq_create_approx_tree.h:1730-1845),
* the reference's file formats around the path: codebook text
  (pq.cpp:267-286), .fvecs/.bvecs (utils.cpp:14-71), codes.bin.plain
  (pq_tree.cpp:1011-1031).

Pure numpy; used by bench.py, the CLI tests and the parity tests.
"""
import itertools
import os

import numpy as np

MAX_DEPTH_M8 = 7       # 3-bit depth field, deltapq_create_approx_tree.h:2883


def make_codebook(M=8, K=256, Ds=16, seed=0):
    """SIFT-shaped codebook [M][K][Ds] fp32: gamma-distributed non-negative
    centres clipped to the SIFT value range."""
    rng = np.random.default_rng(seed)
    cb = rng.gamma(shape=1.2, scale=28.0, size=(M, K, Ds))
    cb = np.clip(cb, 0.0, 218.0)
    return cb.astype(np.float32)


def make_queries(nq, D=128, seed=1, integer=True):
    """nq SIFT-shaped query vectors [nq][D] fp32 (integer-valued like .bvecs /
    SIFT .fvecs when integer=True)."""
    rng = np.random.default_rng(seed)
    q = np.clip(rng.gamma(shape=1.2, scale=28.0, size=(nq, D)), 0.0, 218.0)
    if integer:
        q = np.rint(q)
    return q.astype(np.float32)


def make_clustered_vectors(n, D=128, seed=0, n_clusters=2000, spread=9.0, centre_seed=None):
    """SIFT-shaped vectors: mixture of Gaussians around
    gamma-distributed non-negative centres, rounded and clipped to 0..218.
    Base and query sets share `centre_seed` (same mixture, different draws)."""
    crng = np.random.default_rng(seed if centre_seed is None else centre_seed)
    centres = np.clip(crng.gamma(shape=1.2, scale=28.0, size=(n_clusters, D)), 0, 218)
    rng = np.random.default_rng(seed)
    which = rng.integers(0, n_clusters, size=n)
    v = centres[which] + rng.normal(0.0, spread, size=(n, D))
    return np.clip(np.rint(v), 0, 218).astype(np.float32)


def kmeans_codebook(vectors, M=8, K=256, iters=8, seed=0, sample=20000):
    """Lloyd k-means per sub-space (stand-in for PQ::Learn / cv::kmeans, pq.cpp:139-155,
    which needs OpenCV): float32 codebook [M][K][Ds]."""
    rng = np.random.default_rng(seed)
    v = np.asarray(vectors, dtype=np.float32)
    if len(v) > sample:
        v = v[rng.choice(len(v), sample, replace=False)]
    n, D = v.shape
    Ds = D // M
    cb = np.zeros((M, K, Ds), dtype=np.float32)
    for m in range(M):
        x = v[:, m * Ds:(m + 1) * Ds].astype(np.float64)
        c = x[rng.choice(n, K, replace=n < K)].copy()
        for _ in range(iters):
            d2 = (x * x).sum(1)[:, None] - 2.0 * x @ c.T + (c * c).sum(1)[None, :]
            a = d2.argmin(1)
            for k in range(K):
                sel = a == k
                if sel.any():
                    c[k] = x[sel].mean(0)
        cb[m] = c.astype(np.float32)
    return cb


def _depth_chain(n, rng, max_depth, p_child, p_sibling):
    """DFS depth sequence for nodes 1..n-1: d_i in [1, min(d_{i-1}+1, max_depth)]."""
    if n <= 1:
        return np.zeros(0, dtype=np.uint8)
    u = rng.random(n - 1)
    up = rng.geometric(0.55, size=n - 1)                     # levels to climb when closing a subtree
    steps = np.where(u < p_child, 1, np.where(u < p_child + p_sibling, 0, -up)).astype(np.int64)
    steps[0] = 1                                             # node 1 is a child of the root
    chain = itertools.accumulate(steps.tolist(), lambda d, s: min(max(d + s, 1), max_depth), initial=0)
    out = np.fromiter(chain, dtype=np.int64, count=n)[1:]
    return out.astype(np.uint8)


def synth_tree(n_codes, M=8, seed=0, mean_diffs=3.0, p_child=0.42, p_sibling=0.33, max_depth=None):
    """Random DeltaTree in DFS layout.

    Returns dict(root=u8[M], depths=u8[n] (depths[0] == 0), masks=uint16[n]
    (masks[0] unused), deltas=u8[total popcount of masks[1:]]).  The tree is
    valid by construction: depth(i) >= 1 and depth(i) <= depth(i-1)+1."""
    rng = np.random.default_rng(seed)
    if max_depth is None:
        max_depth = MAX_DEPTH_M8 if M <= 8 else 15
    depths = np.zeros(n_codes, dtype=np.uint8)
    depths[1:] = _depth_chain(n_codes, rng, max_depth, p_child, p_sibling)
    bits = rng.random((n_codes, M)) < (mean_diffs / M)
    weights = (1 << np.arange(M)).astype(np.uint32)
    masks = (bits * weights).sum(axis=1).astype(np.uint16)
    masks[0] = 0
    n_diffs = int(bits[1:].sum())
    deltas = rng.integers(0, 256, size=n_diffs, dtype=np.uint8)
    root = rng.integers(0, 256, size=M, dtype=np.uint8)
    return dict(root=root, depths=depths, masks=masks, deltas=deltas, M=M)


def synth_tree_large(n_codes, M=8, seed=0, mean_diffs=3.0, block=1 << 22, max_depth=None):
    """Like synth_tree for 10^7..10^9 nodes: the DFS depth chain of one block of
    `block` nodes is generated once and repeated (every block starts a new
    depth-1 subtree, which is valid anywhere); masks and changed bytes are fresh
    random data per block, generated in bounded memory."""
    if n_codes <= block:
        return synth_tree(n_codes, M, seed, mean_diffs, max_depth=max_depth)
    rng = np.random.default_rng(seed)
    if max_depth is None:
        max_depth = MAX_DEPTH_M8 if M <= 8 else 15
    chain = _depth_chain(block + 1, rng, max_depth, 0.42, 0.33)          # `block` depths, first one == 1
    depths = np.empty(n_codes, dtype=np.uint8)
    masks = np.empty(n_codes, dtype=np.uint16)
    depths[0] = 0
    masks[0] = 0
    weights = (1 << np.arange(M)).astype(np.uint16)
    pos, n_diffs = 1, 0
    while pos < n_codes:
        m = min(block, n_codes - pos)
        depths[pos:pos + m] = chain[:m]
        bits = rng.random((m, M), dtype=np.float32) < np.float32(mean_diffs / M)
        masks[pos:pos + m] = (bits * weights).sum(axis=1, dtype=np.uint16)
        n_diffs += int(bits.sum())
        pos += m
    deltas = rng.integers(0, 256, size=n_diffs, dtype=np.uint8)
    root = rng.integers(0, 256, size=M, dtype=np.uint8)
    return dict(root=root, depths=depths, masks=masks, deltas=deltas, M=M)


def popcount16(x):
    x = x.astype(np.uint32)
    x = x - ((x >> 1) & 0x5555)
    x = (x & 0x3333) + ((x >> 2) & 0x3333)
    x = (x + (x >> 4)) & 0x0F0F
    return ((x + (x >> 8)) & 0x1F).astype(np.int64)


def encode_dtc(tree):
    """Serialise a tree as the reference DTC payload (M <= 8).

    Layout (deltapq_create_approx_tree.h:1771-1826): root = M raw bytes; then for
    node pairs (i, i+1), i = 1, 3, ...: one byte depth_i | depth_{i+1} << 4,
    node i: mask byte + changed bytes (ascending position), node i+1 likewise;
    a final unpaired node gets a full depth byte.  Returns (payload u8[], n_bytes)
    where n_bytes == M + n_diffs + (3*(N-1)+1)//2 (h:1765)."""
    M = tree["M"]
    # M <= 8 is the reference format (h:1765, 1791-1795).  M in 9..16 is this build's
    # own extension (the reference has none): 2-byte little-endian masks, 4-bit depths.
    mb = 1 if M <= 8 else 2
    depths, masks, deltas, root = tree["depths"], tree["masks"], tree["deltas"], tree["root"]
    n = len(depths)
    assert n >= 1 and np.all(depths[1:] >= 1) and np.all(depths[1:] <= (7 if M <= 8 else 15))
    pc = popcount16(masks)
    pc[0] = 0
    idx = np.arange(n, dtype=np.int64)
    # every odd node i >= 1 is preceded by one depth byte (shared with i+1, or alone)
    n_depth_bytes_before = (idx + 1) // 2                      # odd j <= i
    size = mb + pc                                             # mask byte(s) + changed bytes
    size[0] = 0
    cum = np.cumsum(size) - size                               # bytes of nodes 1..i-1
    mask_off = M + cum + n_depth_bytes_before                  # offset of node i's mask byte
    n_bytes = M + int(pc.sum()) + mb * (n - 1) + n // 2        # == h:1765 for M == 8
    out = np.zeros(n_bytes, dtype=np.uint8)
    out[:M] = root
    if n > 1:
        odd = idx[1::2]
        depth_off = mask_off[odd] - 1
        dbytes = depths[odd].astype(np.uint8)
        has_pair = odd + 1 < n
        dbytes[has_pair] |= (depths[odd[has_pair] + 1].astype(np.uint8) << 4)
        out[depth_off] = dbytes
        out[mask_off[1:]] = (masks[1:] & 0xFF).astype(np.uint8)
        if mb == 2:
            out[mask_off[1:] + 1] = (masks[1:] >> 8).astype(np.uint8)
        # changed bytes follow each node's mask byte(s)
        starts = mask_off[1:] + mb
        reps = pc[1:]
        if reps.sum() > 0:
            base = np.repeat(starts, reps)
            within = np.arange(int(reps.sum()), dtype=np.int64) - np.repeat(np.cumsum(reps) - reps, reps)
            out[base + within] = deltas
        last = mask_off[n - 1] + mb + pc[n - 1]
        assert last == n_bytes, (last, n_bytes)
    return out, n_bytes


def decode_tree_codes(tree):
    """Raw PQ codes [n][M] of every DFS position, by resolving parents level by
    level (used to build plain-code comparators and by tests)."""
    M = tree["M"]
    depths, masks, deltas, root = tree["depths"], tree["masks"], tree["deltas"], tree["root"]
    n = len(depths)
    codes = np.zeros((n, M), dtype=np.uint8)
    codes[0] = root
    pc = popcount16(masks)
    pc[0] = 0
    doff = np.cumsum(pc) - pc
    # parent(i) = nearest preceding node with depth(i)-1
    parent = np.zeros(n, dtype=np.int64)
    last_at = np.full(17, -1, dtype=np.int64)
    last_at[0] = 0
    dl = depths.tolist()
    par = parent.tolist()
    la = last_at.tolist()
    for i in range(1, n):
        d = dl[i]
        par[i] = la[d - 1]
        la[d] = i
    parent = np.asarray(par, dtype=np.int64)
    bit = ((masks[:, None] >> np.arange(M)[None, :]) & 1).astype(bool)
    rank = np.cumsum(bit, axis=1) - bit                         # index of the changed byte inside the node
    for d in range(1, int(depths.max()) + 1 if n > 1 else 1):
        sel = np.flatnonzero(depths == d)
        if sel.size == 0:
            continue
        c = codes[parent[sel]].copy()
        b = bit[sel]
        src = doff[sel][:, None] + rank[sel]
        c[b] = deltas[src[b]]
        codes[sel] = c
    return codes


# ---------------------------------------------------------------------------
# File formats of the reference around the path.
# ---------------------------------------------------------------------------

def dtc_file_name(dataset_dir, M, K, N):
    """h:2812-2814"""
    return os.path.join(dataset_dir, "M%dK%d_Approx_compressed_codes_opt_N%d" % (M, K, N))


def write_dtc_file(path, n_codes, payload):
    """int64 n_codes, int64 n_bytes, payload (h:1839-1842)."""
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(np.asarray([n_codes, payload.size], dtype=np.int64).tobytes())
        f.write(payload.tobytes())


def write_codewords_txt(path, codebook):
    """PQ::WriteCodewords (pq.cpp:267-286): default ostream precision (%g, 6 digits)."""
    cb = np.asarray(codebook, dtype=np.float32)
    M, K, Ds = cb.shape
    with open(path, "w") as f:
        f.write("%d,%d,%d\n" % (M, K, Ds))
        for m in range(M):
            f.write("%d:\n" % m)
            for k in range(K):
                f.write("".join("%g," % v for v in cb[m, k]))
                f.write("\n")


def read_codewords_txt(path):
    """Python mirror of PQ::ReadCodewords (pq.cpp:288-312) for tests/tools."""
    with open(path) as f:
        M, K, Ds = (int(x) for x in f.readline().strip().split(","))
        cb = np.zeros((M, K, Ds), dtype=np.float32)
        for m in range(M):
            assert int(f.readline().strip().rstrip(":")) == m
            for k in range(K):
                vals = f.readline().strip().rstrip(",").split(",")
                cb[m, k] = np.asarray([np.float32(v) for v in vals], dtype=np.float32)
    return cb


def write_fvecs(path, vecs):
    v = np.asarray(vecs, dtype=np.float32)
    n, D = v.shape
    rec = np.empty((n, D + 1), dtype=np.float32)
    rec[:, 0] = np.asarray([D], dtype=np.int32).view(np.float32)[0]
    rec[:, 1:] = v
    rec.tofile(path)


def write_bvecs(path, vecs):
    v = np.asarray(vecs, dtype=np.uint8)
    n, D = v.shape
    rec = np.empty((n, D + 4), dtype=np.uint8)
    rec[:, :4] = np.asarray([D], dtype=np.int32).view(np.uint8)[None, :]
    rec[:, 4:] = v
    rec.tofile(path)


def write_codes_plain(path, codes):
    """PQTree::Write (pq_tree.cpp:1011-1031): int64 N then N*M bytes."""
    c = np.ascontiguousarray(codes, dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(np.asarray([c.shape[0]], dtype=np.int64).tobytes())
        f.write(c.tobytes())


def make_dataset_dir(dataset_dir, n_codes, nq, M=8, K=256, Ds=16, seed=0, ext="fvecs", mean_diffs=3.0):
    """Materialise a reference-style dataset directory: codebook text, query
    file, DTC index file.  Returns (tree, codebook_as_parsed_from_text, queries)."""
    os.makedirs(dataset_dir, exist_ok=True)
    cb = make_codebook(M, K, Ds, seed)
    cb_path = os.path.join(dataset_dir, "M%dK%dcodewords.txt" % (M, K))
    write_codewords_txt(cb_path, cb)
    cb_rt = read_codewords_txt(cb_path)             # what any reader of the text file sees
    queries = make_queries(nq, M * Ds, seed + 1)
    if ext == "bvecs":
        write_bvecs(os.path.join(dataset_dir, "query.bvecs"), queries.astype(np.uint8))
    else:
        write_fvecs(os.path.join(dataset_dir, "query.fvecs"), queries)
    tree = synth_tree(n_codes, M, seed + 2, mean_diffs=mean_diffs)
    payload, _ = encode_dtc(tree)
    write_dtc_file(dtc_file_name(dataset_dir, M, K, n_codes), n_codes, payload)
    return tree, cb_rt, queries
