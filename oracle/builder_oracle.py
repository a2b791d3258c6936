"""oracle/builder_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy + plain Python) of the reference's DeltaTree builder, `-task approx_tree` with
`-method 1` ("h:" = /root/reference/deltapq_create_approx_tree.h):

  find_edges            find_edges_by_diff_approx h:1207-1313 around
                        partition_linear_opt_approx_with_constraint h:445-627: for diff = 0..M, for every subset of
                        M - diff kept positions (nchoosek, create_tree.h:75-95: lexicographic), sort the still
                        unmerged codes by the kept bytes (the 128-bit hash of h:493-520 puts position p at bit
                        8 p: the highest position is the most significant), every run of equal keys is a clique:
                        its tallest member becomes the parent (h:547-558), a parent whose height ties the
                        second tallest grows (h:569), a parent reaching M*h - 2 is frozen as a finalist
                        (h:570-575), the others are merged under it (h:577-599); finally every finalist hangs
                        under the first (h:1292-1313).
  layout                edges_to_tree_index_approx_dfs_layout h:1334-1487: adjacency in edge order
                        (h:1067-1104), max_dist / max_dist2p over up to 16 ancestors with the centroid tables of
                        main:101-118 and the fp32 sum of cal_distance_by_tables, siblings by max_dist2p descending
                        (h:1420-1426), DFS numbering (dfs_node_layout h:1156-1183).

PARITY UNPINNED: written from reading the source; the reference cannot be built here (OpenCV, pq.h:8).
Where the reference leaves an order to an unstable sort (__gnu_parallel::sort h:524, std::sort h:1077 and
h:1422), this restatement -- like the product -- takes the STABLE order; any order gives a lossless tree.
The product's builder (dpq_build.cpp host half, dpq_build_gpu.hip) is compared with this, not with itself."""
import itertools

import numpy as np


def find_edges(codes, max_height_folds=1):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n, M = codes.shape
    max_h = M * max_height_folds                       # h:1262
    heights = np.zeros(n, dtype=np.int64)
    merged = np.zeros(n, dtype=bool)
    cur = np.arange(n)
    finalists, edges = [], []
    for diff in range(M + 1):                          # h:1263
        for kept in itertools.combinations(range(M), M - diff):      # lexicographic = nchoosek's prev_permutation order
            act = cur[~merged[cur]]                                    # h:483-488
            if len(act) < 2:
                break
            if kept:
                # stable sort by the kept bytes, highest position most significant (lexsort: last key is primary)
                order = np.lexsort([codes[act, p] for p in kept])
                act = act[order]
                keys = codes[act][:, kept]
                new = np.any(keys[1:] != keys[:-1], axis=1)
            else:
                new = np.zeros(len(act) - 1, dtype=bool)               # diff = M: everything is one clique
            starts = np.flatnonzero(np.concatenate(([True], new)))
            ends = np.concatenate((starts[1:], [len(act)]))
            for s, e in zip(starts.tolist(), ends.tolist()):
                if e - s < 2:
                    continue
                group = act[s:e]
                h = heights[group]
                parent = int(group[int(np.argmax(h))])                 # first of the tallest (h:547-558)
                top = int(h.max())
                others = h[group != parent]
                second = int(others.max()) if len(others) else 0       # h:560-568 (starts from 0)
                if second == top:
                    heights[parent] += 1                               # h:569
                if top + 1 >= max_h - 2:                               # h:570-575
                    finalists.append(parent)
                    merged[parent] = True
                for c in group.tolist():
                    if c != parent:
                        merged[c] = True
                        edges.append((parent, c))
        cur = cur[~merged[cur]]                                        # h:602-607
        if len(cur) <= 1:                                              # h:1288
            break
    finalists.extend(cur.tolist())                                     # h:1292-1294 (at most one node is left)
    return finalists, edges


def centroid_tables(codebook):
    """main:101-118: `float dist += pow(float - float, 2)` per dimension, for every centroid pair of a sub-space."""
    cb = np.asarray(codebook, dtype=np.float32)
    M, K, Ds = cb.shape
    tab = np.zeros((M, K, K), dtype=np.float32)
    for d in range(Ds):
        df = (cb[:, :, None, d] - cb[:, None, :, d]).astype(np.float32)
        tab = (tab.astype(np.float64) + df.astype(np.float64) ** 2).astype(np.float32)
    return tab


def layout(codes, finalists, edges, codebook=None):
    """Returns dict(root_id, vec_id, parent_pos, depth, mask, deltas, root) in DFS order."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n, M = codes.shape
    root = finalists[0]
    edges = list(edges) + [(root, f) for f in finalists[1:]]           # h:1297-1313
    assert len(edges) == n - 1
    parents = np.full(n, -1, dtype=np.int64)
    children = [[] for _ in range(n)]
    for p, c in edges:                                                 # adjacency in edge order (stable sort by parent)
        parents[c] = p
        children[p].append(c)
    if codebook is not None:
        tab = centroid_tables(codebook)
        max_dists = np.zeros(n, dtype=np.float32)
        max_d2p = np.zeros(n, dtype=np.float32)
        vid = np.arange(n)
        prev = vid.copy()
        anc = parents.copy()
        for _ in range(16):                                            # h:1403: at most 16 ancestors
            live = anc >= 0
            if not live.any():
                break
            v, a, pv = vid[live], anc[live], prev[live]
            dist = np.zeros(len(v), dtype=np.float32)
            for m in range(M):                                         # cal_distance_by_tables h:186-194: fp32 sum
                dist = (dist + tab[m, codes[v, m], codes[a, m]]).astype(np.float32)
            np.maximum.at(max_dists, a, dist)
            np.maximum.at(max_d2p, pv, dist)
            vid, prev, anc = v, a, parents[a]
        for p in range(n):                                             # h:1420-1426, stable
            if len(children[p]) > 1:
                children[p].sort(key=lambda c: -float(max_d2p[c]))
    vec_id = np.zeros(n, dtype=np.uint32)
    parent_pos = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    depth = np.zeros(n, dtype=np.uint8)
    mask = np.zeros(n, dtype=np.uint16)
    deltas = []
    vec_id[0] = root
    pos = 0
    stack = [(root, 0, iter(children[root]))]
    while stack:                                                       # dfs_node_layout h:1156-1183
        pid, ppos, it = stack[-1]
        c = next(it, None)
        if c is None:
            stack.pop()
            continue
        pos += 1
        vec_id[pos], parent_pos[pos], depth[pos] = c, ppos, len(stack)
        ch = np.flatnonzero(codes[pid] != codes[c])
        mask[pos] = int(sum(1 << int(m) for m in ch))
        deltas.extend(codes[c, ch].tolist())
        stack.append((c, pos, iter(children[c])))
    assert pos == n - 1
    return dict(root_id=root, vec_id=vec_id, parent_pos=parent_pos, depths=depth, masks=mask,
                deltas=np.array(deltas, dtype=np.uint8), root=codes[root].copy(), M=M)


def build(codes, codebook=None, max_height_folds=1):
    finalists, edges = find_edges(codes, max_height_folds)
    t = layout(codes, finalists, edges, codebook)
    t["edges"] = np.array(edges + [(finalists[0], f) for f in finalists[1:]], dtype=np.uint32).reshape(-1, 2)
    return t
