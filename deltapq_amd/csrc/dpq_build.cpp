// dpq_build.cpp -- DeltaTree construction on the host.
//
// Own implementation of the reference's `method 1` builder
// (/root/reference/deltapq_create_approx_tree.h, "h:"):
//   for diff = 0..M (h:1263): for every subset of M-diff kept positions
//   (h:468, 480): group the still-unmerged codes that agree on the kept
//   positions (the reference hashes to 128-bit keys and sorts, h:493-528; here a
//   stable LSD radix sort over the kept bytes), make the tallest member the
//   parent of the group (h:547-599), freeze parents that reach the height cap
//   M*h - 2 as "finalists" (h:572-575); finally hang all finalists under the
//   first one (h:1297-1313).  Siblings are ordered by max_dist2p, descending
//   (h:1396-1426), the tree is numbered in DFS order (h:1156-1183).
// Any tree produced this way is lossless; the exact shape depends on sort tie
// order, which the reference leaves to __gnu_parallel::sort.
#include "dpq_build.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>

#include "../../include/deltapq_amd.h"
#include "dpq_format.h"

namespace dpq {

namespace {

void combinations(int n, int k, std::vector<std::vector<int>>* out) {  // lexicographic (create_tree.h:75-95)
    out->clear();
    std::vector<int> c(k);
    std::iota(c.begin(), c.end(), 0);
    if (k == 0) {
        out->push_back(c);
        return;
    }
    while (true) {
        out->push_back(c);
        int i = k - 1;
        while (i >= 0 && c[i] == n - k + i) --i;
        if (i < 0) break;
        ++c[i];
        for (int j = i + 1; j < k; ++j) c[j] = c[j - 1] + 1;
    }
}

// stable sort of ids by the bytes at `kept` positions; the HIGHEST position is the most significant
// digit, i.e. the order of the masked code read as a little-endian integer (what the GPU edge finder sorts)
void radix_by_positions(const uint8_t* codes, int M, const std::vector<int>& kept, std::vector<uint32_t>* ids,
                        std::vector<uint32_t>* tmp) {
    const size_t n = ids->size();
    tmp->resize(n);
    for (int p = 0; p < (int)kept.size(); ++p) {
        const int pos = kept[p];
        size_t count[257] = {0};
        for (size_t i = 0; i < n; ++i) count[codes[(size_t)(*ids)[i] * M + pos] + 1]++;
        for (int b = 0; b < 256; ++b) count[b + 1] += count[b];
        for (size_t i = 0; i < n; ++i) {
            const uint32_t id = (*ids)[i];
            (*tmp)[count[codes[(size_t)id * M + pos]]++] = id;
        }
        ids->swap(*tmp);
    }
}

inline bool same_key(const uint8_t* codes, int M, const std::vector<int>& kept, uint32_t a, uint32_t b) {
    for (int pos : kept)
        if (codes[(size_t)a * M + pos] != codes[(size_t)b * M + pos]) return false;
    return true;
}

}  // namespace

static int check_build_args(const uint8_t* codes, int64_t n, int M, int K, int max_height_folds, Tree* out,
                            std::string* err) {
    if (!codes || n < 1 || M < 1 || M > 16 || K < 1 || K > 256 || max_height_folds < 1 || !out) {
        if (err) *err = "bad argument to build_tree";
        return DPQ_ERR_ARG;
    }
    if (n >= (int64_t)INT32_MAX) {  // h:982-986
        if (err) *err = "number of codes is too large";
        return DPQ_ERR_ARG;
    }
    return DPQ_OK;
}

void position_subsets(int M, int keep, std::vector<std::vector<int>>* out) { combinations(M, keep, out); }

// ---- edges (find_edges_by_diff_approx, h:1207-1313), host version ----
int find_edges_host(const uint8_t* codes, int64_t n, int M, int max_height_folds, std::vector<uint32_t>* finalists_out,
                    std::vector<std::pair<uint32_t, uint32_t>>* edges, std::string* err) {
    (void)err;
    const int MAXH = M * max_height_folds;  // h:1262
    std::vector<uint32_t> cur((size_t)n), next, act, tmp;
    std::vector<uint32_t>& finalists = *finalists_out;
    finalists.clear();
    std::iota(cur.begin(), cur.end(), 0u);
    std::vector<uint8_t> heights((size_t)n, 0), merged((size_t)n, 0);
    std::vector<std::vector<int>> combos;
    edges->clear();
    edges->reserve((size_t)n);
    for (int diff = 0; diff <= M; ++diff) {
        combinations(M, M - diff, &combos);
        for (const auto& kept : combos) {
            act.clear();
            for (uint32_t id : cur)
                if (!merged[id]) act.push_back(id);
            if (act.size() < 2) break;
            radix_by_positions(codes, M, kept, &act, &tmp);
            for (size_t i = 0; i < act.size();) {
                size_t end = i + 1;
                while (end < act.size() && same_key(codes, M, kept, act[i], act[end])) ++end;
                if (end - i >= 2) {
                    // the tallest member becomes the parent (h:547-558); first one wins ties
                    int max_h = -1, second_h = 0;
                    uint32_t parent = act[i];
                    for (size_t j = i; j < end; ++j)
                        if ((int)heights[act[j]] > max_h) {
                            max_h = heights[act[j]];
                            parent = act[j];
                        }
                    for (size_t j = i; j < end; ++j)
                        if (act[j] != parent && (int)heights[act[j]] > second_h) second_h = heights[act[j]];
                    if (second_h == max_h) heights[parent]++;  // h:569
                    if (max_h + 1 >= MAXH - 2) {                // h:570-575: freeze at the height cap
                        finalists.push_back(parent);
                        merged[parent] = 1;
                    }
                    for (size_t j = i; j < end; ++j)
                        if (act[j] != parent) {
                            merged[act[j]] = 1;
                            edges->emplace_back(parent, act[j]);
                        }
                }
                i = end;
            }
        }
        next.clear();
        for (uint32_t id : cur)
            if (!merged[id]) next.push_back(id);
        cur.swap(next);
        if (cur.size() <= 1) break;  // h:1288
    }
    for (uint32_t id : cur) finalists.push_back(id);  // h:1292-1294
    return DPQ_OK;
}

int build_tree(const uint8_t* codes, int64_t n, int M, int K, int max_height_folds, const float* codewords, int Ds,
               Tree* out, std::string* err) {
    int rc = check_build_args(codes, n, M, K, max_height_folds, out, err);
    if (rc) return rc;
    std::vector<uint32_t> finalists;
    std::vector<std::pair<uint32_t, uint32_t>> edges;
    rc = find_edges_host(codes, n, M, max_height_folds, &finalists, &edges, err);
    if (rc) return rc;
    return layout_tree(codes, n, M, K, max_height_folds, codewords, Ds, finalists, &edges, out, err);
}

// Everything after the edge search: root, adjacency, sibling order, DFS numbering, per-node diffs.
int layout_tree(const uint8_t* codes, int64_t n, int M, int K, int max_height_folds, const float* codewords, int Ds,
                const std::vector<uint32_t>& finalists, std::vector<std::pair<uint32_t, uint32_t>>* edges_in, Tree* out,
                std::string* err) {
    int rc = check_build_args(codes, n, M, K, max_height_folds, out, err);
    if (rc) return rc;
    Tree& t = *out;
    t = Tree();
    t.M = M;
    t.K = K;
    t.max_height_folds = max_height_folds;
    t.n = n;
    t.edges.swap(*edges_in);
    const int levels = levels_for(M);
    if (finalists.empty()) {
        if (err) *err = "internal: no root found";
        return DPQ_ERR_FORMAT;
    }
    t.root_id = finalists[0];
    for (size_t i = 1; i < finalists.size(); ++i) t.edges.emplace_back(t.root_id, finalists[i]);  // h:1297-1313
    if ((int64_t)t.edges.size() != n - 1) {
        if (err) *err = "internal: edge count != n - 1";
        return DPQ_ERR_FORMAT;
    }

    // ---- adjacency in edge order (edges_to_adj_lists_approx, h:1067-1100) ----
    std::vector<uint32_t> parents((size_t)n, 0xffffffffu), offsets((size_t)n + 1, 0), children((size_t)(n - 1));
    for (const auto& e : t.edges) {
        parents[e.second] = e.first;
        offsets[e.first + 1]++;
    }
    for (int64_t i = 0; i < n; ++i) offsets[(size_t)i + 1] += offsets[(size_t)i];
    {
        std::vector<uint32_t> fill(offsets.begin(), offsets.end() - 1);
        for (const auto& e : t.edges) children[fill[e.first]++] = e.second;
    }

    // ---- sibling order by max_dist2p, descending (h:1396-1426); needs the codebook ----
    std::vector<float> max_dists, max_d2p;
    if (codewords && Ds > 0) {
        // centroid-to-centroid tables, main:101-118: `float dist += pow(float - float, 2)`
        std::vector<float> tab((size_t)M * K * K);
        for (int m = 0; m < M; ++m)
            for (int j = 0; j < K; ++j)
                for (int k = 0; k < K; ++k) {
                    float dist = 0;
                    for (int d = 0; d < Ds; ++d) {
                        volatile float df = codewords[((size_t)m * K + j) * Ds + d] - codewords[((size_t)m * K + k) * Ds + d];
                        dist = (float)((double)dist + (double)df * (double)df);
                    }
                    tab[((size_t)m * K + j) * K + k] = dist;
                }
        max_dists.assign((size_t)n, 0.f);
        max_d2p.assign((size_t)n, 0.f);
        for (int64_t vid = 0; vid < n; ++vid) {
            uint32_t parent = parents[(size_t)vid], prev = (uint32_t)vid;
            int depth = 0;
            while (parent != 0xffffffffu) {
                if (depth++ >= 16) break;  // h:1403
                float dist = 0;            // cal_distance_by_tables, h:186-194
                for (int m = 0; m < M; ++m)
                    dist += tab[((size_t)m * K + codes[(size_t)vid * M + m]) * K + codes[(size_t)parent * M + m]];
                if (dist > max_dists[parent]) max_dists[parent] = dist;
                if (dist > max_d2p[prev]) max_d2p[prev] = dist;
                prev = parent;
                parent = parents[parent];
            }
        }
        for (int64_t v = 0; v < n; ++v)
            std::stable_sort(children.begin() + offsets[(size_t)v], children.begin() + offsets[(size_t)v + 1],
                             [&](uint32_t a, uint32_t b) { return max_d2p[a] > max_d2p[b]; });
    }

    // ---- DFS numbering and per-node diffs (dfs_node_layout, h:1156-1183) ----
    t.vec_id.assign((size_t)n, 0);
    t.parent_pos.assign((size_t)n, 0xffffffffu);
    t.subtree.assign((size_t)n, 0);
    t.depth.assign((size_t)n, 0);
    t.mask.assign((size_t)n, 0);
    t.root_code.assign(codes + (size_t)t.root_id * M, codes + (size_t)t.root_id * M + M);
    struct Frame {
        uint32_t id, pos, next_child;
    };
    std::vector<Frame> stack;
    stack.push_back({t.root_id, 0, offsets[t.root_id]});
    t.vec_id[0] = t.root_id;
    uint32_t pos = 0;
    while (!stack.empty()) {
        Frame& f = stack.back();
        if (f.next_child == offsets[f.id + 1]) {
            t.subtree[f.pos] = pos - f.pos;  // h:1182
            stack.pop_back();
            continue;
        }
        const uint32_t child = children[f.next_child++];
        const uint32_t ppos = f.pos, pid = f.id;
        const int depth = (int)stack.size();
        ++pos;
        if (depth >= levels) {
            if (err) *err = "tree is deeper than the DTC depth field allows (lower -h)";
            return DPQ_ERR_FORMAT;
        }
        t.vec_id[pos] = child;
        t.parent_pos[pos] = ppos;
        t.depth[pos] = (uint8_t)depth;
        unsigned mk = 0;
        for (int m = 0; m < M; ++m) {
            const uint8_t from = codes[(size_t)pid * M + m], to = codes[(size_t)child * M + m];
            if (from != to) {
                mk |= 1u << m;
                t.deltas.push_back(to);
                t.delta_from.push_back(from);
            }
        }
        t.mask[pos] = (uint16_t)mk;
        t.n_diffs += __builtin_popcount(mk);
        if (depth > t.max_depth) t.max_depth = depth;
        t.depth_hist[depth]++;
        stack.push_back({child, pos, offsets[child]});
    }
    t.depth_hist[0] = 1;
    if ((int64_t)pos != n - 1) {
        if (err) *err = "internal: DFS did not reach every node";
        return DPQ_ERR_FORMAT;
    }
    if (!max_d2p.empty()) {
        t.max_dist.resize((size_t)n);
        t.max_dist2p.resize((size_t)n);
        for (int64_t p = 0; p < n; ++p) {
            t.max_dist[(size_t)p] = std::sqrt(max_dists[t.vec_id[(size_t)p]]);      // h:1455
            t.max_dist2p[(size_t)p] = std::sqrt(max_d2p[t.vec_id[(size_t)p]]);      // h:1456
        }
    }
    return DPQ_OK;
}

int tree_encode(const Tree& t, std::vector<uint8_t>* payload, std::string* err) {
    int64_t nb = 0;
    int rc = encode(t.root_code.data(), t.depth.data(), t.mask.data(), t.deltas.data(), t.n, t.M, nullptr, &nb, err);
    if (rc) return rc;
    payload->resize((size_t)nb);
    return encode(t.root_code.data(), t.depth.data(), t.mask.data(), t.deltas.data(), t.n, t.M, payload->data(), &nb,
                  err);
}

int tree_write_files(const Tree& t, const std::string& dir, std::string* err) {
    const std::string mk = "/M" + std::to_string(t.M) + "K" + std::to_string(t.K);
    const std::string nn = "_N" + std::to_string(t.n);
    auto open_w = [&](const std::string& p) -> FILE* {
        FILE* f = fopen(p.c_str(), "wb");
        if (!f && err) *err = "cannot open " + p;
        return f;
    };
    {   // edges: root_id, then (parent, child) pairs (h:1326-1327)
        FILE* f = open_w(dir + mk + "H" + std::to_string(t.max_height_folds) + "_Approx_Edges" + nn);
        if (!f) return DPQ_ERR_IO;
        fwrite(&t.root_id, sizeof(uint32_t), 1, f);
        if (!t.edges.empty()) fwrite(t.edges.data(), sizeof(t.edges[0]), t.edges.size(), f);
        fclose(f);
    }
    if (t.M <= 8) {  // QNode records (h:79-101): 60 bytes, N + 1 of them (h:1484)
        FILE* f = open_w(dir + mk + "_Approx_TreeNodesDFS" + nn);
        if (!f) return DPQ_ERR_IO;
        size_t doff = 0;
        for (int64_t p = 0; p <= t.n; ++p) {
            uint8_t rec[60];
            memset(rec, 0, sizeof rec);
            uint32_t u[5] = {0, 0, 0, 0, 1};  // vec_id, parent_pos, child_pos_start, child_num, sub_tree_size
            float fl[3] = {0.f, 0.f, 0.f};    // qdist, max_dist, max_dist2p
            if (p < t.n) {
                u[0] = t.vec_id[(size_t)p];
                u[1] = t.parent_pos[(size_t)p];
                u[2] = (uint32_t)p + 1;
                u[3] = t.subtree[(size_t)p];
                if (!t.max_dist.empty()) {
                    fl[1] = t.max_dist[(size_t)p];
                    fl[2] = t.max_dist2p[(size_t)p];
                }
                int nd = 0;
                if (p == 0) {  // h:1437-1445
                    for (int m = 0; m < t.M; ++m) {
                        rec[34 + 3 * m] = (uint8_t)m;
                        rec[35 + 3 * m] = 0xff;
                        rec[36 + 3 * m] = t.root_code[(size_t)m];
                    }
                    nd = t.M;
                } else {
                    for (int m = 0; m < t.M; ++m)
                        if (t.mask[(size_t)p] & (1u << m)) {
                            rec[34 + 3 * nd] = (uint8_t)m;
                            rec[35 + 3 * nd] = t.delta_from[doff];
                            rec[36 + 3 * nd] = t.deltas[doff];
                            ++doff;
                            ++nd;
                        }
                }
                rec[32] = (uint8_t)nd;
                rec[33] = t.depth[(size_t)p];
            }
            memcpy(rec, u, 20);
            memcpy(rec + 20, fl, 12);
            fwrite(rec, 1, sizeof rec, f);
        }
        fclose(f);
    }
    {   // DTC (h:1839-1842)
        std::vector<uint8_t> payload;
        int rc = tree_encode(t, &payload, err);
        if (rc) return rc;
        FILE* f = open_w(dtc_file_name(dir, t.M, t.K, t.n));
        if (!f) return DPQ_ERR_IO;
        int64_t h[2] = {t.n, (int64_t)payload.size()};
        fwrite(h, sizeof(int64_t), 2, f);
        fwrite(payload.data(), 1, payload.size(), f);
        fclose(f);
    }
    return DPQ_OK;
}

int read_qnode_ids(const std::string& path, int64_t n, std::vector<uint32_t>* ids, std::string* err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        if (err) *err = "cannot open " + path;
        return DPQ_ERR_IO;
    }
    ids->resize((size_t)n);
    uint8_t rec[60];
    for (int64_t p = 0; p < n; ++p) {
        if (fread(rec, 1, sizeof rec, f) != sizeof rec) {
            fclose(f);
            if (err) *err = "short TreeNodesDFS file " + path;
            return DPQ_ERR_IO;
        }
        memcpy(&(*ids)[(size_t)p], rec, 4);
    }
    fclose(f);
    return DPQ_OK;
}

int read_codes_plain(const std::string& path, int M, int64_t* n, std::vector<uint8_t>* codes, std::string* err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        if (err) *err = "cannot open " + path;
        return DPQ_ERR_IO;
    }
    int64_t nn = 0;
    if (fread(&nn, sizeof(int64_t), 1, f) != 1 || nn < 0 || nn > (int64_t)INT32_MAX) {
        fclose(f);
        if (err) *err = "bad header in " + path;
        return DPQ_ERR_FORMAT;
    }
    *n = nn;
    if (codes) {
        codes->resize((size_t)nn * M);
        if (nn && fread(codes->data(), 1, codes->size(), f) != codes->size()) {
            fclose(f);
            if (err) *err = "short read on " + path;
            return DPQ_ERR_IO;
        }
    }
    fclose(f);
    return DPQ_OK;
}

int write_codes_plain(const std::string& path, const uint8_t* codes, int64_t n, int M, std::string* err) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) {
        if (err) *err = "cannot open " + path;
        return DPQ_ERR_IO;
    }
    fwrite(&n, sizeof(int64_t), 1, f);
    if (n) fwrite(codes, 1, (size_t)n * M, f);
    fclose(f);
    return DPQ_OK;
}

}  // namespace dpq
