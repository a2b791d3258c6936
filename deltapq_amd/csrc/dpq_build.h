// dpq_build.h -- DeltaTree construction (host): PQ codes -> tree in DFS layout
// -> DTC stream.  SURVEY.md section 8f row 1; reference:
// /root/reference/deltapq_create_approx_tree.h:445-627 (partition per diff
// level), :1207-1332 (find_edges_by_diff_approx), :1334-1487
// (edges_to_tree_index_approx_dfs_layout), :1156-1183 (dfs_node_layout).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace dpq {

struct Tree {
    int M = 8, K = 256, max_height_folds = 1;
    int64_t n = 0;
    uint32_t root_id = 0;                // original id of the root
    std::vector<uint32_t> vec_id;        // DFS position -> original code id (QNode.vec_id, h:80)
    std::vector<uint32_t> parent_pos;    // DFS position of the parent, 0xffffffff for the root
    std::vector<uint32_t> subtree;       // descendants below each position (QNode.child_num, h:1182)
    std::vector<uint8_t> depth;
    std::vector<uint16_t> mask;          // bit m set <=> position m differs from the parent
    std::vector<uint8_t> deltas;         // changed bytes of nodes 1.., ascending position
    std::vector<uint8_t> delta_from;     // the parent's bytes at those positions (QNode.diffs[].from)
    std::vector<uint8_t> root_code;
    std::vector<float> max_dist, max_dist2p;  // per DFS position (sqrt applied, h:1455-1456); empty without a codebook
    int64_t n_diffs = 0;
    int max_depth = 0;
    int64_t depth_hist[16] = {0};
    std::vector<std::pair<uint32_t, uint32_t>> edges;  // (parent id, child id), n-1 of them
};

// Build the tree of `method 1` (tree with height cap M * max_height_folds).
// codewords ([M][K][Ds], may be NULL) only orders siblings (by max_dist2p, h:1420-1426).
int build_tree(const uint8_t* codes, int64_t n, int M, int K, int max_height_folds, const float* codewords, int Ds,
               Tree* out, std::string* err);
// The two halves of build_tree, shared with the GPU edge finder (dpq_build_gpu.hip):
// the edge search (find_edges_by_diff_approx, h:1207-1313) and everything after it.
int find_edges_host(const uint8_t* codes, int64_t n, int M, int max_height_folds, std::vector<uint32_t>* finalists,
                    std::vector<std::pair<uint32_t, uint32_t>>* edges, std::string* err);
int layout_tree(const uint8_t* codes, int64_t n, int M, int K, int max_height_folds, const float* codewords, int Ds,
                const std::vector<uint32_t>& finalists, std::vector<std::pair<uint32_t, uint32_t>>* edges, Tree* out,
                std::string* err);
void position_subsets(int M, int keep, std::vector<std::vector<int>>* out);  // lexicographic, create_tree.h:75-95
// Edge search on the GPU (same groups, same parents, same edge order as find_edges_host).  prefilter: every position
// subset's sort / group / emit passes run on the nodes whose masked key occurs at least twice (hash_mark_kernel); false:
// on all active nodes, as rounds 2 - 3 did (developer A/B: the same edges either way).
int find_edges_gpu(const uint8_t* codes, int64_t n, int M, int max_height_folds, int device,
                   std::vector<uint32_t>* finalists, std::vector<std::pair<uint32_t, uint32_t>>* edges,
                   std::string* err, bool prefilter = true);
// Everything after the edge search on the GPU: the same Tree as layout_tree, array for array.
int layout_tree_gpu(const uint8_t* codes, int64_t n, int M, int K, int max_height_folds, const float* codewords, int Ds,
                    const std::vector<uint32_t>& finalists, std::vector<std::pair<uint32_t, uint32_t>>* edges, int device,
                    Tree* out, std::string* err);
// DTC payload of the tree (qnodes_to_compressed_codes_opt, h:1765-1826).
int tree_encode(const Tree& t, std::vector<uint8_t>* payload, std::string* err);
// The reference's artefacts in `dir`: M{M}K{K}H{h}_Approx_Edges_N{N} (h:1326-1327),
// M{M}K{K}_Approx_TreeNodesDFS_N{N} (60-byte QNode records, h:1484; M <= 8 only),
// M{M}K{K}_Approx_compressed_codes_opt_N{N} (h:1839-1842).
int tree_write_files(const Tree& t, const std::string& dir, std::string* err);
// DFS position -> original id from a TreeNodesDFS file.
int read_qnode_ids(const std::string& path, int64_t n, std::vector<uint32_t>* ids, std::string* err);
// codes.bin.plain.M{M}K{K}N{N}: int64 N + N*M bytes (pq_tree.cpp:1011-1081).
int read_codes_plain(const std::string& path, int M, int64_t* n, std::vector<uint8_t>* codes, std::string* err);
int write_codes_plain(const std::string& path, const uint8_t* codes, int64_t n, int M, std::string* err);

}  // namespace dpq
