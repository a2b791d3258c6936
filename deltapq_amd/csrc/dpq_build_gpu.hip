// dpq_build_gpu.hip -- the edge search of the DeltaTree builder on the GPU
// ("hash/sort/group per position-subset is a radix-sort
// workload").  Same algorithm, same groups, same parents and same edge order as
// find_edges_host (dpq_build.cpp), which restates the reference's
// find_edges_by_diff_approx / partition_linear_opt_approx_with_constraint
// (/root/reference/deltapq_create_approx_tree.h:445-627, 1207-1313):
//
//   for diff = 0..M, for every subset of M-diff kept positions:
//     act    = still-unmerged ids of the current list, order kept     (stable compaction)
//     key    = code masked to the kept positions                      (reference: 128-bit hash, h:493-516)
//     sort (key, id) stably                                           (reference: __gnu_parallel::sort, h:524)
//     per group of equal keys with >= 2 members: tallest member (first wins) becomes the
//     parent, height / finalist bookkeeping, the others become its children (h:534-600)
//
// Sorting and scans use hipCUB (rocPRIM); the grouping is hand-written.  The
// tail of the build (adjacency, sibling order, DFS numbering, masks and changed
// bytes) follows below (layout_tree_gpu).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/deltapq_amd.h"
#include "dpq_build.h"

namespace dpq {

namespace {

#define GB_HIP(expr)                                                              \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            if (err) *err = std::string(#expr) + ": " + hipGetErrorString(_e);    \
            return false;                                                         \
        }                                                                         \
    } while (0)

// keys[w][i] = word w of (code of ids[i]) & mask; W = 1 (M <= 8) or 2 (M <= 16)
__global__ void make_keys_kernel(const uint8_t* __restrict__ codes, int M, const uint32_t* __restrict__ ids,
                                 int64_t n, uint64_t mask_lo, uint64_t mask_hi, uint64_t* __restrict__ key_lo,
                                 uint64_t* __restrict__ key_hi) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* c = codes + (size_t)ids[i] * M;
    uint64_t lo = 0, hi = 0;
    for (int m = 0; m < M && m < 8; ++m) lo |= (uint64_t)c[m] << (8 * m);
    for (int m = 8; m < M; ++m) hi |= (uint64_t)c[m] << (8 * (m - 8));
    key_lo[i] = lo & mask_lo;
    if (key_hi) key_hi[i] = hi & mask_hi;
}

__global__ void gather_u64_kernel(const uint64_t* __restrict__ src, const uint32_t* __restrict__ perm, int64_t n,
                                  uint64_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}

__global__ void gather_u32_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ perm, int64_t n,
                                  uint32_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}

__global__ void iota_kernel(uint32_t* p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (uint32_t)i;
}

// One thread per group head walks its group (h:534-600).  Writes, per sorted slot:
// parent_of[i] = parent id for children (0xffffffff otherwise), is_child, is_final.
__global__ void group_kernel(const uint64_t* __restrict__ key_lo, const uint64_t* __restrict__ key_hi,
                             const uint32_t* __restrict__ ids, int64_t n, int max_h_cap, uint8_t* __restrict__ heights,
                             uint8_t* __restrict__ merged, uint32_t* __restrict__ parent_of,
                             uint32_t* __restrict__ is_child, uint32_t* __restrict__ is_final) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t klo = key_lo[i], khi = key_hi ? key_hi[i] : 0;
    if (i > 0 && key_lo[i - 1] == klo && (!key_hi || key_hi[i - 1] == khi)) return;  // not a group head
    int64_t end = i + 1;
    while (end < n && key_lo[end] == klo && (!key_hi || key_hi[end] == khi)) ++end;
    if (end - i < 2) return;
    int max_h = -1, second_h = 0;
    int64_t pj = i;
    for (int64_t j = i; j < end; ++j) {  // tallest member, first wins (h:547-558)
        const int h = heights[ids[j]];
        if (h > max_h) {
            max_h = h;
            pj = j;
        }
    }
    const uint32_t parent = ids[pj];
    for (int64_t j = i; j < end; ++j)
        if (j != pj) {
            const int h = heights[ids[j]];
            if (h > second_h) second_h = h;
        }
    if (second_h == max_h) heights[parent] = (uint8_t)(max_h + 1);  // h:569
    if (max_h + 1 >= max_h_cap) {                                   // h:570-575
        is_final[pj] = 1;
        merged[parent] = 1;
    }
    for (int64_t j = i; j < end; ++j)
        if (j != pj) {
            merged[ids[j]] = 1;
            is_child[j] = 1;
            parent_of[j] = parent;
        }
}

// append edges / finalists at deterministic slots (sorted order), then bump the device counters
__global__ void emit_kernel(const uint32_t* __restrict__ ids, const uint32_t* __restrict__ parent_of,
                            const uint32_t* __restrict__ is_child, const uint32_t* __restrict__ is_final,
                            const uint32_t* __restrict__ child_slot, const uint32_t* __restrict__ final_slot, int64_t n,
                            uint32_t* __restrict__ edges, uint32_t* __restrict__ finalists,
                            const uint32_t* __restrict__ counters) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (is_child[i]) {
        const size_t e = (size_t)counters[0] + child_slot[i];
        edges[2 * e] = parent_of[i];
        edges[2 * e + 1] = ids[i];
    }
    if (is_final[i]) finalists[(size_t)counters[1] + final_slot[i]] = ids[i];
}

__global__ void bump_kernel(const uint32_t* __restrict__ is_child, const uint32_t* __restrict__ is_final,
                            const uint32_t* __restrict__ child_slot, const uint32_t* __restrict__ final_slot, int64_t n,
                            uint32_t* __restrict__ counters) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && n > 0) {
        counters[0] += child_slot[n - 1] + is_child[n - 1];
        counters[1] += final_slot[n - 1] + is_final[n - 1];
    }
}

// Pre-filter of a position subset (round 4): only nodes whose masked key occurs at least TWICE among the active nodes can
// be members of a clique, and at M = 16 they are a few per cent -- the sort / group / emit passes then run on those alone
// (a node alone under its key changes nothing: group_kernel returns on groups of one; the members' relative order, hence the
// stable sort's and the emitted edges' order, is what it was), and a subset in which no key occurs twice ends there.
// An open-addressing table of 64-bit words, epoch << 33 | pair << 32 | fingerprint, never cleared: a word of an older epoch
// is free.  A node claims the first free word of its probe sequence (CAS) or, meeting its own fingerprint, sets the word's
// pair bit; it is kept iff the word with its fingerprint ends with the pair bit set -- or it found no word in kProbes steps
// (then so did every node with its key: all of them are kept).  Equal keys walk the same sequence, so no member of a clique
// is ever dropped; two different keys with one fingerprint and one slot (2^-32) only keep a node the grouping finds alone.
constexpr int kProbes = 8;
__device__ __forceinline__ uint64_t key_hash(uint64_t lo, uint64_t hi) {
    uint64_t h = lo * 0x9E3779B97F4A7C15ull;
    h ^= (hi + 0x7F4A7C159E3779B9ull) * 0xC2B2AE3D27D4EB4Full;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return h;
}

// The subset's masked keys of the unmerged nodes of the diff's list (`cur`, compacted once per diff: a node merged since
// is skipped here, so no per-subset compaction), written for the passes behind and entered into the table.  Thread 0
// also clears the counter of kept nodes that hash_flag_kernel (the next launch) adds to.
__global__ void keys_mark_kernel(const uint8_t* __restrict__ codes, int M, const uint32_t* __restrict__ cur, int64_t n,
                                 const uint8_t* __restrict__ merged, uint64_t mask_lo, uint64_t mask_hi,
                                 uint64_t* __restrict__ key_lo, uint64_t* __restrict__ key_hi, uint32_t epoch,
                                 uint32_t slot_mask, unsigned long long* __restrict__ table, uint32_t* __restrict__ n_kept) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *n_kept = 0;
    if (i >= n) return;
    const uint32_t id = cur[i];
    if (merged[id]) return;
    const uint8_t* c = codes + (size_t)id * M;
    uint64_t lo = 0, hi = 0;
    for (int m = 0; m < M && m < 8; ++m) lo |= (uint64_t)c[m] << (8 * m);
    for (int m = 8; m < M; ++m) hi |= (uint64_t)c[m] << (8 * (m - 8));
    lo &= mask_lo;
    hi &= mask_hi;
    key_lo[i] = lo;
    if (key_hi) key_hi[i] = hi;
    const uint64_t h = key_hash(lo, key_hi ? hi : 0);
    const uint32_t fp = (uint32_t)(h >> 32);
    const unsigned long long mine = ((unsigned long long)epoch << 33) | fp;
    uint32_t s = (uint32_t)h & slot_mask;
    for (int p = 0; p < kProbes; ++p, s = (s + 1) & slot_mask) {
        unsigned long long c0 = __hip_atomic_load(table + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((uint32_t)(c0 >> 33) != epoch) {  // a word of an older epoch is free: claim it
            const unsigned long long old = atomicCAS(table + s, c0, mine);
            if (old == c0) return;
            c0 = old;  // somebody else took it in the meantime: look at what is there now
        }
        if ((uint32_t)c0 == fp) {  // this key is here already: a pair
            atomicOr(table + s, 1ull << 32);
            return;
        }
    }
}

// flags[i] = the node is kept; the kept nodes' positions in `cur` are also appended (in any order: what sorts them
// breaks ties by position) to `kept` while they number at most `list_cap` -- the short list of small_subset_kernel.
__global__ void hash_flag_kernel(const uint32_t* __restrict__ cur, int64_t n, const uint8_t* __restrict__ merged,
                                 const uint64_t* __restrict__ key_lo, const uint64_t* __restrict__ key_hi, uint32_t epoch,
                                 uint32_t slot_mask, const unsigned long long* __restrict__ table,
                                 uint8_t* __restrict__ flags, uint32_t* __restrict__ kept, uint32_t* __restrict__ n_kept,
                                 uint32_t list_cap) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t keep = 0;
    if (!merged[cur[i]]) {
        const uint64_t h = key_hash(key_lo[i], key_hi ? key_hi[i] : 0);
        const uint32_t fp = (uint32_t)(h >> 32);
        uint32_t s = (uint32_t)h & slot_mask;
        keep = 1;  // no word within kProbes steps: kept (and so is every node with this key)
        for (int p = 0; p < kProbes; ++p, s = (s + 1) & slot_mask) {
            const unsigned long long c0 = table[s];
            if ((uint32_t)(c0 >> 33) != epoch) break;  // (cannot happen before the node's own word; kept)
            if ((uint32_t)c0 == fp) {
                keep = (uint8_t)((c0 >> 32) & 1ull);
                break;
            }
        }
        if (keep) {
            const uint32_t slot = atomicAdd(n_kept, 1u);
            if (slot < list_cap) kept[slot] = (uint32_t)i;
        }
    }
    flags[i] = keep;
}

// A position subset whose pre-filter kept at most kSmallMax nodes, in ONE launch of one block: what the radix sorts, the
// gathers, group_kernel, the two scans, emit_kernel and bump_kernel do for a long list (about forty launches).  The kept
// nodes (pos[] = their positions in the diff's list `act`, in any order) are sorted by (key_hi, key_lo, position) -- the
// order of the two stable radix passes over the list -- by a bitonic network in LDS; a thread per group head walks its group exactly as group_kernel does; the
// children / finalists are appended in sorted order through block-wide prefix sums.
constexpr int kSmallMax = 2048;
constexpr int kSmallThreads = 1024;
__global__ __launch_bounds__(kSmallThreads) void small_subset_kernel(
    const uint64_t* __restrict__ key_lo, const uint64_t* __restrict__ key_hi, const uint32_t* __restrict__ act,
    const uint32_t* __restrict__ pos, int n, int max_h_cap, uint8_t* __restrict__ heights, uint8_t* __restrict__ merged,
    uint32_t* __restrict__ edges, uint32_t* __restrict__ finalists, uint32_t* __restrict__ counters) {
    __shared__ uint64_t s_lo[kSmallMax], s_hi[kSmallMax];
    __shared__ uint32_t s_id[kSmallMax], s_par[kSmallMax];
    __shared__ uint32_t s_ix[kSmallMax];  // position in the diff's list: the ties' order (the list arrives in any order)
    __shared__ uint8_t s_child[kSmallMax], s_final[kSmallMax];
    __shared__ uint32_t s_wave[2][kSmallThreads / 64];
    const int tid = threadIdx.x;
    const uint32_t base_e = counters[0], base_f = counters[1];  // read before anybody adds to them (thread 0, at the end)
    int P = 2;
    while (P < n) P <<= 1;
    for (int i = tid; i < P; i += kSmallThreads) {
        if (i < n) {
            const uint32_t p = pos[i];
            s_lo[i] = key_lo[p];
            s_hi[i] = key_hi ? key_hi[p] : 0ull;
            s_id[i] = act[p];
        } else {  // padding sorts behind every real entry (its index is larger than any real one)
            s_lo[i] = ~0ull;
            s_hi[i] = ~0ull;
            s_id[i] = 0xffffffffu;
        }
        s_ix[i] = i < n ? pos[i] : 0xffffffffu;
        s_child[i] = 0;
        s_final[i] = 0;
        s_par[i] = 0xffffffffu;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += kSmallThreads) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // t-th compare-exchange of the stage: bit j of i clear
                const int ixj = i | j;
                const bool up = (i & k) == 0;
                const uint64_t ah = s_hi[i], bh = s_hi[ixj], al = s_lo[i], bl = s_lo[ixj];
                const uint32_t ai = s_ix[i], bi = s_ix[ixj];
                // (real entries are distinct by position; padding entries compare equal among themselves: either order)
                const bool a_gt_b = ah != bh ? ah > bh : al != bl ? al > bl : ai > bi;
                if (a_gt_b == up) {
                    s_hi[i] = bh, s_hi[ixj] = ah;
                    s_lo[i] = bl, s_lo[ixj] = al;
                    s_ix[i] = bi, s_ix[ixj] = ai;
                    const uint32_t x = s_id[i];
                    s_id[i] = s_id[ixj], s_id[ixj] = x;
                }
            }
            __syncthreads();
        }
    }
    // group heads walk their groups (group_kernel, h:534-600)
    for (int i = tid; i < n; i += kSmallThreads) {
        const uint64_t klo = s_lo[i], khi = s_hi[i];
        if (i > 0 && s_lo[i - 1] == klo && s_hi[i - 1] == khi) continue;  // not a group head
        int end = i + 1;
        while (end < n && s_lo[end] == klo && s_hi[end] == khi) ++end;
        if (end - i < 2) continue;
        int max_h = -1, second_h = 0, pj = i;
        for (int j = i; j < end; ++j) {  // tallest member, first wins (h:547-558)
            const int h = heights[s_id[j]];
            if (h > max_h) {
                max_h = h;
                pj = j;
            }
        }
        const uint32_t parent = s_id[pj];
        for (int j = i; j < end; ++j)
            if (j != pj) {
                const int h = heights[s_id[j]];
                if (h > second_h) second_h = h;
            }
        if (second_h == max_h) heights[parent] = (uint8_t)(max_h + 1);  // h:569
        if (max_h + 1 >= max_h_cap) {                                   // h:570-575
            s_final[pj] = 1;
            merged[parent] = 1;
        }
        for (int j = i; j < end; ++j)
            if (j != pj) {
                merged[s_id[j]] = 1;
                s_child[j] = 1;
                s_par[j] = parent;
            }
    }
    __syncthreads();
    // slots of the children / finalists in sorted order: thread t owns entries 2 t, 2 t + 1
    const int e0 = 2 * tid, e1 = 2 * tid + 1;
    const uint32_t c0 = e0 < n ? s_child[e0] : 0u, c1 = e1 < n ? s_child[e1] : 0u;
    const uint32_t f0 = e0 < n ? s_final[e0] : 0u, f1 = e1 < n ? s_final[e1] : 0u;
    uint32_t ci = c0 + c1, fi = f0 + f1;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t cu = (uint32_t)__shfl_up((int)ci, off, 64), fu = (uint32_t)__shfl_up((int)fi, off, 64);
        if ((tid & 63) >= off) ci += cu, fi += fu;
    }
    if ((tid & 63) == 63) {
        s_wave[0][tid >> 6] = ci;
        s_wave[1][tid >> 6] = fi;
    }
    __syncthreads();
    uint32_t cb = 0, fb = 0, ct = 0, ft = 0;
    for (int w = 0; w < kSmallThreads / 64; ++w) {
        const uint32_t cw = s_wave[0][w], fw = s_wave[1][w];
        cb += w < (tid >> 6) ? cw : 0u, fb += w < (tid >> 6) ? fw : 0u;
        ct += cw, ft += fw;
    }
    uint32_t cs = cb + ci - (c0 + c1), fs = fb + fi - (f0 + f1);  // exclusive prefixes at entry e0
    if (c0) {
        const size_t e = (size_t)base_e + cs;
        edges[2 * e] = s_par[e0], edges[2 * e + 1] = s_id[e0];
    }
    cs += c0;
    if (c1) {
        const size_t e = (size_t)base_e + cs;
        edges[2 * e] = s_par[e1], edges[2 * e + 1] = s_id[e1];
    }
    if (f0) finalists[(size_t)base_f + fs] = s_id[e0];
    fs += f0;
    if (f1) finalists[(size_t)base_f + fs] = s_id[e1];
    if (tid == 0) {
        counters[0] = base_e + ct;
        counters[1] = base_f + ft;
    }
}

__global__ void unmerged_flags_kernel(const uint32_t* __restrict__ ids, const uint8_t* __restrict__ merged, int64_t n,
                                      uint8_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = merged[ids[i]] ? 0 : 1;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

int find_edges_gpu(const uint8_t* codes, int64_t n, int M, int max_height_folds, int device,
                   std::vector<uint32_t>* finalists_out, std::vector<std::pair<uint32_t, uint32_t>>* edges_out,
                   std::string* err, bool prefilter) {
    if (!codes || n < 1 || M < 1 || M > 16 || max_height_folds < 1) {
        if (err) *err = "bad argument to find_edges_gpu";
        return DPQ_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        if (err) *err = "no usable GPU for the edge search";
        return DPQ_ERR_NO_DEVICE;
    }
    const bool wide = M > 8;
    const int MAXH = M * max_height_folds;
    uint8_t *d_codes = nullptr, *d_heights = nullptr, *d_merged = nullptr, *d_flags = nullptr;
    uint32_t *d_child = nullptr, *d_final = nullptr;
    uint32_t *d_cur = nullptr, *d_act = nullptr, *d_ids_a = nullptr, *d_ids_b = nullptr, *d_parent = nullptr,
             *d_cslot = nullptr, *d_fslot = nullptr, *d_edges = nullptr, *d_finalists = nullptr, *d_counters = nullptr,
             *d_num = nullptr, *d_perm_a = nullptr, *d_perm_b = nullptr;
    uint64_t *d_klo_a = nullptr, *d_klo_b = nullptr, *d_khi_a = nullptr, *d_khi_b = nullptr;
    unsigned long long* d_table = nullptr;  // the subsets' pre-filter (keys_mark_kernel)
    uint32_t *d_iota = nullptr, *d_pos = nullptr;
    uint32_t slot_mask = 0, epoch = 0;
    void* d_temp = nullptr;
    size_t temp_bytes = 0;
    int64_t n_cur = n;
    uint32_t h_counters[2] = {0, 0};
    std::vector<std::vector<int>> combos;
    const size_t N = (size_t)n;

    auto run = [&]() -> bool {
    GB_HIP(hipSetDevice(device));
    GB_HIP(hipMalloc(&d_codes, N * M));
    GB_HIP(hipMemcpy(d_codes, codes, N * M, hipMemcpyHostToDevice));
    GB_HIP(hipMalloc(&d_heights, N));
    GB_HIP(hipMalloc(&d_merged, N));
    GB_HIP(hipMemset(d_heights, 0, N));
    GB_HIP(hipMemset(d_merged, 0, N));
    GB_HIP(hipMalloc(&d_flags, N));
    GB_HIP(hipMalloc(&d_child, N * 4));
    GB_HIP(hipMalloc(&d_final, N * 4));
    GB_HIP(hipMalloc(&d_cur, N * 4));
    GB_HIP(hipMalloc(&d_act, N * 4));
    GB_HIP(hipMalloc(&d_ids_a, N * 4));
    GB_HIP(hipMalloc(&d_ids_b, N * 4));
    GB_HIP(hipMalloc(&d_parent, N * 4));
    GB_HIP(hipMalloc(&d_cslot, N * 4));
    GB_HIP(hipMalloc(&d_fslot, N * 4));
    GB_HIP(hipMalloc(&d_edges, N * 8));
    GB_HIP(hipMalloc(&d_finalists, N * 4));
    GB_HIP(hipMalloc(&d_counters, 8));
    GB_HIP(hipMemset(d_counters, 0, 8));
    GB_HIP(hipMalloc(&d_num, 8));
    GB_HIP(hipMalloc(&d_klo_a, N * 8));
    GB_HIP(hipMalloc(&d_klo_b, N * 8));
    if (wide) {
        GB_HIP(hipMalloc(&d_khi_a, N * 8));
        GB_HIP(hipMalloc(&d_khi_b, N * 8));
        GB_HIP(hipMalloc(&d_perm_a, N * 4));
        GB_HIP(hipMalloc(&d_perm_b, N * 4));
    }
    {   // one temp buffer large enough for every hipCUB call below
        size_t a = 0, b = 0, c = 0;
        hipcub::DeviceRadixSort::SortPairs(nullptr, a, d_klo_a, d_klo_b, d_ids_a, d_ids_b, (int)n);
        hipcub::DeviceSelect::Flagged(nullptr, b, d_cur, d_flags, d_act, d_num, (int)n);
        hipcub::DeviceScan::ExclusiveSum(nullptr, c, d_child, d_cslot, (int)n);
        temp_bytes = std::max(a, std::max(b, c)) + 256;
        GB_HIP(hipMalloc(&d_temp, temp_bytes));
    }
    hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(n)), dim3(256), 0, 0, d_cur, n);
    if (prefilter) {
        size_t slots = 1u << 16;
        while (slots < 4 * N && slots < ((size_t)1 << 27)) slots <<= 1;  // load <= 1/4 (probing resolves the collisions); 1 GB at most
        slot_mask = (uint32_t)(slots - 1);
        GB_HIP(hipMalloc(&d_table, slots * 8));
        GB_HIP(hipMemset(d_table, 0, slots * 8));  // epoch 0 is never used
        GB_HIP(hipMalloc(&d_iota, N * 4));
        GB_HIP(hipMalloc(&d_pos, N * 4));
        hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(n)), dim3(256), 0, 0, d_iota, n);
    }

    for (int diff = 0; diff <= M; ++diff) {
        position_subsets(M, M - diff, &combos);
        for (const auto& kept : combos) {
            size_t tb = temp_bytes;
            uint64_t mlo = 0, mhi = 0;
            for (int pos : kept) {
                if (pos < 8) mlo |= 0xffull << (8 * pos);
                else mhi |= 0xffull << (8 * (pos - 8));
            }
            uint32_t n_act = 0;
            if (prefilter) {
                // keys of the unmerged nodes of the diff's list + the pair table; then who is kept (a count and a short list)
                ++epoch;  // < 2^31: at most 2^16 subsets per diff, M + 1 diffs
                hipLaunchKernelGGL(keys_mark_kernel, dim3(blocks_for(n_cur)), dim3(256), 0, 0, d_codes, M, d_cur, n_cur,
                                   d_merged, mlo, mhi, d_klo_a, wide ? d_khi_a : nullptr, epoch, slot_mask, d_table, d_num);
                hipLaunchKernelGGL(hash_flag_kernel, dim3(blocks_for(n_cur)), dim3(256), 0, 0, d_cur, n_cur, d_merged,
                                   d_klo_a, wide ? d_khi_a : nullptr, epoch, slot_mask, d_table, d_flags, d_pos, d_num,
                                   (uint32_t)kSmallMax);
                uint32_t n_keep = 0;
                GB_HIP(hipMemcpy(&n_keep, d_num, 4, hipMemcpyDeviceToHost));
                // (h:1288 stops a diff's subsets once fewer than two nodes are unmerged: such subsets keep nothing here)
                if (n_keep < 2) continue;  // no two unmerged nodes agree on the kept positions: nothing to merge
                if (n_keep <= (uint32_t)kSmallMax) {  // a short list: sort, group and emit in one launch
                    hipLaunchKernelGGL(small_subset_kernel, dim3(1), dim3(kSmallThreads), 0, 0, d_klo_a,
                                       wide ? d_khi_a : nullptr, d_cur, d_pos, (int)n_keep, MAXH - 2, d_heights, d_merged,
                                       d_edges, d_finalists, d_counters);
                    continue;
                }
                // a long list: the kept nodes, in the list's order, are what the passes below sort
                GB_HIP(hipcub::DeviceSelect::Flagged(d_temp, tb, d_iota, d_flags, d_pos, d_num, (int)n_cur));
                hipLaunchKernelGGL(gather_u32_kernel, dim3(blocks_for(n_keep)), dim3(256), 0, 0, d_cur, d_pos,
                                   (int64_t)n_keep, d_act);
                hipLaunchKernelGGL(gather_u64_kernel, dim3(blocks_for(n_keep)), dim3(256), 0, 0, d_klo_a, d_pos,
                                   (int64_t)n_keep, d_klo_b);
                GB_HIP(hipMemcpyAsync(d_klo_a, d_klo_b, (size_t)n_keep * 8, hipMemcpyDeviceToDevice, 0));
                if (wide) {
                    hipLaunchKernelGGL(gather_u64_kernel, dim3(blocks_for(n_keep)), dim3(256), 0, 0, d_khi_a, d_pos,
                                       (int64_t)n_keep, d_khi_b);
                    GB_HIP(hipMemcpyAsync(d_khi_a, d_khi_b, (size_t)n_keep * 8, hipMemcpyDeviceToDevice, 0));
                }
                n_act = n_keep;
            } else {
                // act = unmerged ids of cur, order kept
                hipLaunchKernelGGL(unmerged_flags_kernel, dim3(blocks_for(n_cur)), dim3(256), 0, 0, d_cur, d_merged, n_cur,
                                   d_flags);
                GB_HIP(hipcub::DeviceSelect::Flagged(d_temp, tb, d_cur, d_flags, d_act, d_num, (int)n_cur));
                GB_HIP(hipMemcpy(&n_act, d_num, 4, hipMemcpyDeviceToHost));
                if (n_act < 2) break;
                hipLaunchKernelGGL(make_keys_kernel, dim3(blocks_for(n_act)), dim3(256), 0, 0, d_codes, M, d_act,
                                   (int64_t)n_act, mlo, mhi, d_klo_a, wide ? d_khi_a : nullptr);
            }
            const uint64_t *s_lo = nullptr, *s_hi = nullptr;
            const uint32_t* s_ids = nullptr;
            if (!wide) {
                tb = temp_bytes;
                GB_HIP(hipcub::DeviceRadixSort::SortPairs(d_temp, tb, d_klo_a, d_klo_b, d_act, d_ids_a, (int)n_act));
                s_lo = d_klo_b;
                s_ids = d_ids_a;
            } else {
                // 128-bit key: stable LSD over the two words, carrying a permutation
                hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(n_act)), dim3(256), 0, 0, d_perm_a, (int64_t)n_act);
                tb = temp_bytes;
                GB_HIP(hipcub::DeviceRadixSort::SortPairs(d_temp, tb, d_klo_a, d_klo_b, d_perm_a, d_perm_b, (int)n_act));
                hipLaunchKernelGGL(gather_u64_kernel, dim3(blocks_for(n_act)), dim3(256), 0, 0, d_khi_a, d_perm_b,
                                   (int64_t)n_act, d_khi_b);                      // hi words in lo-sorted order
                tb = temp_bytes;
                GB_HIP(hipcub::DeviceRadixSort::SortPairs(d_temp, tb, d_khi_b, d_khi_a, d_perm_b, d_perm_a, (int)n_act));
                // d_perm_a = final permutation (indices into act); d_khi_a = sorted hi words
                hipLaunchKernelGGL(gather_u64_kernel, dim3(blocks_for(n_act)), dim3(256), 0, 0, d_klo_a, d_perm_a,
                                   (int64_t)n_act, d_klo_b);
                hipLaunchKernelGGL(gather_u32_kernel, dim3(blocks_for(n_act)), dim3(256), 0, 0, d_act, d_perm_a,
                                   (int64_t)n_act, d_ids_a);                      // ids in final order
                s_lo = d_klo_b;
                s_hi = d_khi_a;
                s_ids = d_ids_a;
            }
            GB_HIP(hipMemsetAsync(d_child, 0, (size_t)n_act * 4, 0));
            GB_HIP(hipMemsetAsync(d_final, 0, (size_t)n_act * 4, 0));
            hipLaunchKernelGGL(group_kernel, dim3(blocks_for(n_act)), dim3(256), 0, 0, s_lo, s_hi, s_ids,
                               (int64_t)n_act, MAXH - 2, d_heights, d_merged, d_parent, d_child, d_final);
            tb = temp_bytes;
            GB_HIP(hipcub::DeviceScan::ExclusiveSum(d_temp, tb, d_child, d_cslot, (int)n_act));
            tb = temp_bytes;
            GB_HIP(hipcub::DeviceScan::ExclusiveSum(d_temp, tb, d_final, d_fslot, (int)n_act));
            hipLaunchKernelGGL(emit_kernel, dim3(blocks_for(n_act)), dim3(256), 0, 0, s_ids, d_parent, d_child, d_final,
                               d_cslot, d_fslot, (int64_t)n_act, d_edges, d_finalists, d_counters);
            hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, 0, d_child, d_final, d_cslot, d_fslot, (int64_t)n_act,
                               d_counters);
        }
        // cur = unmerged of cur (h:610-615)
        hipLaunchKernelGGL(unmerged_flags_kernel, dim3(blocks_for(n_cur)), dim3(256), 0, 0, d_cur, d_merged, n_cur,
                           d_flags);
        size_t tb = temp_bytes;
        GB_HIP(hipcub::DeviceSelect::Flagged(d_temp, tb, d_cur, d_flags, d_act, d_num, (int)n_cur));
        uint32_t n_next = 0;
        GB_HIP(hipMemcpy(&n_next, d_num, 4, hipMemcpyDeviceToHost));
        std::swap(d_cur, d_act);
        n_cur = n_next;
        if (n_cur <= 1) break;  // h:1288
    }
    GB_HIP(hipDeviceSynchronize());
    GB_HIP(hipMemcpy(h_counters, d_counters, 8, hipMemcpyDeviceToHost));
    {
        edges_out->resize(h_counters[0]);
        if (h_counters[0])
            GB_HIP(hipMemcpy(edges_out->data(), d_edges, (size_t)h_counters[0] * 8, hipMemcpyDeviceToHost));
        finalists_out->resize((size_t)h_counters[1] + (size_t)n_cur);
        if (h_counters[1])
            GB_HIP(hipMemcpy(finalists_out->data(), d_finalists, (size_t)h_counters[1] * 4, hipMemcpyDeviceToHost));
        if (n_cur)  // h:1292-1294: what is left joins the finalists
            GB_HIP(hipMemcpy(finalists_out->data() + h_counters[1], d_cur, (size_t)n_cur * 4, hipMemcpyDeviceToHost));
    }
    return true;
    };
    const bool ok = run();
    hipFree(d_codes); hipFree(d_heights); hipFree(d_merged); hipFree(d_flags); hipFree(d_child); hipFree(d_final);
    hipFree(d_cur); hipFree(d_act); hipFree(d_ids_a); hipFree(d_ids_b); hipFree(d_parent); hipFree(d_cslot);
    hipFree(d_fslot); hipFree(d_edges); hipFree(d_finalists); hipFree(d_counters); hipFree(d_num); hipFree(d_perm_a);
    hipFree(d_perm_b); hipFree(d_klo_a); hipFree(d_klo_b); hipFree(d_khi_a); hipFree(d_khi_b); hipFree(d_temp);
    hipFree(d_table); hipFree(d_iota); hipFree(d_pos);
    return ok ? DPQ_OK : DPQ_ERR_HIP;
}


// ===========================================================================
// The rest of the build on the GPU (edges_to_tree_index_approx_dfs_layout,
// h:1334-1487, and dfs_node_layout, h:1156-1183): adjacency, max_dist /
// max_dist2p, sibling order, DFS numbering, per-node masks and changed bytes.
// Same result as layout_tree (dpq_build.cpp), array for array.
//
//   parents[child] = parent                                   scatter over the edges
//   adjacency in edge order                                   stable radix sort of the edges by parent (h:1077)
//   centroid tables (main:101-118)                            one thread per (m, j, k)
//   max_dist / max_dist2p over <= 16 ancestors (h:1398-1418)  one thread per node, float max as uint atomicMax
//   siblings by max_dist2p, descending, ties in edge order    two stable sorts: by ~bits(max_dist2p), then by parent
//   depth                                                     level by level (a DeltaTree is at most M h levels deep)
//   subtree sizes                                             levels bottom-up, atomicAdd into the parent
//   DFS position                                              pos(child) = pos(parent) + 1 + sizes of the siblings
//                                                             before it (a scan over the sibling-ordered list),
//                                                             levels top-down
//   masks / changed bytes                                     popcounts in DFS order -> exclusive scan -> scatter
// ===========================================================================
namespace {

__global__ void centroid_table_kernel(const float* __restrict__ cw, int M, int K, int Ds, float* __restrict__ tab) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)M * K * K) return;
    const int k = (int)(t % K), j = (int)((t / K) % K), m = (int)(t / ((int64_t)K * K));
    float dist = 0.0f;  // main:101-118: `float dist += pow(float - float, 2)`
    for (int d = 0; d < Ds; ++d) {
        const float df = __fsub_rn(cw[((size_t)m * K + j) * Ds + d], cw[((size_t)m * K + k) * Ds + d]);
        dist = (float)__dadd_rn((double)dist, __dmul_rn((double)df, (double)df));
    }
    tab[t] = dist;
}

__global__ void set_parents_kernel(const uint32_t* __restrict__ edges, int64_t n_edges, uint32_t* __restrict__ parents,
                                   uint32_t* __restrict__ e_parent, uint32_t* __restrict__ e_child) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_edges) return;
    const uint32_t p = edges[2 * i], c = edges[2 * i + 1];
    parents[c] = p;
    e_parent[i] = p;
    e_child[i] = c;
}

__global__ void ancestor_dist_kernel(const uint8_t* __restrict__ codes, int M, int K, const uint32_t* __restrict__ parents,
                                     const float* __restrict__ tab, int64_t n, uint32_t* __restrict__ max_dists,
                                     uint32_t* __restrict__ max_d2p) {
    const int64_t vid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vid >= n) return;
    uint32_t parent = parents[vid], prev = (uint32_t)vid;
    for (int depth = 0; depth < 16 && parent != 0xffffffffu; ++depth) {  // h:1403
        float dist = 0.0f;                                               // cal_distance_by_tables, h:186-194: fp32 sum
        for (int m = 0; m < M; ++m)
            dist = __fadd_rn(dist, tab[((size_t)m * K + codes[(size_t)vid * M + m]) * K + codes[(size_t)parent * M + m]]);
        const uint32_t bits = __float_as_uint(dist);  // distances are >= 0: uint order = float order
        atomicMax(&max_dists[parent], bits);
        atomicMax(&max_d2p[prev], bits);
        prev = parent;
        parent = parents[parent];
    }
}

__global__ void sibling_key_kernel(const uint32_t* __restrict__ e_child, const uint32_t* __restrict__ max_d2p, int64_t n_edges,
                                   uint32_t* __restrict__ key) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_edges) key[i] = ~max_d2p[e_child[i]];  // ascending ~bits = descending distance
}

__global__ void count_children_kernel(const uint32_t* __restrict__ e_parent, int64_t n_edges, uint32_t* __restrict__ n_children) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_edges) atomicAdd(&n_children[e_parent[i]], 1u);
}

// depth[v] = level for the children of nodes at level - 1; counts how many were set
__global__ void depth_level_kernel(const uint32_t* __restrict__ parents, int64_t n, int level, uint8_t* __restrict__ depth,
                                   uint32_t* __restrict__ n_set) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint32_t p = parents[v];
    if (p != 0xffffffffu && depth[v] == 0xff && depth[p] == (uint8_t)(level - 1)) {
        depth[v] = (uint8_t)level;
        atomicAdd(n_set, 1u);
    }
}

__global__ void subtree_level_kernel(const uint32_t* __restrict__ parents, const uint8_t* __restrict__ depth, int64_t n,
                                     int level, uint32_t* __restrict__ size) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n && depth[v] == (uint8_t)level) atomicAdd(&size[parents[v]], size[v]);
}

__global__ void gather_sizes_kernel(const uint32_t* __restrict__ s_child, const uint32_t* __restrict__ size, int64_t n_edges,
                                    uint32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_edges) out[i] = size[s_child[i]];
}

// pos(child) = pos(parent) + 1 + (sizes of the siblings before it) for children at `level`
__global__ void position_level_kernel(const uint32_t* __restrict__ s_parent, const uint32_t* __restrict__ s_child,
                                      const uint32_t* __restrict__ size_scan, const uint32_t* __restrict__ offsets,
                                      const uint8_t* __restrict__ depth, int64_t n_edges, int level,
                                      uint32_t* __restrict__ pos) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_edges) return;
    const uint32_t c = s_child[i];
    if (depth[c] != (uint8_t)level) return;
    const uint32_t p = s_parent[i];
    pos[c] = pos[p] + 1 + (size_scan[i] - size_scan[offsets[p]]);
}

// per node, at its DFS position: original id, parent position, depth, descendants, mask, popcount, sqrt'ed distances
__global__ void node_records_kernel(const uint8_t* __restrict__ codes, int M, const uint32_t* __restrict__ parents,
                                    const uint32_t* __restrict__ pos, const uint8_t* __restrict__ depth,
                                    const uint32_t* __restrict__ size, const uint32_t* __restrict__ max_dists,
                                    const uint32_t* __restrict__ max_d2p, int64_t n, uint32_t* __restrict__ vec_id,
                                    uint32_t* __restrict__ parent_pos, uint8_t* __restrict__ depth_out,
                                    uint32_t* __restrict__ subtree, uint16_t* __restrict__ mask,
                                    uint32_t* __restrict__ n_changed, float* __restrict__ md, float* __restrict__ md2p) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint32_t q = pos[v], p = parents[v];
    vec_id[q] = (uint32_t)v;
    parent_pos[q] = p == 0xffffffffu ? 0xffffffffu : pos[p];
    depth_out[q] = depth[v];
    subtree[q] = size[v] - 1;  // h:1182: node_id - parent_node_id
    unsigned mk = 0;
    if (p != 0xffffffffu)
        for (int m = 0; m < M; ++m)
            if (codes[(size_t)p * M + m] != codes[(size_t)v * M + m]) mk |= 1u << m;
    mask[q] = (uint16_t)mk;
    n_changed[q] = (uint32_t)__popc(mk);
    if (md) {
        md[q] = sqrtf(__uint_as_float(max_dists[v]));   // h:1455-1456 (sqrtf is correctly rounded)
        md2p[q] = sqrtf(__uint_as_float(max_d2p[v]));
    }
}

__global__ void changed_bytes_kernel(const uint8_t* __restrict__ codes, int M, const uint32_t* __restrict__ vec_id,
                                     const uint32_t* __restrict__ parent_pos, const uint16_t* __restrict__ mask,
                                     const uint32_t* __restrict__ at, int64_t n, uint8_t* __restrict__ deltas,
                                     uint8_t* __restrict__ delta_from) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n || q == 0) return;
    const uint32_t v = vec_id[q], p = vec_id[parent_pos[q]];
    uint32_t o = at[q];
    const unsigned mk = mask[q];
    for (int m = 0; m < M; ++m)
        if (mk >> m & 1u) {
            deltas[o] = codes[(size_t)v * M + m];
            delta_from[o] = codes[(size_t)p * M + m];
            ++o;
        }
}

}  // namespace

int layout_tree_gpu(const uint8_t* codes, int64_t n, int M, int K, int max_height_folds, const float* codewords, int Ds,
                    const std::vector<uint32_t>& finalists, std::vector<std::pair<uint32_t, uint32_t>>* edges_in, int device,
                    Tree* out, std::string* err) {
    if (!codes || n < 1 || n >= (int64_t)INT32_MAX || M < 1 || M > 16 || K < 1 || K > 256 || !out || finalists.empty()) {
        if (err) *err = "bad argument to layout_tree_gpu";
        return DPQ_ERR_ARG;
    }
    Tree& t = *out;
    t = Tree();
    t.M = M;
    t.K = K;
    t.max_height_folds = max_height_folds;
    t.n = n;
    t.edges.swap(*edges_in);
    t.root_id = finalists[0];
    for (size_t i = 1; i < finalists.size(); ++i) t.edges.emplace_back(t.root_id, finalists[i]);  // h:1297-1313
    if ((int64_t)t.edges.size() != n - 1) {
        if (err) *err = "internal: edge count != n - 1";
        return DPQ_ERR_FORMAT;
    }
    t.root_code.assign(codes + (size_t)t.root_id * M, codes + (size_t)t.root_id * M + M);
    const int levels = M > 8 ? 16 : 8;
    const size_t N = (size_t)n, E = N - 1;
    const bool with_cb = codewords && Ds > 0;
    uint8_t *d_codes = nullptr, *d_depth = nullptr, *d_depth_out = nullptr, *d_deltas = nullptr, *d_from = nullptr;
    uint32_t *d_edges = nullptr, *d_parents = nullptr, *d_ep = nullptr, *d_ec = nullptr, *d_ep2 = nullptr, *d_ec2 = nullptr,
             *d_key = nullptr, *d_key2 = nullptr, *d_md = nullptr, *d_md2p = nullptr, *d_nch = nullptr, *d_off = nullptr,
             *d_size = nullptr, *d_sz_e = nullptr, *d_scan = nullptr, *d_pos = nullptr, *d_vec = nullptr, *d_ppos = nullptr,
             *d_sub = nullptr, *d_cnt = nullptr, *d_at = nullptr, *d_nset = nullptr;
    uint16_t* d_mask = nullptr;
    float *d_cw = nullptr, *d_tab = nullptr, *d_fmd = nullptr, *d_fmd2p = nullptr;
    void* d_temp = nullptr;
    size_t temp_bytes = 0;
    int max_depth = 0;
    bool too_deep = false;

    auto run = [&]() -> bool {
    GB_HIP(hipSetDevice(device));
    GB_HIP(hipMalloc(&d_codes, N * M));
    GB_HIP(hipMemcpy(d_codes, codes, N * M, hipMemcpyHostToDevice));
    GB_HIP(hipMalloc(&d_parents, N * 4));
    GB_HIP(hipMemset(d_parents, 0xff, N * 4));
    GB_HIP(hipMalloc(&d_md, N * 4));
    GB_HIP(hipMalloc(&d_md2p, N * 4));
    GB_HIP(hipMemset(d_md, 0, N * 4));
    GB_HIP(hipMemset(d_md2p, 0, N * 4));
    GB_HIP(hipMalloc(&d_nch, (N + 1) * 4));
    GB_HIP(hipMemset(d_nch, 0, (N + 1) * 4));
    GB_HIP(hipMalloc(&d_off, (N + 1) * 4));
    GB_HIP(hipMalloc(&d_depth, N));
    GB_HIP(hipMemset(d_depth, 0xff, N));
    GB_HIP(hipMalloc(&d_size, N * 4));
    GB_HIP(hipMalloc(&d_pos, N * 4));
    GB_HIP(hipMemset(d_pos, 0, N * 4));
    GB_HIP(hipMalloc(&d_nset, 4));
    if (E > 0) {
        GB_HIP(hipMalloc(&d_edges, E * 8));
        GB_HIP(hipMemcpy(d_edges, t.edges.data(), E * 8, hipMemcpyHostToDevice));
        GB_HIP(hipMalloc(&d_ep, E * 4));
        GB_HIP(hipMalloc(&d_ec, E * 4));
        GB_HIP(hipMalloc(&d_ep2, E * 4));
        GB_HIP(hipMalloc(&d_ec2, E * 4));
        GB_HIP(hipMalloc(&d_key, E * 4));
        GB_HIP(hipMalloc(&d_key2, E * 4));
        GB_HIP(hipMalloc(&d_sz_e, E * 4));
        GB_HIP(hipMalloc(&d_scan, E * 4));
        size_t a = 0, b = 0;
        hipcub::DeviceRadixSort::SortPairs(nullptr, a, d_key, d_key2, d_ec, d_ec2, (int)E);
        hipcub::DeviceScan::ExclusiveSum(nullptr, b, d_nch, d_off, (int)(N + 1));
        temp_bytes = std::max(a, b) + 256;
        GB_HIP(hipMalloc(&d_temp, temp_bytes));
        hipLaunchKernelGGL(set_parents_kernel, dim3(blocks_for((int64_t)E)), dim3(256), 0, 0, d_edges, (int64_t)E, d_parents,
                           d_ep, d_ec);
        if (with_cb) {
            const size_t cwn = (size_t)M * K * Ds, tn = (size_t)M * K * K;
            GB_HIP(hipMalloc(&d_cw, cwn * 4));
            GB_HIP(hipMemcpy(d_cw, codewords, cwn * 4, hipMemcpyHostToDevice));
            GB_HIP(hipMalloc(&d_tab, tn * 4));
            hipLaunchKernelGGL(centroid_table_kernel, dim3(blocks_for((int64_t)tn)), dim3(256), 0, 0, d_cw, M, K, Ds, d_tab);
            hipLaunchKernelGGL(ancestor_dist_kernel, dim3(blocks_for(n)), dim3(256), 0, 0, d_codes, M, K, d_parents, d_tab, n,
                               d_md, d_md2p);
            // siblings by max_dist2p descending, ties in edge order: stable sort by ~bits, then (below) by parent
            hipLaunchKernelGGL(sibling_key_kernel, dim3(blocks_for((int64_t)E)), dim3(256), 0, 0, d_ec, d_md2p, (int64_t)E,
                               d_key);
            size_t tb = temp_bytes;
            GB_HIP(hipcub::DeviceRadixSort::SortPairs(d_temp, tb, d_key, d_key2, d_ec, d_ec2, (int)E));
            tb = temp_bytes;
            GB_HIP(hipcub::DeviceRadixSort::SortPairs(d_temp, tb, d_key, d_key2, d_ep, d_ep2, (int)E));
            std::swap(d_ec, d_ec2);
            std::swap(d_ep, d_ep2);
        }
        // stable sort by parent: the adjacency lists (h:1077), each in sibling order
        size_t tb = temp_bytes;
        GB_HIP(hipcub::DeviceRadixSort::SortPairs(d_temp, tb, d_ep, d_key2, d_ec, d_ec2, (int)E));
        std::swap(d_ec, d_ec2);   // d_ec: children, grouped by parent
        std::swap(d_ep, d_key2);  // d_ep: their parents (sorted)
        hipLaunchKernelGGL(count_children_kernel, dim3(blocks_for((int64_t)E)), dim3(256), 0, 0, d_ep, (int64_t)E, d_nch);
        tb = temp_bytes;
        GB_HIP(hipcub::DeviceScan::ExclusiveSum(d_temp, tb, d_nch, d_off, (int)(N + 1)));
    }
    // depth, level by level from the root
    {
        const uint8_t zero = 0;
        GB_HIP(hipMemcpy(d_depth + t.root_id, &zero, 1, hipMemcpyHostToDevice));
        for (int level = 1; level <= 64; ++level) {
            GB_HIP(hipMemset(d_nset, 0, 4));
            hipLaunchKernelGGL(depth_level_kernel, dim3(blocks_for(n)), dim3(256), 0, 0, d_parents, n, level, d_depth, d_nset);
            uint32_t n_set = 0;
            GB_HIP(hipMemcpy(&n_set, d_nset, 4, hipMemcpyDeviceToHost));
            if (n_set == 0) break;
            max_depth = level;
            if (level >= levels) {
                too_deep = true;
                return true;
            }
        }
    }
    // subtree sizes bottom-up, DFS positions top-down
    {
        std::vector<uint32_t> ones(N, 1u);
        GB_HIP(hipMemcpy(d_size, ones.data(), N * 4, hipMemcpyHostToDevice));
    }
    for (int level = max_depth; level >= 1; --level)
        hipLaunchKernelGGL(subtree_level_kernel, dim3(blocks_for(n)), dim3(256), 0, 0, d_parents, d_depth, n, level, d_size);
    if (E > 0) {
        hipLaunchKernelGGL(gather_sizes_kernel, dim3(blocks_for((int64_t)E)), dim3(256), 0, 0, d_ec, d_size, (int64_t)E, d_sz_e);
        size_t tb = temp_bytes;
        GB_HIP(hipcub::DeviceScan::ExclusiveSum(d_temp, tb, d_sz_e, d_scan, (int)E));
        for (int level = 1; level <= max_depth; ++level)
            hipLaunchKernelGGL(position_level_kernel, dim3(blocks_for((int64_t)E)), dim3(256), 0, 0, d_ep, d_ec, d_scan, d_off,
                               d_depth, (int64_t)E, level, d_pos);
    }
    // node records in DFS order, then the changed bytes
    GB_HIP(hipMalloc(&d_vec, N * 4));
    GB_HIP(hipMalloc(&d_ppos, N * 4));
    GB_HIP(hipMalloc(&d_depth_out, N));
    GB_HIP(hipMalloc(&d_sub, N * 4));
    GB_HIP(hipMalloc(&d_mask, N * 2));
    GB_HIP(hipMalloc(&d_cnt, (N + 1) * 4));
    GB_HIP(hipMemset(d_cnt, 0, (N + 1) * 4));
    GB_HIP(hipMalloc(&d_at, (N + 1) * 4));
    if (with_cb) {
        GB_HIP(hipMalloc(&d_fmd, N * 4));
        GB_HIP(hipMalloc(&d_fmd2p, N * 4));
    }
    hipLaunchKernelGGL(node_records_kernel, dim3(blocks_for(n)), dim3(256), 0, 0, d_codes, M, d_parents, d_pos, d_depth, d_size,
                       d_md, d_md2p, n, d_vec, d_ppos, d_depth_out, d_sub, d_mask, d_cnt, d_fmd, d_fmd2p);
    {
        size_t b = 0;
        hipcub::DeviceScan::ExclusiveSum(nullptr, b, d_cnt, d_at, (int)(N + 1));
        if (b + 256 > temp_bytes) {
            hipFree(d_temp);
            d_temp = nullptr;
            temp_bytes = b + 256;
            GB_HIP(hipMalloc(&d_temp, temp_bytes));
        }
        size_t tb = temp_bytes;
        GB_HIP(hipcub::DeviceScan::ExclusiveSum(d_temp, tb, d_cnt, d_at, (int)(N + 1)));
    }
    uint32_t n_diffs = 0;
    GB_HIP(hipMemcpy(&n_diffs, d_at + N, 4, hipMemcpyDeviceToHost));
    GB_HIP(hipMalloc(&d_deltas, (size_t)n_diffs + 16));
    GB_HIP(hipMalloc(&d_from, (size_t)n_diffs + 16));
    hipLaunchKernelGGL(changed_bytes_kernel, dim3(blocks_for(n)), dim3(256), 0, 0, d_codes, M, d_vec, d_ppos, d_mask, d_at, n,
                       d_deltas, d_from);
    GB_HIP(hipDeviceSynchronize());
    t.vec_id.resize(N);
    t.parent_pos.resize(N);
    t.subtree.resize(N);
    t.depth.resize(N);
    t.mask.resize(N);
    t.deltas.resize(n_diffs);
    t.delta_from.resize(n_diffs);
    GB_HIP(hipMemcpy(t.vec_id.data(), d_vec, N * 4, hipMemcpyDeviceToHost));
    GB_HIP(hipMemcpy(t.parent_pos.data(), d_ppos, N * 4, hipMemcpyDeviceToHost));
    GB_HIP(hipMemcpy(t.subtree.data(), d_sub, N * 4, hipMemcpyDeviceToHost));
    GB_HIP(hipMemcpy(t.depth.data(), d_depth_out, N, hipMemcpyDeviceToHost));
    GB_HIP(hipMemcpy(t.mask.data(), d_mask, N * 2, hipMemcpyDeviceToHost));
    if (n_diffs) {
        GB_HIP(hipMemcpy(t.deltas.data(), d_deltas, n_diffs, hipMemcpyDeviceToHost));
        GB_HIP(hipMemcpy(t.delta_from.data(), d_from, n_diffs, hipMemcpyDeviceToHost));
    }
    if (with_cb) {
        t.max_dist.resize(N);
        t.max_dist2p.resize(N);
        GB_HIP(hipMemcpy(t.max_dist.data(), d_fmd, N * 4, hipMemcpyDeviceToHost));
        GB_HIP(hipMemcpy(t.max_dist2p.data(), d_fmd2p, N * 4, hipMemcpyDeviceToHost));
    }
    t.n_diffs = n_diffs;
    return true;
    };
    const bool ok = run();
    hipFree(d_codes); hipFree(d_depth); hipFree(d_depth_out); hipFree(d_deltas); hipFree(d_from); hipFree(d_edges);
    hipFree(d_parents); hipFree(d_ep); hipFree(d_ec); hipFree(d_ep2); hipFree(d_ec2); hipFree(d_key); hipFree(d_key2);
    hipFree(d_md); hipFree(d_md2p); hipFree(d_nch); hipFree(d_off); hipFree(d_size); hipFree(d_sz_e); hipFree(d_scan);
    hipFree(d_pos); hipFree(d_vec); hipFree(d_ppos); hipFree(d_sub); hipFree(d_cnt); hipFree(d_at); hipFree(d_nset);
    hipFree(d_mask); hipFree(d_cw); hipFree(d_tab); hipFree(d_fmd); hipFree(d_fmd2p); hipFree(d_temp);
    if (!ok) return DPQ_ERR_HIP;
    if (too_deep) {
        if (err) *err = "tree is deeper than the DTC depth field allows (lower -h)";
        return DPQ_ERR_FORMAT;
    }
    t.max_depth = max_depth;
    for (size_t p = 0; p < N; ++p) t.depth_hist[t.depth[p]]++;
    return DPQ_OK;
}

}  // namespace dpq
