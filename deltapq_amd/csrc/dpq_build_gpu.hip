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
// O(n) tail of the build (adjacency, sibling order, DFS numbering) stays on
// the host (layout_tree).
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

__global__ void unmerged_flags_kernel(const uint32_t* __restrict__ ids, const uint8_t* __restrict__ merged, int64_t n,
                                      uint8_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = merged[ids[i]] ? 0 : 1;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

int find_edges_gpu(const uint8_t* codes, int64_t n, int M, int max_height_folds, int device,
                   std::vector<uint32_t>* finalists_out, std::vector<std::pair<uint32_t, uint32_t>>* edges_out,
                   std::string* err) {
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

    for (int diff = 0; diff <= M; ++diff) {
        position_subsets(M, M - diff, &combos);
        for (const auto& kept : combos) {
            // act = unmerged ids of cur, order kept
            hipLaunchKernelGGL(unmerged_flags_kernel, dim3(blocks_for(n_cur)), dim3(256), 0, 0, d_cur, d_merged, n_cur,
                               d_flags);
            size_t tb = temp_bytes;
            GB_HIP(hipcub::DeviceSelect::Flagged(d_temp, tb, d_cur, d_flags, d_act, d_num, (int)n_cur));
            uint32_t n_act = 0;
            GB_HIP(hipMemcpy(&n_act, d_num, 4, hipMemcpyDeviceToHost));
            if (n_act < 2) break;
            uint64_t mlo = 0, mhi = 0;
            for (int pos : kept) {
                if (pos < 8) mlo |= 0xffull << (8 * pos);
                else mhi |= 0xffull << (8 * (pos - 8));
            }
            hipLaunchKernelGGL(make_keys_kernel, dim3(blocks_for(n_act)), dim3(256), 0, 0, d_codes, M, d_act,
                               (int64_t)n_act, mlo, mhi, d_klo_a, wide ? d_khi_a : nullptr);
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
    return ok ? DPQ_OK : DPQ_ERR_HIP;
}

}  // namespace dpq
