// dpq_kernels.h -- launch interface of the gfx950 kernels (dpq_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dpq {

constexpr int kScanThreads = 1024;  // 16 wavefronts, one workgroup per CU (LDS-bound residency)
constexpr int kScanWaves = kScanThreads / 64;
constexpr int kMaxTopK = 2048;
constexpr int kSelectThreads = 512;
constexpr int kStage = 128;  // LDS-staged candidates per query per scan workgroup

// queries per LUT group: one scan workgroup keeps QG per-query M x 256 fp32
// tables in LDS (128 KB) and decodes each chunk once for all of them.
inline int queries_per_group(int M) { return M <= 8 ? 16 : 8; }
inline size_t lut_group_floats(int M) { return (size_t)queries_per_group(M) * M * 256; }

// The SoA image of one shard in HBM (see DESIGN.md "Data layout").
struct DeviceImage {
    const uint8_t* nib = nullptr;             // 4-bit depths
    const uint8_t* mask = nullptr;            // 1 or 2 bytes per node
    const uint8_t* delta = nullptr;           // changed bytes
    const uint64_t* seg_delta_off = nullptr;  // [n_segments + 1]
    const uint8_t* seg_ckpt = nullptr;        // [n_segments][levels][M]
    int64_t n_local = 0;                      // nodes in this shard
    int64_t n_codes_total = 0;                // N of the whole index (even-N id quirk)
    uint32_t id_base = 0;                     // global DFS position of local node 0
    int32_t n_segments = 0;
    int32_t chunks_per_segment = 4;
    int32_t M = 8, K = 256;
};

struct ScanArgs {
    DeviceImage img;
    const uint32_t* seg_list;   // segments of this cascade level, or NULL = all
    int32_t n_seg_pass;
    const float* lut;           // grouped images [group][QG/4][M][256][4]
    const int32_t* group_list;  // slot -> LUT group, or NULL = identity
    const float* thr_hi;        // [slots*QG] conservative fp32 accept bound (+inf = take everything)
    const float* thr_lo;        // [slots*QG] below this the candidate is certainly inside
    const uint64_t* thr_key;    // [slots*QG] exact threshold key (fp32 bits << 32 | id)
    uint32_t* cand_count;       // [slots*QG]
    uint32_t* cand_id;          // [slots*QG][cap]
    uint32_t* cand_code;        // [slots*QG][cap][M/4] dwords
    int32_t cap;
};

struct SelectArgs {
    uint32_t* cand_count;        // in: candidates per slot; out (non-final): winners carried to the next level
    uint32_t* cand_id;
    uint32_t* cand_code;
    int32_t cap;
    const float* lut;            // grouped images
    const int32_t* slot_query;   // slot -> query index in the batch (LUT lookup + output row), NULL = identity
    uint64_t* keys;              // scratch [slots][cap]
    int32_t M;
    int32_t top_k;
    int32_t final_pass;          // 1: write ids/dists; 0: write thresholds only
    uint64_t* thr_key;           // out [slots]
    float* thr_hi;               // out [slots]
    float* thr_lo;               // out [slots]
    uint32_t* overflow;          // out [slots]: set to 1 when a level dropped candidates (sticky)
    int32_t* out_ids;            // [nq][top_k]
    float* out_dists;            // [nq][top_k]
    int64_t n_codes_total;
    int64_t n_local;             // nodes in the shard (top_k may exceed it)
};

hipError_t launch_lut_build(const float* d_codebook, const float* d_queries, int nq, int nq_padded, int M, int K,
                            int Ds, float* d_lut, hipStream_t stream);
hipError_t launch_scan(const ScanArgs& a, int n_slots_groups, int splits, hipStream_t stream);
hipError_t launch_select(const SelectArgs& a, int n_slots, hipStream_t stream);
hipError_t launch_init_thresholds(uint64_t* thr_key, float* thr_hi, float* thr_lo, int n, int n_real,
                                  hipStream_t stream);
hipError_t launch_merge(const int32_t* d_ids, const float* d_dists, int n_lists, int nq, int top_k, int32_t* d_out_ids,
                        float* d_out_dists, hipStream_t stream);
size_t scan_lds_bytes(int M);

}  // namespace dpq
