// dpq_kernels.h -- launch interface of the gfx950 kernels (dpq_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "dpq_format.h"  // layout constants of the images (kChunk, kRunLen, kStripNodes, kPhaseLen)

namespace dpq {

constexpr int kScanThreads = 1024;  // 16 wavefronts, one workgroup per CU (LDS-bound residency)
constexpr int kScanWaves = kScanThreads / 64;
constexpr int kMaxTopK = 2048;
constexpr int kSelectThreads = 512;
constexpr int kMaxSplits = 256;                  // scan workgroups per query group
constexpr int kRegionStride = 1 + kMaxSplits;    // candidate-count words per slot
constexpr int kLevel0Nodes = 3840;               // level-0 list: its select block (keys + exact tables) stays under 40 KB of LDS
constexpr int kSortMax = 4096;  // candidate keys the select kernel holds in LDS; more -> radix select on the HBM list
// queries per scan workgroup = what 128 KB of filter tables hold: 8-bit entries for M = 8, 16-bit for M = 16
inline int queries_per_group(int M) { return M <= 8 ? 64 : 32; }
// In-scan threshold tightening: per slot and filter level a histogram of the candidates found so far, by how many
// steps the level's cut could be lowered without losing them (scan_kernel).
constexpr int kTightBuckets = 8;
constexpr int kTightSplits = 32;                 // scan workgroups of a query group that can share their counts
constexpr int kTightWords = kTightSplits * kTightBuckets / 4;  // u32 words per slot: [split][bucket] saturating bytes

// The SoA image of one shard in HBM (see DESIGN.md "Data layout").
struct DeviceImage {
    const uint8_t* nib = nullptr;             // 4 bits per node: stack level its in-chunk ancestor chain hangs from
    const uint8_t* par = nullptr;             // per node: lane of its parent in the chunk, 0xFF = precedes the chunk
    const uint8_t* carry = nullptr;           // [chunks][levels]: lane of the chunk's last node of each depth, 0xFF = none
    const uint8_t* mask = nullptr;            // 1 or 2 bytes per node
    const uint8_t* delta = nullptr;           // changed bytes
    const uint64_t* seg_delta_off = nullptr;  // [n_segments + 1]
    const uint8_t* seg_ckpt = nullptr;        // [n_segments][levels][M]
    const uint8_t* raw = nullptr;             // non-NULL: plain (uncompressed) index, codes[n][M], padded to whole segments
    // strand image (dpq_format.h; NULL = not built): the same nodes, a lane per run of 64
    const uint64_t* st_ckpt = nullptr;        // [n_strips][8][64]
    const uint32_t* st_mask = nullptr;        // [n_strips][16][64]: four mask bytes
    const uint16_t* st_depth = nullptr;       // [n_strips][16][64]: four depth nibbles
    const uint32_t* st_pbase = nullptr;       // [n_strips * 16 + 1], units of 16 bytes
    const uint8_t* st_delta = nullptr;
    int32_t n_strips = 0;
    int64_t n_local = 0;                      // nodes in this shard
    int64_t n_codes_total = 0;                // N of the whole index (even-N id quirk)
    uint32_t id_base = 0;                     // global DFS position of local node 0
    int32_t n_segments = 0;
    int32_t chunks_per_segment = 4;
    int32_t M = 8, K = 256;
};

// Filter scan of one cascade level.
struct ScanArgs {
    DeviceImage img;
    const uint32_t* seg_list;   // segments of this level (a slice of the visiting order), or NULL = all
    int32_t n_seg_pass;
    const float* lut32;         // exact tables [query][m][256] fp32, rows by the labels of the codes the launch reads
                                // (code values; the per-batch scratch: its bank-aware labels)
    const float* lut_min;       // [query][M][4] per-sub-space minima of the exact tables (four partial minima each)
    const uint64_t* thr_key;    // [slots] threshold key of each slot (~0 = keep everything)
    const int32_t* slot_query;  // slot -> query of the batch, NULL = identity, -1 = unused slot (nothing passes)
    int32_t n_queries;          // slots >= n_queries are padding when slot_query == NULL
    int32_t debug_pass;         // developer experiments: 0 normal, 1 nothing passes, 2 everything passes
    unsigned long long* wg_times;  // developer diagnostics: [workgroups][2] start / end on the 100 MHz clock (NULL = off)
    int32_t raw_by_pos;         // img.raw is the per-batch scratch of THIS launch's list: entry s of seg_list at position s
    int32_t append;             // the launch continues a level: region counts start from cand_count instead of 0
    int32_t fp32_accum;         // plain index (-task pqscan): exact distances are fp32 sums in position order (h:2658-2662)
    // candidate buffer of a slot: [region 0: winners carried from the previous level, region_off keys]
    // [region 1 + s: what workgroup (split) s of this launch found, region_cap keys each]
    uint32_t* cand_count;       // [slots][kRegionStride] keys per region (region 1 + s written here, may exceed region_cap)
    uint64_t* cand_key;         // [slots][cand_stride] exact keys (distance bits << 32 | DFS position)
    int64_t cand_stride;
    int32_t region_off, region_cap;
    unsigned long long* counters;  // statistics (may be NULL): [0] += pairs checked exactly, [1] += candidates
    // this level's filter tables, [groups][qtab_bytes_per_group / 16] x 16 B: written by launch_quantise
    // (from lut32 / lut_min / thr_key), copied into LDS by every scan workgroup of the group
    uint4* qtab;
    unsigned long long* stamps;    // developer diagnostics (NULL in every query call): per-section cycle sums
    // In-scan threshold tightening (NULL = off): [groups][kTightSplits][slots of a group][kTightBuckets] saturating byte
    // counters (one row per scan workgroup of the group, written by that workgroup alone), zero when the level starts
    // (lut_build_kernel / quantise_kernel clear them); tight_k = top_k.
    uint32_t* tight_hist;
    int32_t tight_k;
};

struct SelectArgs {
    // candidate source: the level-0 list shared by all queries, or the per-slot buffers
    const uint32_t* shared_id;     // non-NULL: level 0 (0xffffffff = padding node)
    const uint32_t* shared_code;
    int32_t shared_n;
    uint32_t* cand_count;          // [slots][kRegionStride] in: keys per region; out (non-final): region 0 = carried winners
    uint64_t* cand_key;            // [slots][cand_stride], regions as in ScanArgs
    int64_t cand_stride;
    int32_t region_off, region_cap;
    int32_t n_regions;             // regions to read: 1 + splits of the level's scan launch (0 with shared_id)
    uint64_t* scratch;             // [slots][cand_stride] contiguous copy when a slot holds more than kSortMax keys
    const float* lut32;            // exact tables [query][m][256] fp32
    const int32_t* slot_query;     // slot -> query index in the batch (LUT + output row), NULL = identity, -1 = skip
    int32_t top_k;
    int32_t final_pass;            // 1: write ids/dists; 0: carry winners to the next level
    uint64_t* thr_key;             // out [slots]: k-th smallest key seen so far (upper bound of the final one)
    uint32_t* overflow;            // out [slots]: set to 1 when a level dropped candidates (sticky)
    uint32_t* any_overflow;        // out (may be NULL): one word in pinned host memory, set to 1 with any overflow[slot]
    int32_t* out_ids;              // [nq][top_k]
    float* out_dists;              // [nq][top_k]
    int64_t n_codes_total;         // N for the even-N id quirk of the DTC scan; odd (-1) for the plain scan
    int32_t fp32_accum;            // 1: plain-scan rule, distance accumulated in fp32 (h:2658-2662)
    int32_t keep_thr;              // 1: thr_key = min(thr_key, this level's k-th key) (levels after a bootstrap)
    unsigned long long* stamps;    // developer diagnostics (NULL in product calls): [slots][8] s_memtime marks
    int32_t threads;               // block size (256 / 512 / 1024); 0 = by top_k (launch_select)
    int32_t fast_final;            // 1: the last level is a bucket sort from one histogram pass (select_kernel); 0: the exact way
};

// Threshold bootstrap from the inverted multi-index (see bootstrap_kernel).
struct BootArgs {
    const uint8_t* relabel;        // [M][256] code value -> label of the first filter level's table rows (NULL: identity)
    const uint8_t* nbr;            // [8 sort slots][256][256]: per centroid of the slot's sub-space, all centroids nearest first
    const uint32_t* cell_start;    // [n_classes][65537] absolute entry positions
    int32_t n_classes;             // 4 (sub-space pairs 0/1 .. 6/7) or 1 (pair 0/1)
    const uint32_t* mi_code;       // [entries][M / 4]
    const uint32_t* mi_id;         // [entries] global DFS positions
    const float* lut32;            // exact tables [query][m][256]
    const int32_t* slot_query;     // slot -> query, NULL = identity, -1 = skip
    int32_t top_k;
    int32_t target;                // stop walking cells once this many nodes are evaluated
    int32_t cap;                   // keys the block holds (LDS): target <= cap
    uint64_t* thr_key;             // out [slots]
    uint32_t* cand_count;          // [slots][kRegionStride]: region 0 (carried winners) is set to 0
    int32_t fp32_accum;
    int32_t n_queries;             // slots >= n_queries (slot_query == NULL) are padding of the last query group
    // non-NULL: also write the slot's fields of the first filter level's tables (what quantise_kernel would build
    // from this threshold); the launch then covers the padding slots too
    uint4* qtab;
    const float* lut_min;          // [query][M][4]
    unsigned long long* stamps;    // developer diagnostics (NULL in product calls): [slots][8] s_memtime marks
    int32_t variant;               // bootstrap_kernel's V: 1 = round 4's kernel (one-pass threshold), 0 = round 3's, kept for A/B
};

// Builds the exact tables of queries [0, nq) and clears the candidate counters / overflow flags of
// slots [0, n_slots) (either may be NULL).
hipError_t launch_lut_build(const float* d_codebook, const float* d_queries, int nq, int n_slots, int M, int K, int Ds,
                            float* d_lut32, float* d_lut_min, uint32_t* d_cand_count, uint32_t* d_overflow,
                            const uint8_t* d_relabel, float* d_lut_labels, uint32_t* d_tight_hist,
                            hipStream_t stream);
hipError_t launch_decode_list(const DeviceImage& img, const uint32_t* seg_list, int n_seg, const uint8_t* relabel,
                              uint32_t* out_code, hipStream_t stream);
hipError_t launch_decode_segments(const DeviceImage& img, const uint32_t* seg_list, int n_seg, uint32_t* out_id,
                                  uint32_t* out_code, hipStream_t stream);
// launch_quantise builds the level's filter tables of every query group (needs the level's thresholds in
// thr_key); launch_scan must follow it on the same stream.
hipError_t launch_stream(const ScanArgs& a, int n_slots, hipStream_t stream);
// The same pass over the strand image (M = 8, img.st_* set): a.seg_list / a.n_seg_pass name STRIPS here.
hipError_t launch_strand(const ScanArgs& a, int n_slots, hipStream_t stream);
int stream_queries_per_pass(int M, int n_slots);  // 1, 2, 4 or 8 (M = 16: at most 4)
// One query per pass over the strand image (strand1_kernel): its workgroups, each with its own candidate region
// (region 1 + w of the slot, a.region_cap keys, count published by the kernel).
int strand1_workgroups(int n_strips);
constexpr int kStrand1Regions = 256;
hipError_t launch_quantise(const ScanArgs& a, int n_slot_groups, hipStream_t stream);
hipError_t launch_scan(const ScanArgs& a, int n_slot_groups, int splits, hipStream_t stream);
size_t qtab_bytes_per_group(int M);
int scan_stamp_count();
hipError_t launch_select(const SelectArgs& a, int M, int n_slots, hipStream_t stream);
hipError_t launch_bootstrap(const BootArgs& a, int M, int n_slots, hipStream_t stream);
// row_stride: elements between consecutive (list, query) rows of d_ids / d_dists (top_k, or 2 * top_k for the packed tensor)
hipError_t launch_merge(const int32_t* d_ids, const float* d_dists, int n_lists, int nq, int top_k, int row_stride,
                        int32_t* d_out_ids, float* d_out_dists, hipStream_t stream);
// PQ encoding: codes[n][M] = argmin_k |v_m - c[m][k]|^2 in fp32.
hipError_t launch_encode_pq(const float* d_vectors, int64_t n, int D, const float* d_codebook, int M, int K, int Ds,
                            uint8_t* d_codes, hipStream_t stream);
size_t scan_lds_bytes(int M);
size_t select_lds_bytes(int M, int top_k, int n_shared);

}  // namespace dpq
