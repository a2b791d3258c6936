/* deltapq_amd.h -- C-ABI of the MI355X-native DeltaPQ query engine.
 *
 * This is the drop-in boundary for ONE path of RunhuiWang/DeltaPQ: the
 * delta-tree scan behind `deltapq -task query` / `-task query_im`.  The
 * reference has no FFI layer; its boundary is a C++ free function called once
 * per query (citations are file:line into the reference repository):
 *
 *   void query_processing_scan_compressed_codes_opt_o_direct(
 *       const string& dataset_path, const vector<float>& query, int top_k,
 *       int M, int K, int m_Ds, uint num_codes,
 *       const vector<PQ::Array>& m_codewords,
 *       vector<pair<int,float>>& results, uchar** decoder);
 *                                   deltapq_create_approx_tree.h:2805-2810
 *   void query_processing_scan_compressed_codes_opt_in_memory(
 *       uchar* codes, long long n_bytes, ...same...);
 *                                   deltapq_create_approx_tree.h:3731-3736
 *
 * A C-ABI replacement splits that call into load-once / query-many:
 *
 *   reference                                   this library
 *   ------------------------------------------  ---------------------------------
 *   open()+read() of the DTC file per query     dpq_open_file / dpq_open_memory
 *     (h:2812-2824; main:624-634 for query_im)    (parse, validate, transcode to
 *                                                 SoA, upload to HBM -- once)
 *   m_codewords argument (h:2809)               dpq_set_codebook
 *   one call per query (main:328-339)           dpq_query_batch[_device]
 *   results[top_k] of (int id, float dist),     ids[nq][top_k] int32,
 *     ascending (h:2977-2982)                     dists[nq][top_k] float, ascending
 *   decoder[256] argument (main:312-325)        gone (popcount / byte permute on GPU)
 *   PQ::ReadCodewords (pq.cpp:288-312)          dpq_read_codewords
 *   ReadTopN(query.{fvecs,bvecs})               dpq_read_vecs
 *     (utils.cpp:14-110)
 *
 * Conventions: plain pointers and sizes only; the caller owns every host
 * buffer it passes, the library owns device memory.  Every function returns a
 * dpq_status (0 = OK, negative = error) instead of the reference's
 * print-and-continue (h:2819-2821); dpq_last_error() gives a thread-local
 * detail string.  A handle is bound to one GPU and is safe to use from one
 * thread at a time.  Nothing here falls back to a CPU implementation: without
 * a usable GPU dpq_open_* fails with DPQ_ERR_NO_DEVICE.
 *
 * Result semantics (identical to the reference, see DESIGN.md "Parity"):
 *   - ids are DFS positions in the index (h:2910, 2979), NOT original vector ids;
 *   - distance = fp64 sum of the M fp32 table entries, rounded to fp32, which is
 *     bit-identical to the reference's incremental fp64 stack (h:2889-2907);
 *   - for even N the last DFS node is reported with id N, not N-1 (h:2949, 2970);
 *   - equal-distance results are ordered by ascending id (the reference emits
 *     them in libstdc++ heap order); at the k-th boundary the lowest ids win.
 */
#ifndef DELTAPQ_AMD_H
#define DELTAPQ_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPQ_VERSION 100 /* 0.1.0 */

typedef enum dpq_status {
    DPQ_OK = 0,
    DPQ_ERR_ARG = -1,       /* bad argument (NULL, M/K unsupported, top_k < 1, ...) */
    DPQ_ERR_IO = -2,        /* cannot open / short read */
    DPQ_ERR_FORMAT = -3,    /* DTC stream violates the format invariants */
    DPQ_ERR_NO_DEVICE = -4, /* no usable gfx950 GPU / bad device ordinal */
    DPQ_ERR_HIP = -5,       /* HIP runtime error (message in dpq_last_error) */
    DPQ_ERR_NOMEM = -6,
    DPQ_ERR_STATE = -7,     /* e.g. query before dpq_set_codebook */
    DPQ_ERR_TOPK = -8       /* top_k > number of codes (reference: pops an empty heap, h:2977-2981) */
} dpq_status;

typedef struct dpq_index dpq_index; /* opaque: one DTC index (or one shard of it) resident on one GPU */
typedef struct dpq_soa dpq_soa;     /* opaque: host-side transcoded image (no GPU needed) */
typedef struct dpq_tree dpq_tree;   /* opaque: a DeltaTree in DFS layout built from raw PQ codes (host) */

/* Options for dpq_open_*.  Zero-initialise, then set what you need. */
typedef struct dpq_open_opts {
    int32_t device;             /* HIP device ordinal */
    int32_t shard_rank;         /* this handle holds shard `shard_rank` of `shard_count` */
    int32_t shard_count;        /* 0 or 1 = whole index; shards are contiguous DFS-position ranges
                                   cut at segment boundaries and balanced by payload bytes */
    int32_t chunks_per_segment; /* 64-node chunks per independently decodable segment; 0 = default (2) */
    int32_t cand_capacity;      /* candidate keys per query and cascade level, shared out evenly to the scan
                                 * workgroups of the query's group; 0 = auto (16 K keys, >= 256 per workgroup) */
    int32_t num_codes;          /* 0 = the whole index; n > 0 = scan only the first n codes of it, the reference's `-N`
                                 * smaller than the header's n_codes (h:2825-2829 "scan only part of the codes").
                                 * Odd n: exactly the reference's result.  Even n: the reference reads the pair byte of
                                 * node n-1 as a whole-byte depth (>= 16: a stack row out of bounds, undefined); this
                                 * build decodes node n-1 with its real depth and reports it with id n, as the trailing
                                 * rule (h:2949, 2970) does for an index of n codes. */
    int32_t bootstrap;          /* threshold bootstrap (an inverted multi-index over the shard's nodes, 12 B per
                                 * sampled node, that gives every query a tight first threshold): 0 = automatic (on
                                 * from 64 K nodes per shard), 1 = on (from 16 K nodes), -1 = off (the spread-sample
                                 * cascade alone).  Results are identical either way.  dpq_soa_build: > 0 = build the
                                 * multi-index with this sampling stride. */
    int32_t batch_decode;       /* where the delta decode happens.  0 = automatic: a batch of >= 3 query groups (64 queries
                                 * each; 32 at M = 16) decodes every segment ONCE, tile by tile (16 M nodes of a filter
                                 * level's segment list at a time; a shard up to that size is one tile), into a
                                 * plain-code scratch (M bytes per node of a tile, per pipeline lane: at most 128 MB at
                                 * M = 8, cache-resident) that all its groups' filter passes read; smaller batches
                                 * decode inside the scan, once per group.  1 = scratch always, -1 = never,
                                 * n >= 2 = scratch always with tiles of n segments (testing aid).  Results are
                                 * identical either way. */
    int64_t global_offset;      /* the payload is a self-contained PART of a larger index (its first node carries a
                                 * whole code): ids are reported as global_offset + position in this payload */
    int64_t global_n_codes;     /* 0 = this payload is the whole index (global_offset must then be 0); else N of the
                                 * larger index (the even-N id rule h:2949, 2970 then applies to its last node only) */
    /* ---- plan and tiling knobs.  0 = the measured default; results are identical whatever they hold.  They are
     * part of the options (not of the environment) so that the ranks of a multi-GPU launch cannot diverge from
     * each other through their environments.  With DPQ_DEV=1 in the environment -- developer sweeps only -- the
     * variables named in brackets override them, read once per dpq_open_*. ---- */
    int32_t stream_max_queries; /* batches of up to this many queries take the stream kernel (1, 2 or 4 queries per pass over
                                 * the compressed image, exact tables in LDS, no filter tables) instead of the 64-query
                                 * filter scan: 0 = the measured switch-over (4), -1 = never  [DPQ_STREAM_MAX_QUERIES] */
    int32_t coarse_below;       /* batches of up to this many queries use the coarse cascade plan on shards without a
                                 * threshold bootstrap; 0 = 128  [DPQ_COARSE_BELOW] */
    int32_t plan_ratios[3];     /* force the size ratios between consecutive filter levels (each >= 2); 0 = automatic
                                 *   [DPQ_PLAN_RATIOS=a,b,c] */
    int32_t boot_cap;           /* nodes a bootstrap block may hold (2048..16384); 0 = 3072 / 6144 (M = 16) up to
                                 * top-256, then 12288 (M = 8 up to top-640: 6144)  [DPQ_BOOT_CAP] */
    int32_t boot_target;        /* nodes after which the bootstrap stops walking cells; 0 = boot_cap  [DPQ_BOOT_TARGET] */
    int32_t flags;              /* DPQ_OPT_* bits below */
    int64_t batch_tile_nodes;   /* nodes per tile of the per-batch plain-code scratch; 0 = 16 M  [DPQ_BATCH_TILE_NODES] */
} dpq_open_opts;

/* dpq_open_opts.flags (developer A/B switches; every combination gives the same results) */
#define DPQ_OPT_NO_RELABEL 1u       /* plain-code scratch holds code values, not bank-aware labels  [DPQ_RELABEL=0] */
#define DPQ_OPT_NO_FUSE_QUANTISE 2u /* first filter level's tables by quantise_kernel, not by the bootstrap  [DPQ_FUSE_QUANTISE=0] */
#define DPQ_OPT_NO_ASYNC_OVERLAP 4u /* dpq_query_batch_device_async: one workspace, the caller's stream  [DPQ_ASYNC_OVERLAP=0] */
#define DPQ_OPT_BOOT_FULLSORT 8u    /* bootstrap ranks all 256 centroids exactly  [DPQ_BOOT_FULLSORT=1] */
#define DPQ_OPT_NO_TIGHTEN 16u      /* filter scans keep a level's thresholds as they were when it started instead of lowering
                                     * them as candidates accumulate  [DPQ_TIGHTEN=0] */
#define DPQ_OPT_NO_STRANDS 32u      /* no second, lane-per-run layout of the index for batches of up to four queries (they then
                                     * take the wavefront-per-chunk decode; saves ~1.2 x the payload in HBM)  [DPQ_STRANDS=0] */
#define DPQ_OPT_FORCE_STRANDS 64u   /* that layout for every small batch, whatever the shard size (by default from 8 M codes per
                                     * GPU): tests, experiments  [DPQ_STRANDS=2] */
#define DPQ_OPT_NO_STRAND1 128u      /* one query per pass over that layout takes the exact-table kernel (strand_kernel<1>) instead of
                                     * the bound-table kernel with in-kernel tightening (strand1_kernel)  [DPQ_STRAND1=0] */

typedef struct dpq_info {
    int64_t n_codes_total;     /* N of the whole index (header field 0, h:1839-1840) */
    int64_t n_bytes_total;     /* payload bytes of the whole index (header field 1, h:1841) */
    int64_t node_lo, node_hi;  /* DFS positions [lo, hi) held by this handle */
    int64_t algorithmic_bytes; /* DTC payload bytes that encode [lo, hi): the per-query roofline numerator */
    int64_t device_bytes;      /* HBM bytes of the SoA image (nibbles + masks + deltas + tables) */
    int64_t n_diffs;           /* changed bytes in [lo, hi) */
    int32_t M, K, Ds;
    int32_t n_segments;
    int32_t chunks_per_segment;
    int32_t max_depth;
    int32_t device;
    int32_t cand_capacity;
    int64_t bootstrap_bytes;   /* HBM bytes of the threshold-bootstrap multi-index (0 = not in use) */
    int32_t bootstrap_stride;  /* every bootstrap_stride-th node is in it */
    int32_t batch_decode_mb;   /* MB of plain-code scratch (one tile, per pipeline lane) a batch decodes into; 0 = this
                                * handle always decodes inside the scan */
    int64_t strand_bytes;      /* HBM bytes of the strand image (the stream pass's own layout of the same nodes, resident
                                * beside the SoA image on shards from 8 M codes); 0 = not built */
} dpq_info;

/* Per-kernel device time accumulated since the last dpq_profile_reset, measured
 * with hipEvents recorded on the launch stream (profiling must be enabled). */
typedef struct dpq_profile {
    double lut_ms, scan_ms, select_ms; /* summed over launches */
    int64_t lut_launches, scan_launches, select_launches;
    int64_t scan_node_query_pairs;     /* (code, query) distance evaluations issued by scan launches */
    int64_t scan_stream_bytes;         /* SoA bytes the scan launches had to read at least once */
    int64_t query_batches, queries;
    int64_t overflow_reruns;           /* queries that needed a second final pass (candidate overflow) */
    int64_t exact_checks;              /* (code, query) pairs the filter let through, checked exactly in the scan */
    int64_t candidates;                /* pairs that passed the exact check (counted with dpq_profile_enable(idx, 1) only) */
    double quantise_ms;                /* filter-table builds (one per scan launch; dpq_profile_enable(idx, 1) only) */
    double decode_ms;                  /* per-batch decodes into the plain-code scratch (dpq_profile_enable(idx, 1) only) */
    double bootstrap_ms;               /* threshold bootstrap launches (NOT part of select_ms; dpq_profile_enable(idx, 1) only) */
    int64_t bootstrap_launches;
    /* which kernel the stream pass (batches of up to stream_max_queries) ran, launch by launch (all three are also
     * counted in scan_launches / scan_ms): wavefront per chunk, lane per run with exact tables, lane per run with the
     * one-query bound table */
    int64_t stream_launches, strand_launches, strand1_launches;
} dpq_profile;

typedef struct dpq_dtc_stats {
    int64_t n_codes, n_bytes, n_diffs;
    int64_t depth_hist[16];
    int32_t max_depth;
    int32_t M;
} dpq_dtc_stats;

/* ---- library ---------------------------------------------------------- */
int dpq_version(void);
const char* dpq_strerror(int status);
const char* dpq_last_error(void);
/* Number of usable GPUs (0 if none / no driver).  Never fails. */
int dpq_device_count(void);

/* ---- loaders around the path (host only; a10 in SURVEY.md section 8) ---- */
/* DTC file header: int64 n_codes, int64 n_bytes (h:1839-1841, read at h:2823-2824). */
int dpq_read_dtc_header(const char* path, int64_t* n_codes, int64_t* n_bytes);
/* PQ::ReadCodewords (pq.cpp:288-312).  Call with out == NULL to get the shape. */
int dpq_read_codewords(const char* path, int32_t* M, int32_t* K, int32_t* Ds, float* out);
/* ReadTopN over .fvecs / .bvecs (utils.cpp:14-110).  Call with out == NULL to
 * get the count and dimension; at most `cap` vectors are stored. */
int dpq_read_vecs(const char* path, int is_bvecs, int64_t* n, int32_t* D, float* out, int64_t cap);
/* Reference file name of the index: <dir>/M{M}K{K}_Approx_compressed_codes_opt_N{N} (h:2812-2814). */
int dpq_dtc_file_name(const char* dataset_dir, int M, int K, int64_t N, char* out, int64_t out_len);

/* ---- format (host only) ------------------------------------------------ */
/* Walk a DTC payload, checking every invariant the scan relies on
 * (depth >= 1, depth <= deepest-seen + 1, depth < M, byte count == n_bytes). */
int dpq_dtc_validate(const uint8_t* payload, int64_t n_bytes, int64_t n_codes, int M, dpq_dtc_stats* stats);
/* Transcode (a shard of) a DTC payload into the structure-of-arrays image the
 * GPU scans, on the host.  Used by dpq_open_* and exposed for CPU-only tests. */
int dpq_soa_build(const uint8_t* payload, int64_t n_bytes, int64_t n_codes, int M, const dpq_open_opts* opts,
                  dpq_soa** out);
int dpq_soa_info(const dpq_soa* soa, dpq_info* info);
/* Borrowed pointers into the image (valid until dpq_soa_free):
 * which = 0 depth nibbles, 1 masks, 2 deltas, 3 segment delta offsets (u64[n_seg+1]),
 * 4 segment ancestor checkpoints (u8[n_seg][levels][M]), 5-7 the bootstrap multi-index (cell starts, codes, ids),
 * 8 parent lanes, 9 carry lanes, 10-14 the strand image of the stream pass (checkpoints u64[strips][8][64],
 * mask bytes u32[strips][16][64] (four nodes each), phase offsets u16[strips][16][64], phase starts u32[strips*16+1] in
 * 16-byte units, changed bytes), 15 the strand image's depth nibbles u16[strips][16][64] (four nodes each). */
int dpq_soa_array(const dpq_soa* soa, int which, const void** ptr, int64_t* n_bytes);
void dpq_soa_free(dpq_soa* soa);
/* Serialise a tree given as per-node arrays into the reference DTC payload
 * (qnodes_to_compressed_codes_opt, h:1765-1826).  depths[0] must be 0;
 * masks[i] bit m set <=> position m changes; deltas = changed bytes in node
 * order, ascending position.  Call with out == NULL to get n_bytes. */
int dpq_dtc_encode(const uint8_t* root_code, const uint8_t* depths, const uint16_t* masks, const uint8_t* deltas,
                   int64_t n_codes, int M, uint8_t* out, int64_t* n_bytes);

/* ---- callers either side of the path -------------- */
/* DeltaTree construction, `deltapq -task approx_tree` with -method 1 (create_approx_tree h:970-1065:
 * find_edges_by_diff_approx h:1207-1332, edges_to_tree_index_approx_dfs_layout h:1334-1487).  Host code.
 * codes[n][M] raw PQ codes; codewords [M][K][Ds] may be NULL (it only orders siblings). */
int dpq_tree_build(const uint8_t* codes, int64_t n_codes, int M, int K, int max_height_folds, const float* codewords,
                   int Ds, dpq_tree** out);
/* Same tree, with the edge search (the sort/group passes over all position subsets) on GPU `device`;
 * the result is identical to dpq_tree_build's, node for node. */
int dpq_tree_build_gpu(const uint8_t* codes, int64_t n_codes, int M, int K, int max_height_folds,
                       const float* codewords, int Ds, int device, dpq_tree** out);
int dpq_tree_stats(const dpq_tree* t, dpq_dtc_stats* stats);
/* Borrowed arrays: which = 0 vec_id u32[n] (DFS position -> original id, QNode.vec_id h:80), 1 parent_pos u32[n],
 * 2 depth u8[n], 3 mask u16[n], 4 changed bytes, 5 root code u8[M], 6 edges (parent id, child id) u32[n-1][2]. */
int dpq_tree_array(const dpq_tree* t, int which, const void** ptr, int64_t* n_bytes);
/* DTC payload of the tree (qnodes_to_compressed_codes_opt h:1765-1826); out == NULL returns the size. */
int dpq_tree_encode(const dpq_tree* t, uint8_t* out, int64_t* n_bytes);
/* Writes the reference's three artefacts into dataset_dir: M{M}K{K}H{h}_Approx_Edges_N{N} (h:1326-1327),
 * M{M}K{K}_Approx_TreeNodesDFS_N{N} (60-byte QNode records h:1484; M <= 8), the DTC index (h:1839-1842). */
int dpq_tree_write_files(const dpq_tree* t, const char* dataset_dir);
void dpq_tree_free(dpq_tree* t);
/* DFS position -> original vector id from a TreeNodesDFS file (QNode.vec_id, h:80, h:1166). */
int dpq_read_qnode_ids(const char* path, int64_t n_codes, uint32_t* vec_ids);
/* codes.bin.plain.M{M}K{K}N{N}: PQTree::Read / Write (pq_tree.cpp:1011-1081).  out == NULL returns n_codes. */
int dpq_read_codes_plain(const char* path, int M, int64_t* n_codes, uint8_t* out);
int dpq_write_codes_plain(const char* path, const uint8_t* codes, int64_t n_codes, int M);
/* The other record layouts PQTree::Read knows (pq_tree.cpp:1050-1078): K > 256 stores two bytes per position
 * (little-endian uint16), `with_id` (flag approx_with_id, main:64) appends a 4-byte int id to every M-byte code.
 * codes_out: n * M * (K > 256 ? 2 : 1) bytes; ids_out: n int32 (with_id only); either may be NULL (first call:
 * both NULL to learn n).  K > 256 together with with_id is refused, as in the reference (pq_tree.cpp:1051-1054).
 * The scan engine itself indexes one byte per position (K <= 256), like the DTC format. */
int dpq_read_codes_plain_ex(const char* path, int M, int K, int with_id, int64_t* n_codes, uint8_t* codes_out,
                            int32_t* ids_out);
/* PQ encoding on the GPU: nearest centroid per sub-space in fp32, first minimum wins
 * (PQTree::EncodePlain pq_tree.cpp:215-237; host buffers in/out). */
int dpq_encode_pq(const float* vectors, int64_t n, int D, const float* codewords, int M, int K, int Ds, int device,
                  uint8_t* codes_out);

/* Plain (uncompressed) PQ index for the comparator scan `-task pqscan` (h:2590-2678): raw codes[n][M],
 * distance accumulated in **fp32** in ascending m (h:2658-2662), ids = positions in the code file (no
 * even-N quirk).  The handle is used with dpq_set_codebook / dpq_query_batch* like a DTC index. */
int dpq_open_plain_memory(const uint8_t* codes, int64_t n_codes, int M, int K, const dpq_open_opts* opts,
                          dpq_index** out);
/* <path> = codes.bin.plain.M{M}K{K}N{N} (pq_tree.cpp:1011-1031). */
int dpq_open_plain_file(const char* path, int M, int K, const dpq_open_opts* opts, dpq_index** out);

/* ---- index lifetime (GPU) ---------------------------------------------- */
/* Replaces the per-query open()/read() of h:2812-2824: loads
 * <path> = int64 n_codes, int64 n_bytes, payload. */
int dpq_open_file(const char* path, int M, int K, const dpq_open_opts* opts, dpq_index** out);
/* In-memory twin (h:3731-3733: `uchar* codes, long long n_bytes`). */
int dpq_open_memory(const uint8_t* payload, int64_t n_bytes, int64_t n_codes, int M, int K,
                    const dpq_open_opts* opts, dpq_index** out);
/* m_codewords[M][K][Ds] (h:2809), row-major fp32, copied to the GPU. */
int dpq_set_codebook(dpq_index* idx, const float* codewords, int Ds);
int dpq_get_info(const dpq_index* idx, dpq_info* info);
int dpq_close(dpq_index* idx);

/* ---- the hot path ------------------------------------------------------- */
/* Answer nq queries (host buffers; synchronous).  queries[nq][M*Ds];
 * ids[nq][top_k], dists[nq][top_k] ascending by (distance, id).  With
 * shard_count > 1 the lists are this shard's partial top-k with GLOBAL DFS
 * positions; rows are padded with id -1 / +inf when the shard holds fewer than
 * top_k codes. */
int dpq_query_batch(dpq_index* idx, const float* queries, int nq, int top_k, int32_t* ids, float* dists);
/* Same with device pointers, enqueued on `hip_stream` (a hipStream_t, NULL =
 * default stream), asynchronous with respect to the host except for one
 * overflow check at the end of the batch. */
int dpq_query_batch_device(dpq_index* idx, const float* d_queries, int nq, int top_k, int32_t* d_ids,
                           float* d_dists, void* hip_stream);
/* Pipelined variant: enqueues the batch and returns at once.  The batch starts once `hip_stream` has reached the
 * point of this call (an event is recorded on it) and runs on one of two internal streams with its own workspace,
 * alternately, so that consecutive batches overlap (a batch's table build runs under the previous batch's scan).
 * Inputs and outputs belong to the library until dpq_finish(idx): they must stay valid, the inputs must not be
 * changed and the outputs not read before it returns.  dpq_finish waits for every enqueued batch and answers again,
 * synchronously, any batch in which a query overflowed its candidate buffers.  A call with a different
 * `hip_stream` first finishes what is in flight; up to 63 batches may be in flight, the 64th call finishes the
 * earlier ones first.  The synchronous entry points finish pending batches before they start.
 * (dpq_open_opts.flags & DPQ_OPT_NO_ASYNC_OVERLAP: every batch on `hip_stream` itself, one workspace.) */
int dpq_query_batch_device_async(dpq_index* idx, const float* d_queries, int nq, int top_k, int32_t* d_ids,
                                 float* d_dists, void* hip_stream);
int dpq_finish(dpq_index* idx);
/* The same pipeline with HOST buffers in and out -- the reference's interface (h:2805-2810: host query, host results),
 * its per-query loop (main:328-339) turned into batches in flight: the queries of batch i + 1 go up and the results of
 * batch i - 1 come down beside batch i's kernels.  Up to sixteen batches in flight (a seventeenth call
 * settles the earlier ones first); `queries`, `ids` and `dists` belong to the library until dpq_finish(idx) returns.  Page-locked
 * buffers (dpq_pin_host / dpq_unpin_host = hipHostRegister, for callers that do not link the HIP runtime) make it a
 * pipeline: the queries go up on a copy stream and the result lists are written by the select kernel straight into the
 * mapped buffers; pageable memory works through staging copies, without the overlap. */
int dpq_query_batch_host_async(dpq_index* idx, const float* queries, int nq, int top_k, int32_t* ids, float* dists);
int dpq_pin_host(void* ptr, int64_t bytes);
int dpq_unpin_host(void* ptr);
/* Merge n_lists partial top-k lists per query (lists[l][nq][top_k]) into the
 * final top_k by (distance, id).  Host version for the single-process
 * multi-GPU CLI, device version for use after an RCCL all-gather. */
int dpq_merge_topk_host(const int32_t* ids, const float* dists, int n_lists, int nq, int top_k, int32_t* out_ids,
                        float* out_dists);
int dpq_merge_topk_device(const int32_t* d_ids, const float* d_dists, int n_lists, int nq, int top_k,
                          int32_t* d_out_ids, float* d_out_dists, int device, void* hip_stream);
/* The same on the tensor the one all-gather of the path delivers: d_packed[n_lists][nq][2 * top_k] int32, a row =
 * top_k ids followed by the bit patterns of the top_k fp32 distances (what every rank contributes in ONE collective). */
int dpq_merge_topk_device_packed(const int32_t* d_packed, int n_lists, int nq, int top_k, int32_t* d_out_ids,
                                 float* d_out_dists, int device, void* hip_stream);

/* Stream-ordered variant for callers that consume the result ON THE DEVICE, in stream order (the sharded driver:
 * select -> pack -> all-gather -> merge without a host round trip per batch): the batch is enqueued on `hip_stream`
 * itself; work enqueued on that stream afterwards sees the result -- PROVIDED no query of the batch overflowed its
 * candidate buffers, which only dpq_finish can tell (it answers such a batch again; whatever consumed the first
 * answer must then be redone).  dpq_finish_count reports how many batches that happened to.
 * Up to TWO streams may have stream-ordered batches in flight at once: each is given one of the library's two
 * workspaces, so a caller that alternates its steps between two streams overlaps a step's table build and bootstrap
 * with the previous step's scan and select (a third stream, or a dpq_query_batch_device_async batch, first settles
 * what is in flight). */
int dpq_query_batch_device_ordered(dpq_index* idx, const float* d_queries, int nq, int top_k, int32_t* d_ids,
                                   float* d_dists, void* hip_stream);
int dpq_finish_count(dpq_index* idx, int32_t* rerun_batches);

/* ---- measurement -------------------------------------------------------- */
int dpq_profile_enable(dpq_index* idx, int on);  /* 0 off, 1 every kernel, 2 scan launches only (less event overhead) */
int dpq_profile_reset(dpq_index* idx);
int dpq_profile_read(dpq_index* idx, dpq_profile* out);

/* ---- developer diagnostics ----------------------------------------------
 * Not part of the drop-in boundary: timing and instrumentation hooks used by scripts/ (limiter studies, kernel
 * section marks).  They return DPQ_ERR_STATE unless the process was started with DPQ_DEV=1 in its environment, and
 * only then read their own DPQ_DEBUG_* variables.  A query call never goes through them. */
/* `reps` filter-scan launches over the whole shard for nq slots of the last batch with the filter pinned (pass_all 0:
 * nothing survives, 1: everything, 2: the batch's thresholds, 3: the last batch's bootstrap + first level as they ran). */
int dpq_debug_scan_time(dpq_index* idx, int nq, int pass_all, int reps, int splits, float* ms_out);
/* One launch of the STAMPS build of the scan kernel (M = 8): per-section cycle sums over all wavefronts. */
int dpq_debug_scan_stamps(dpq_index* idx, int nq, int splits, unsigned long long* out, int n_out, float* ms_out);
/* Phase marks of the bootstrap and the last select launch (first call arms them). */
int dpq_debug_boot_stamps(dpq_index* idx, int nq, double* out);
/* The level-0 select alone (shards without a bootstrap). */
int dpq_debug_select_time(dpq_index* idx, int nq, int top_k, int flags, int reps, float* ms_out);
/* Per-wavefront marks of strand1_kernel on the 100 MHz clock, [256][16][16] words (first call arms them). */
int dpq_debug_strand1_stamps(dpq_index* idx, unsigned long long* out, int n_words);

#ifdef __cplusplus
}
#endif
#endif /* DELTAPQ_AMD_H */
