// dpq_capi.cpp -- the C-ABI of include/deltapq_amd.h: index lifetime, the
// threshold-cascade driver around the scan/select kernels, profiling.
// Compiled with hipcc (HIP runtime calls only; kernels live in dpq_kernels.hip).
//
// There is no CPU implementation of the query in this library: every
// dpq_query_* call runs the HIP kernels or fails with an error code.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/deltapq_amd.h"
#include "dpq_build.h"
#include "dpq_format.h"
#include "dpq_kernels.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

// No C++ exception may cross the C ABI (a corrupt header can ask for a multi-GB vector): every
// extern "C" entry point runs its body through guarded().
template <class F>
int guarded(F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(DPQ_ERR_NOMEM, "out of host memory");
    } catch (const std::length_error& e) {
        return fail(DPQ_ERR_NOMEM, std::string("allocation size out of range: ") + e.what());
    } catch (const std::exception& e) {
        return fail(DPQ_ERR_STATE, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(DPQ_ERR_STATE, "internal error: unknown exception");
    }
}

#define DPQ_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            return fail(DPQ_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
        }                                                                                          \
    } while (0)

template <class T>
int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
    if (e != hipSuccess) return fail(DPQ_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return DPQ_OK;
}

struct EventPair {
    int kind;  // 0 lut, 1 scan, 2 select, 3 quantise, 4 per-batch decode, 5 bootstrap
    hipEvent_t a, b;
};

constexpr int kMaxBatchQueries = 2048;

}  // namespace

struct dpq_soa {
    dpq::SoA soa;
};

struct dpq_tree {
    dpq::Tree tree;
};

// Plan and tiling knobs of a handle: dpq_open_opts' fields with the defaults filled in (resolve_tuning).
struct Tuning {
    // batches up to this size take stream_kernel (1, 2 or 4 queries per pass).  Measured at 125 M codes (ms per call,
    // stream / 64-query filter path): 1 query 0.64 / 1.26, 2: 0.66 / 1.22, 4: 0.85 / 1.22, 8 (two passes of four, or one of
    // eight at 8 wavefronts per CU): 1.7-2.3 / 1.21, 16: 4.3 / 1.23 -- the switch-over sits behind four
    int stream_max = 4;
    int coarse_below = 128;
    int plan_ratios[3] = {0, 0, 0};
    int boot_cap = 0, boot_target = 0;
    int boot_variant = 1;  // bootstrap_kernel's V (dpq_kernels.hip); 0 = the kernel of rounds 2 - 3  [DPQ_BOOT_VARIANT]
    int select_threads = 0;  // select_kernel block size, 0 = by top_k  [DPQ_SELECT_THREADS]
    int select_fast = 1;     // the last level as a bucket sort from one histogram pass (select_kernel)  [DPQ_SELECT_FAST=0: radix select + rank count / bitonic network]
    int64_t batch_tile_nodes = (int64_t)16 << 20;
    bool relabel = true, fuse_quantise = true, async_overlap = true, boot_fullsort = false, tighten = true, strands = true,
         force_strands = false, strand1 = true;
    int s1_debug = 0;  // developer experiments of strand1_kernel  [DPQ_S1_DEBUG]
    // a lane's scan waits for the other lane's previous select: 0 = never (default), 1 = only for the second batch of a
    // burst, 2 = every batch  [DPQ_LANE_GATE].  Measured (scripts/gpu_lane_gate.sh, same box, M q/s): 0: 6.24 / 6.48,
    // 1: 6.46 / 6.30, 2: 5.89 -- the gate turns the lanes' lockstep (both bootstraps, then both scans, then both selects:
    // profiles/r04_timeline_default.txt) into the staggered order (profiles/r04_timeline_lane_gate.txt: a step = scan +
    // select beside the other lane's bootstrap + a 12 us event wait) and the step does not get shorter: kept as a knob.
    int lane_gate = 0;
    int s1_scatter = 0;  // ... its one level in the low-discrepancy strip order instead of storage order  [DPQ_S1_SCATTER=1]
    bool dummy_ = false;  // strand1: one query per pass takes strand1_kernel  [DPQ_STRAND1=0: strand_kernel<1>]
};

struct dpq_index {
    Tuning tune;
    int device = 0;
    int M = 8, K = 256, Ds = 0;
    int cap = 0;
    bool cap_auto = true;
    dpq_info info{};
    dpq::DeviceImage img;
    // owned device memory of the image
    // strand image (dpq_format.h): the stream pass's own layout of the same nodes (M = 8, shards with a bootstrap)
    uint64_t* d_st_ckpt = nullptr;
    uint32_t* d_st_mask = nullptr;
    uint16_t* d_st_depth = nullptr;
    uint32_t *d_st_pbase = nullptr, *d_strip_order = nullptr;
    uint32_t* d_strip_segs = nullptr;      // the segments of the strips, in strip visiting order (a level too small for the
    std::vector<int64_t> strip_seg_off;    // strand pass runs the chunk-per-wavefront pass over ITS strips' segments)
    uint8_t* d_st_delta = nullptr;
    int64_t strand_bytes = 0;
    uint8_t *d_nib = nullptr, *d_par = nullptr, *d_carry = nullptr, *d_mask = nullptr, *d_delta = nullptr, *d_ckpt = nullptr,
            *d_raw = nullptr;
    bool plain = false;  // uncompressed comparator index (fp32-accumulate rule, no id quirk)
    uint64_t* d_seg_off = nullptr;
    // threshold bootstrap: inverted multi-index over the shard's nodes (dpq::SoA::mi_*); boot = it is in use
    uint32_t *d_mi_cell = nullptr, *d_mi_code = nullptr, *d_mi_id = nullptr;
    bool boot = false;
    int boot_classes = 0;
    unsigned long long* d_boot_stamps = nullptr;  // developer diagnostics (dpq_debug_boot_stamps)
    unsigned long long* d_s1_stamps = nullptr;    // developer diagnostics (dpq_debug_strand1_stamps)
    uint8_t* d_batch_raw = nullptr;  // active lane: the shard's plain codes, decoded once per batch (batch_decode)
    int batch_decode = 0;            // dpq_open_opts.batch_decode
    uint8_t* d_relabel = nullptr;    // [M][256] code value -> label in the plain-code scratch (bank-aware; NULL = code values)
    uint8_t* d_nbr = nullptr;        // [8][256][256] centroid neighbour lists of the bootstrap's sub-spaces (dpq_set_codebook)
    float* d_codebook = nullptr;
    // workspace, sized for ws_slots padded queries and ws_cap candidates each
    int ws_slots = 0, ws_cap = 0;
    float* d_lut32 = nullptr;       // exact tables [query][8][256]
    float* d_lut32r = nullptr;      // the same, rows by the labels of the plain-code scratch (only with d_relabel)
    float* d_lut_min = nullptr;     // [query][8] minima (anchor of the filter quantisation)
    uint4* d_qtab = nullptr;        // [slot groups][128 KB] filter tables of the cascade level being scanned
    uint32_t *d_cand_count = nullptr, *d_overflow = nullptr;
    uint64_t *d_cand_key = nullptr, *d_thr_key = nullptr;  // candidate keys [slots][ws_cap], threshold keys [slots]
    unsigned long long* d_counters = nullptr;  // [2] scan statistics (dpq_profile.exact_checks / candidates)
    uint64_t* d_scratch = nullptr;   // [slots][ws_cap] contiguous copy of a slot's keys when they exceed the select's LDS list
    uint32_t* h_overflow = nullptr;  // pinned
    // pinned + mapped words, one per batch in flight: set by select_kernel when any query of the batch overflowed
    static constexpr int kFlagSlots = 64;
    uint32_t* h_any = nullptr;       // [kFlagSlots]; slot 0 serves the synchronous calls
    uint32_t* d_any = nullptr;       // device address of h_any
    struct Pending {                 // a batch enqueued by dpq_query_batch_device_async and not yet finished
        const float* d_queries;
        int nq, top_k;
        int32_t* d_ids;
        float* d_dists;
        hipStream_t stream;          // the stream it runs on (the caller's, or one of the two lane streams)
        hipStream_t user_stream;     // the stream the caller enqueued it on
        int flag_slot;
        int host_slot = -1;          // >= 0: a dpq_query_batch_host_async batch staged in host_slots[host_slot]
    };
    // dpq_query_batch_host_async: up to kHostSlots batches in flight, each with its own staging buffers; queries go up
    // on copy_in (the lanes' batches wait for it), results come down on copy_out behind the batch's last kernel
    static constexpr int kHostSlots = 16;
    struct HostSlot {
        float* d_q = nullptr;
        int32_t* d_ids = nullptr;
        float* d_d = nullptr;
        size_t qf = 0, oe = 0;       // capacities (floats / elements)
        int32_t* h_ids = nullptr;    // the caller's buffers of the batch in flight
        float* h_d = nullptr;
        size_t n_out = 0;
        bool busy = false, redo = false;
        bool direct = false;         // the result buffers are page-locked and mapped: the select kernel writes them itself
        hipEvent_t kernels_done = nullptr;
    };
    HostSlot host_slots[kHostSlots];
    hipStream_t copy_in = nullptr, copy_out = nullptr;
    uint64_t host_seq = 0;
    // Pipelined batches alternate between two LANES = two workspaces + two internal streams, so that a batch's
    // table build runs under the previous batch's scan (the scan fills every CU's LDS and half its wave slots:
    // the LUT kernel needs neither) and its bootstrap next to the previous batch's select.  The d_* workspace
    // fields above are the ACTIVE lane's; the other lane is parked here.
    struct Lane {
        float *d_lut32 = nullptr, *d_lut_min = nullptr, *d_lut32r = nullptr;
        uint4* d_qtab = nullptr;
        uint32_t *d_cand_count = nullptr, *d_overflow = nullptr;
        uint64_t *d_cand_key = nullptr, *d_thr_key = nullptr, *d_scratch = nullptr;
        uint8_t* d_batch_raw = nullptr;
        int ws_slots = 0, ws_cap = 0;
    };
    Lane parked;                     // workspace of the lane that is not active
    int active_lane = 0;
    hipStream_t lane_stream[2] = {nullptr, nullptr};
    hipEvent_t lane_ready[2] = {nullptr, nullptr};   // recorded on the caller's stream: the batch's inputs are there
    // recorded behind a laned batch's last select: the OTHER lane's next scan waits for it.  Without it the two lanes fall
    // into lockstep (kernel-trace timeline, profiles/r04_timeline_*): both bootstraps side by side, then both scans one
    // after the other -- the second scan's workgroups take every CU the first one frees, and the first batch's select
    // (35 KB of LDS) starves until the second scan's tail: 2 scans + select + bootstrap per 2 steps.  With the wait a step is
    // scan + max(select, the other lane's bootstrap).
    hipEvent_t lane_select_done[2] = {nullptr, nullptr};
    bool lane_select_recorded[2] = {false, false};
    int run_lane = -1;               // >= 0 while run_batch enqueues a laned batch: its lane
    bool gate_this_batch = false;    // the batch being enqueued is the second of a burst (one batch in flight, on the other lane)
    uint64_t async_seq = 0;
    hipStream_t ordered_stream[2] = {nullptr, nullptr};  // caller streams with stream-ordered batches: stream k <-> workspace k
    bool ordered_stream_set[2] = {false, false};
    int64_t finish_reruns = 0;       // batches dpq_finish had to answer again (a query overflowed its candidate buffers)
    std::vector<Pending> pending;
    // staging for the host-pointer entry point
    float* d_q_stage = nullptr;
    int32_t* d_ids_stage = nullptr;
    float* d_dists_stage = nullptr;
    size_t q_stage_floats = 0, out_stage_elems = 0;
    // cascade plan: visiting order of the segments, level bounds, decoded level 0
    int plan_top_k = -1, plan_cap = -1, plan_coarse = -1;
    std::vector<int> level_off, level_cnt;
    uint32_t* d_order = nullptr;
    uint32_t *d_l0_id = nullptr, *d_l0_code = nullptr;
    int l0_segments = 0;
    // profiling
    bool prof = false;
    bool prof_scan_only = false;     // events around the scan launches only (each event pair costs ~4 us of stream time)
    std::vector<EventPair> events;
    std::vector<hipEvent_t> ev_pool;
    dpq_profile prof_acc{};
    hipEvent_t get_event() {
        if (!ev_pool.empty()) {
            hipEvent_t e = ev_pool.back();
            ev_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        // timing events between kernels of one stream: no system-scope fence needed (saves ~2 us per record)
        if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) {
            prof_failed = true;
            return nullptr;
        }
        return e;
    }
    bool prof_failed = false;        // an event could not be created / recorded: dpq_profile_read reports it
    // developer hooks (dpq_debug_scan_time mode 3): the last batch's bootstrap and first-level scan launches as they were
    dpq::BootArgs dbg_ba{};
    dpq::ScanArgs dbg_sa{};
    int dbg_boot_slots = 0, dbg_groups = 0, dbg_splits = 0;
};

namespace {

// Developer diagnostics (the dpq_debug_* entry points, include/deltapq_amd.h) and developer environment variables are
// live only in a process started with DPQ_DEV=1.
bool dev_mode() {
    const char* dev = getenv("DPQ_DEV");
    return dev && atoi(dev) != 0;
}
#define DPQ_DEV_ONLY() \
    if (!dev_mode()) return fail(DPQ_ERR_STATE, "developer diagnostics: start the process with DPQ_DEV=1")

// dpq_open_opts -> Tuning.  The environment takes part only with DPQ_DEV=1 (developer sweeps: scripts/), and only
// here, once per dpq_open_*: a product process' plan never depends on its environment.
Tuning resolve_tuning(const dpq_open_opts& o) {
    Tuning t;
    if (o.stream_max_queries != 0) t.stream_max = std::min(16, std::max(0, o.stream_max_queries));  // (four queries per pass at most)
    if (o.coarse_below > 0) t.coarse_below = o.coarse_below;
    for (int i = 0; i < 3; ++i) t.plan_ratios[i] = o.plan_ratios[i];
    t.boot_cap = std::max(0, o.boot_cap);
    t.boot_target = std::max(0, o.boot_target);
    if (o.batch_tile_nodes > 0) t.batch_tile_nodes = o.batch_tile_nodes;
    t.relabel = !(o.flags & DPQ_OPT_NO_RELABEL);
    t.fuse_quantise = !(o.flags & DPQ_OPT_NO_FUSE_QUANTISE);
    t.async_overlap = !(o.flags & DPQ_OPT_NO_ASYNC_OVERLAP);
    t.boot_fullsort = (o.flags & DPQ_OPT_BOOT_FULLSORT) != 0;
    t.tighten = !(o.flags & DPQ_OPT_NO_TIGHTEN);
    t.strands = !(o.flags & DPQ_OPT_NO_STRANDS);
    t.force_strands = (o.flags & DPQ_OPT_FORCE_STRANDS) != 0;
    t.strand1 = !(o.flags & DPQ_OPT_NO_STRAND1);
    if (dev_mode()) {
        auto geti = [](const char* name, int* v) { if (const char* e = getenv(name)) *v = atoi(e); };
        geti("DPQ_STREAM_MAX_QUERIES", &t.stream_max);
        geti("DPQ_COARSE_BELOW", &t.coarse_below);
        if (const char* e = getenv("DPQ_PLAN_RATIOS")) sscanf(e, "%d,%d,%d", &t.plan_ratios[0], &t.plan_ratios[1], &t.plan_ratios[2]);
        geti("DPQ_BOOT_CAP", &t.boot_cap);
        geti("DPQ_BOOT_TARGET", &t.boot_target);
        geti("DPQ_BOOT_VARIANT", &t.boot_variant);
        if (const char* e = getenv("DPQ_BATCH_TILE_NODES")) t.batch_tile_nodes = std::max<int64_t>(1, atoll(e));
        int v = 1;
        geti("DPQ_RELABEL", &v); t.relabel = t.relabel && v != 0;
        v = 1; geti("DPQ_FUSE_QUANTISE", &v); t.fuse_quantise = t.fuse_quantise && v != 0;
        v = 1; geti("DPQ_ASYNC_OVERLAP", &v); t.async_overlap = t.async_overlap && v != 0;
        v = 0; geti("DPQ_BOOT_FULLSORT", &v); t.boot_fullsort = t.boot_fullsort || v != 0;
        v = 1; geti("DPQ_TIGHTEN", &v); t.tighten = t.tighten && v != 0;
        geti("DPQ_SELECT_THREADS", &t.select_threads);
        geti("DPQ_SELECT_FAST", &t.select_fast);
        v = 1; geti("DPQ_STRANDS", &v); t.strands = t.strands && v != 0;
        t.force_strands = t.force_strands || v == 2;
        v = 1; geti("DPQ_STRAND1", &v); t.strand1 = t.strand1 && v != 0;
        geti("DPQ_S1_DEBUG", &t.s1_debug);
        geti("DPQ_S1_SCATTER", &t.s1_scatter);
        geti("DPQ_LANE_GATE", &t.lane_gate);
    }
    return t;
}

void switch_lane(dpq_index* x, int lane) {
    if (lane == x->active_lane) return;
    x->dbg_groups = x->dbg_boot_slots = 0;  // (dpq_debug_scan_time mode 3 replays launches of the ACTIVE workspace)
    dpq_index::Lane cur;
    cur.d_lut32 = x->d_lut32; cur.d_lut_min = x->d_lut_min; cur.d_qtab = x->d_qtab; cur.d_lut32r = x->d_lut32r;
    cur.d_cand_count = x->d_cand_count; cur.d_overflow = x->d_overflow;
    cur.d_cand_key = x->d_cand_key; cur.d_thr_key = x->d_thr_key; cur.d_scratch = x->d_scratch;
    cur.ws_slots = x->ws_slots; cur.ws_cap = x->ws_cap; cur.d_batch_raw = x->d_batch_raw;
    const dpq_index::Lane& o = x->parked;
    x->d_lut32 = o.d_lut32; x->d_lut_min = o.d_lut_min; x->d_qtab = o.d_qtab; x->d_lut32r = o.d_lut32r;
    x->d_cand_count = o.d_cand_count; x->d_overflow = o.d_overflow;
    x->d_cand_key = o.d_cand_key; x->d_thr_key = o.d_thr_key; x->d_scratch = o.d_scratch;
    x->ws_slots = o.ws_slots; x->ws_cap = o.ws_cap; x->d_batch_raw = o.d_batch_raw;
    x->parked = cur;
    x->active_lane = lane;
}

void free_parked_lane(dpq_index* x) {
    dpq_index::Lane& o = x->parked;
    hipFree(o.d_lut32); hipFree(o.d_lut_min); hipFree(o.d_qtab); hipFree(o.d_cand_count); hipFree(o.d_overflow);
    hipFree(o.d_lut32r);
    hipFree(o.d_cand_key); hipFree(o.d_thr_key); hipFree(o.d_scratch); hipFree(o.d_batch_raw);
    o = dpq_index::Lane();
}

void free_workspace(dpq_index* x) {
    x->dbg_groups = x->dbg_boot_slots = 0;  // the recorded launch arguments point into what is freed here
    hipFree(x->d_lut32);
    hipFree(x->d_lut32r);
    x->d_lut32r = nullptr;
    hipFree(x->d_lut_min);
    hipFree(x->d_qtab);
    hipFree(x->d_cand_count);
    hipFree(x->d_cand_key);
    hipFree(x->d_scratch);
    hipFree(x->d_overflow);
    hipFree(x->d_thr_key);
    x->d_lut32 = nullptr;
    x->d_lut_min = nullptr;
    x->d_qtab = nullptr;
    x->d_cand_count = x->d_overflow = nullptr;
    x->d_cand_key = x->d_thr_key = x->d_scratch = nullptr;
    x->ws_slots = x->ws_cap = 0;
}

int ensure_workspace(dpq_index* x, int slots, int cap) {
    if (slots <= x->ws_slots && cap <= x->ws_cap) return DPQ_OK;
    slots = std::max(slots, x->ws_slots);
    cap = std::max(cap, x->ws_cap);
    free_workspace(x);
    int rc;
    if ((rc = dev_alloc(&x->d_lut32, (size_t)slots * x->M * 256))) return rc;
    if (x->d_relabel && (rc = dev_alloc(&x->d_lut32r, (size_t)slots * x->M * 256))) return rc;
    if ((rc = dev_alloc(&x->d_lut_min, (size_t)slots * x->M * 4))) return rc;  // four partial minima per (query, m)
    if ((rc = dev_alloc(&x->d_qtab, (size_t)(slots / dpq::queries_per_group(x->M) + 1) *
                                        (dpq::qtab_bytes_per_group(x->M) / sizeof(uint4)))))
        return rc;
    if ((rc = dev_alloc(&x->d_cand_count, (size_t)slots * dpq::kRegionStride))) return rc;
    if ((rc = dev_alloc(&x->d_cand_key, (size_t)slots * cap))) return rc;
    if ((rc = dev_alloc(&x->d_scratch, (size_t)slots * cap))) return rc;
    // [slots] overflow flags, then [slots][kTightWords] tightening counters of the level being scanned
    if ((rc = dev_alloc(&x->d_overflow, (size_t)slots * (1 + dpq::kTightWords)))) return rc;
    if ((rc = dev_alloc(&x->d_thr_key, (size_t)slots))) return rc;
    if (!x->h_overflow) DPQ_HIP(hipHostMalloc(reinterpret_cast<void**>(&x->h_overflow), sizeof(uint32_t) * 4096));
    if (!x->d_counters) {
        if ((rc = dev_alloc(&x->d_counters, 2))) return rc;
        DPQ_HIP(hipMemset(x->d_counters, 0, 16));
    }
    if (!x->h_any) {
        DPQ_HIP(hipHostMalloc(reinterpret_cast<void**>(&x->h_any), sizeof(uint32_t) * dpq_index::kFlagSlots,
                              hipHostMallocMapped));
        DPQ_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&x->d_any), x->h_any, 0));
    }
    x->ws_slots = slots;
    x->ws_cap = cap;
    return DPQ_OK;
}

int auto_cap(int top_k) { return std::max(4096, 32 * top_k); }

// Progressive cascade plan.  Segments are visited in a low-discrepancy order
// (so every prefix is a spread-out sample of the DFS stream); level l covers
// order[bound[l-1] : bound[l]] -- every segment exactly once over the whole
// cascade.  Level 0 (<= 4096 nodes) is query independent: its segments are
// decoded once here and every query evaluates that list exactly.  Later levels
// are filter scans; expected survivors of level l = top_k * (bound[l]/bound[l-1] - 1).
// Small batches (coarse == 1) are bound by the fixed cost per level (a launch + a select), not by survivor
// handling, so they use steps of 16 and end up with three levels instead of five.
int ensure_plan(dpq_index* x, int top_k, int cap, int shape) {
    if (x->plan_top_k == top_k && x->plan_cap == cap && x->plan_coarse == shape) return DPQ_OK;
    const int coarse = shape & 1;
    const bool tight = (shape & 2) != 0;  // the scan tightens its thresholds as it goes (run_batch)
    const bool one_level = (shape & 4) != 0;  // one query per pass on the strand image: strand1_kernel tightens in the kernel
    const int64_t S = (int64_t)dpq::kChunk * x->img.chunks_per_segment;
    const int64_t nseg = x->img.n_segments;
    const int64_t s0 = std::min<int64_t>(nseg, std::max<int64_t>(1, dpq::kLevel0Nodes / S));
    std::vector<int64_t> bounds;
    if (x->boot) {
        // The bootstrap kernel delivers the first threshold (no segments consumed: level 0 is empty), as tight
        // as the k-th of a spread sample of a quarter of a 1 M-node index; the filter levels then cover ALL
        // segments.  One level up to 2 M nodes; beyond, levels growing by 8 (a larger shard's first threshold
        // admits more nodes in absolute terms); top_k > 512: see below.  dpq_open_opts.plan_ratios forces levels.
        bounds.push_back(nseg);
        std::vector<int> ratios;
        const int* forced = x->tune.plan_ratios;
        if (forced[0] >= 2) {
            for (int i = 0; i < 3; ++i)
                if (forced[i] >= 2) ratios.push_back(forced[i]);
        } else if (forced[0] == 1 || one_level) {
            // one level whatever the shard size (experiments; the one-query strand pass)
        } else {
            for (int64_t b = nseg; b * S > ((int64_t)2 << 20); b /= 8) ratios.push_back(8);
            // A large top_k takes its first threshold from a worse quantile of the bootstrap sample (the 1000th of
            // 8 K nodes): two short levels in front tighten it before the bulk of the index is filtered.  Measured
            // on 1 M codes (scripts/gpu_sweep_plans.sh): top-1000 1.40 M q/s with 3,3 against 0.95 M with one level
            // (M = 16: 0.79 M against 0.40 M); top-300 and below are fastest with one level.
            // With in-scan tightening the scan does that itself and one level wins again: top-1000 2.27 M q/s
            // against 1.89 M with 3,3 (M = 16: 1.27 M against 1.21 M), top-2048 1.19 M against 0.97 M.
            if (top_k > 512 && ratios.size() < 2 && !(tight && ratios.empty())) ratios = {3, 3};
        }
        int64_t b = nseg;
        for (int r : ratios) {
            b /= r;
            if (b < 1) break;
            bounds.push_back(b);
        }
        bounds.push_back(0);
        std::reverse(bounds.begin(), bounds.end());
    } else if (s0 >= nseg) {
        bounds.push_back(nseg);
    } else {
        bounds.push_back(nseg);
        // large batches: level sizes shrink by 4, 8, 8 from the full index down at top-100 (three filter levels
        // at 1 M codes); measured best trade between per-level fixed cost and survivor handling.
        // dpq_open_opts.plan_ratios overrides it for experiments.
        // A level costs a fixed ~45 us (scan prologue, launch, select) plus ~top_k * (ratio - 1) candidates per
        // query to check and select: the best ratio falls with top_k (measured: 8 at top-100, 3 at top-1000).
        const int r = (int)std::lround(std::min(16.0, std::max(2.0, 8.0 * std::sqrt(100.0 / (double)top_k))));
        int fine[] = {r >= 16 ? r : std::max(2, r / 2), r, r};  // few candidates (small top_k): the fewest levels win
        for (int i = 0; i < 3; ++i)
            if (x->tune.plan_ratios[i] >= 2) fine[i] = x->tune.plan_ratios[i];
        for (int& f : fine) f = std::max(2, f);
        // expected survivors of a level = top_k * (ratio - 1) must stay well inside the candidate buffer
        const int wide = (int)std::max<int64_t>(2, std::min<int64_t>(16, cap / (2 * (int64_t)top_k)));
        int64_t b = nseg;
        for (int i = 0;; ++i) {
            const int64_t nb = b / (coarse ? wide : fine[std::min(i, 2)]);
            if (nb < 2 * s0) break;
            bounds.push_back(nb);
            b = nb;
        }
        bounds.push_back(s0);
        std::reverse(bounds.begin(), bounds.end());
    }
    x->level_off.clear();
    x->level_cnt.clear();
    int64_t prev = 0;
    for (int64_t bnd : bounds) {
        x->level_off.push_back((int)prev);
        x->level_cnt.push_back((int)(bnd - prev));
        prev = bnd;
    }
    if (!x->d_order && nseg > 0) {
        // order[j] = j * P mod nseg, P ~ nseg / golden ratio, gcd(P, nseg) = 1
        std::vector<uint32_t> order((size_t)nseg);
        int64_t P = std::max<int64_t>(1, (int64_t)((double)nseg * 0.6180339887498949));
        auto gcd = [](int64_t a, int64_t b) { while (b) { int64_t t = a % b; a = b; b = t; } return a; };
        while (gcd(P, nseg) != 1) ++P;
        for (int64_t j = 0; j < nseg; ++j) order[(size_t)j] = (uint32_t)((j * P) % nseg);
        int rc = dev_alloc(&x->d_order, (size_t)nseg);
        if (rc) return rc;
        DPQ_HIP(hipMemcpy(x->d_order, order.data(), (size_t)nseg * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    if (!x->boot && nseg > 0 && x->l0_segments != (int)s0) {
        hipFree(x->d_l0_id);
        hipFree(x->d_l0_code);
        x->d_l0_id = x->d_l0_code = nullptr;
        int rc = dev_alloc(&x->d_l0_id, (size_t)(s0 * S));
        if (!rc) rc = dev_alloc(&x->d_l0_code, (size_t)(s0 * S * (x->M / 4)));
        if (rc) return rc;
        DPQ_HIP(dpq::launch_decode_segments(x->img, x->d_order, (int)s0, x->d_l0_id, x->d_l0_code, nullptr));
        DPQ_HIP(hipStreamSynchronize(nullptr));
        x->l0_segments = (int)s0;
    }
    x->plan_top_k = top_k;
    x->plan_cap = cap;
    x->plan_coarse = shape;
    return DPQ_OK;
}

struct Timer {
    dpq_index* x;
    hipStream_t s;
    int kind;
    bool on;
    EventPair ep{};
    Timer(dpq_index* x_, hipStream_t s_, int kind_)
        : x(x_), s(s_), kind(kind_), on(x_->prof && (kind_ == 1 || !x_->prof_scan_only)) {
        if (on) {
            ep.kind = kind;
            ep.a = x->get_event();
            ep.b = x->get_event();
            if (!ep.a || !ep.b || hipEventRecord(ep.a, s) != hipSuccess) {
                x->prof_failed = true;
                if (ep.a) x->ev_pool.push_back(ep.a);
                if (ep.b) x->ev_pool.push_back(ep.b);
                on = false;
            }
        }
    }
    ~Timer() {
        if (on) {
            if (hipEventRecord(ep.b, s) != hipSuccess) x->prof_failed = true;
            x->events.push_back(ep);
        }
    }
};

// plain-code scratch of a shard (per pipeline lane), and whether a batch of n_groups query groups uses it
// Plain-code scratch of a batch (per pipeline lane): one TILE of a filter level's segment list at a time -- decode the
// tile, scan it with every query group, next tile -- so the scratch stays Infinity-Cache-sized whatever the shard
// (16 M nodes = 128 MB at M = 8; the 1 M-code headline and a 12.5 M-code shard are one tile, a 125 M-code shard's
// last level eight).  dpq_open_opts.batch_tile_nodes overrides; dpq_open_opts.batch_decode >= 2 = tile of that many segments.
int64_t batch_tile_segments(const dpq_index* x) {
    const int64_t S = (int64_t)dpq::kChunk * x->img.chunks_per_segment;
    int64_t t = x->batch_decode >= 2 ? x->batch_decode : std::max<int64_t>(1, x->tune.batch_tile_nodes / S);
    return std::min<int64_t>(t, std::max<int64_t>(1, x->img.n_segments));
}
int64_t batch_raw_bytes(const dpq_index* x) {
    return batch_tile_segments(x) * dpq::kChunk * x->img.chunks_per_segment * x->M;
}
bool batch_decode_possible(const dpq_index* x) { return !x->plain && x->batch_decode >= 0 && x->img.n_segments > 0; }
bool use_batch_decode(const dpq_index* x, int n_groups) {
    return batch_decode_possible(x) && (x->batch_decode > 0 || n_groups >= 3);
}
int ensure_batch_raw(dpq_index* x) {
    if (x->d_batch_raw) return DPQ_OK;
    return dev_alloc(&x->d_batch_raw, (size_t)batch_raw_bytes(x));
}

int splits_for(int n_seg_pass, int n_groups) {
    // One workgroup per CU is resident (LDS), so aim at ONE chip-wave: <= 256 workgroups.  Small levels
    // still spread over as many CUs as they have segments: the exact checks of the filter survivors
    // are bound by each CU's vector-memory address rate, not by its wavefront count.
    const int want = std::max(1, dpq::kMaxSplits / std::max(1, n_groups));
    return std::max(1, std::min(n_seg_pass, want));
}

// Candidate-buffer geometry of one scan launch: region 0 (top_k keys) carries the winners of the
// previous level, then one region per scan workgroup of a query group (no global atomics: a
// workgroup appends to its own region).
struct Regions {
    int splits;
    int region_cap;
    int64_t stride;  // keys per slot
};

Regions regions_for(const dpq_index* x, int n_seg_pass, int n_groups, int top_k, int cap) {
    Regions r;
    r.splits = splits_for(n_seg_pass, n_groups);
    r.region_cap = (cap - top_k) / r.splits;
    // automatic sizing: a query's candidates cluster in few segments (DFS neighbours are similar codes), so a
    // region must absorb a few dense segments; 16 K keys per slot, at least one segment's worth per region,
    // and four times a level's expected candidates (top_k x 16) spread over the regions
    if (x->cap_auto)
        r.region_cap = std::max(r.region_cap, std::max(std::max(256, 16384 / r.splits), 64 * top_k / r.splits));
    r.region_cap = std::max(r.region_cap, 1);
    r.stride = (int64_t)top_k + (int64_t)r.splits * r.region_cap;
    return r;
}

// One sub-batch (nq <= kMaxBatchQueries) end to end on `stream`.
// flag_slot 0: synchronous (waits, checks the overflow word, reruns what overflowed); > 0: enqueue only,
// dpq_finish looks at the word later.
int run_batch(dpq_index* x, const float* d_queries, int nq, int top_k, int32_t* d_ids, float* d_dists,
              hipStream_t stream, int flag_slot = 0) {
    const int QG = dpq::queries_per_group(x->M);
    const int nqp = (nq + QG - 1) / QG * QG;
    const int ngroups = nqp / QG;
    const int cap = x->cap_auto ? auto_cap(top_k) : std::max(x->cap, top_k);
    int rc;
    // In-scan tightening is live when the scan runs its plain-code instantiation with at most kTightSplits
    // workgroups per query group (8 groups or more): the plan then keeps one filter level also for a large top_k.
    const bool tight_plan = x->tune.tighten && nq > x->tune.stream_max &&
                            ngroups * dpq::kTightSplits >= dpq::kMaxSplits && (x->plain || use_batch_decode(x, ngroups));
    // Which stream pass a batch of up to stream_max queries takes: the strand image (a lane per run of 64 nodes) has 64 x
    // fewer, 64 x longer work items than the chunk-per-wavefront decode, so it wants a big shard (dpq::kStrandMinNodes).
    // ONE query per pass runs strand1_kernel, which tightens its threshold while it runs: one level over the whole shard.
    const bool direct = !x->plain && nq <= x->tune.stream_max;
    const bool strands = direct && x->img.st_ckpt != nullptr && (x->tune.force_strands || x->img.n_local >= dpq::kStrandMinNodes);
    const bool strand1 = strands && x->tune.strand1 && dpq::stream_queries_per_pass(x->M, nq) == 1;
    const bool strand1_tight = strand1 && x->tune.tighten && x->tune.plan_ratios[0] == 0;
    if ((rc = ensure_plan(x, top_k, cap, (nq <= x->tune.coarse_below ? 1 : 0) | (tight_plan ? 2 : 0) | (strand1_tight ? 4 : 0)))) return rc;
    int64_t stride = top_k;
    for (size_t l = 1; l < x->level_cnt.size(); ++l)
        stride = std::max(stride, regions_for(x, x->level_cnt[l], ngroups, top_k, cap).stride);
    // strand1_kernel: a region per workgroup (no global atomics on the candidates' way): room for top_k keys each, 64..256
    const int s1_region_cap = std::min(std::max(64, top_k), 256);
    if (strand1) stride = std::max<int64_t>(stride, (int64_t)top_k + (int64_t)dpq::kStrand1Regions * s1_region_cap);
    if (stride > INT32_MAX) return fail(DPQ_ERR_NOMEM, "candidate buffer too large");
    if ((rc = ensure_workspace(x, nqp, (int)stride))) return rc;
    stride = x->ws_cap;

    dpq::ScanArgs sa{};
    sa.img = x->img;
    sa.fp32_accum = x->plain ? 1 : 0;
    // Where the decode happens (dpq_open_opts.batch_decode): a batch of several query groups decodes the shard
    // once into plain codes that every group's filter pass reads (through L2 / Infinity Cache at the headline
    // sizes) -- the scan kernel then runs its plain-code instantiation with the DTC distance rule; a batch of one or
    // two groups, or a shard whose plain codes exceed the scratch budget, decodes inside the scan, once per group.
    // Measured on 1 M codes x 1000 queries (16 groups): scan 0.166 -> 0.122 ms, step 0.212 -> 0.172 ms.
    // One or two queries (the reference's own call shape): one query per pass over the compressed image, every node
    // evaluated against the exact table (stream_kernel) -- no filter tables, no 64-query group machinery.
    // dpq_open_opts.stream_max_queries: the batch size up to which this mode is used (-1 = never).
    const bool scratch = !direct && use_batch_decode(x, ngroups);
    if (scratch && (rc = ensure_batch_raw(x))) return rc;
    const int64_t tile_segs = scratch ? batch_tile_segments(x) : 0;
    // One tile covers the whole shard (the common case): it is decoded here, ahead of the table build, so that both
    // run under the previous pipelined batch's scan (neither needs LDS; the bootstrap that follows has to wait for
    // that scan's LDS anyway), and every level scans its slice of the shard-ordered scratch.  Otherwise each level
    // decodes its own tiles in list order.
    const bool one_tile = scratch && tile_segs >= x->img.n_segments;
    if (one_tile) {
        {
            Timer t(x, stream, 4);
            DPQ_HIP(dpq::launch_decode_list(x->img, nullptr, x->img.n_segments, x->d_relabel,
                                            reinterpret_cast<uint32_t*>(x->d_batch_raw), stream));
        }
        sa.img.raw = x->d_batch_raw;
    }
    // the scratch holds relabelled codes (bank-aware labels): the filter tables of this batch are laid
    // out by label, the scan maps a surviving node's code back before the exact check
    const bool labelled = scratch && x->d_relabel != nullptr;
    {
        Timer t(x, stream, 0);
        // also clears the overflow flags of the nqp slots
        // batches that scan the plain-code scratch also get the tables in the scratch's label order
        DPQ_HIP(dpq::launch_lut_build(x->d_codebook, d_queries, nq, nqp, x->M, x->K, x->Ds, x->d_lut32, x->d_lut_min,
                                      nullptr, x->d_overflow, scratch ? x->d_relabel : nullptr, x->d_lut32r,
                                      x->tune.tighten ? x->d_overflow + x->ws_slots : nullptr, stream));
    }
    if (x->prof) x->prof_acc.lut_launches++;
    x->h_any[flag_slot] = 0;  // the slot is free: its previous batch has been finished

    sa.lut32 = labelled ? x->d_lut32r : x->d_lut32;
    sa.lut_min = x->d_lut_min;
    sa.thr_key = x->d_thr_key;
    sa.slot_query = nullptr;
    sa.n_queries = nq;
    sa.cand_count = x->d_cand_count;
    sa.cand_key = x->d_cand_key;
    sa.cand_stride = stride;
    sa.region_off = top_k;
    sa.counters = x->prof && !x->prof_scan_only ? x->d_counters : nullptr;
    sa.qtab = x->d_qtab;
    // in-scan threshold tightening (plain-code scans of this batch; scan_kernel): histograms behind the overflow flags.
    // Measured (1 M codes x 1000 queries): top-100 +2 % queries/s (exact checks 3002 -> 1813 and candidates 802 -> 404 per
    // query, select 21 -> 16 us); top-300 +7 %, top-512 +22 %.  On top of the two short levels a large top_k used to get
    // it cost 2 % (M = 8) to 7 % (M = 16) -- but ONE level with the tightening beats those plans (top-1000 1.89 -> 2.27 M
    // q/s, M = 16 1.21 -> 1.27 M, top-2048 0.97 -> 1.19 M): ensure_plan gives top_k > 512 one level when the launch
    // tightens, and the tightening stays off only where such a top_k meets a plan of several levels (shards > 2 M codes).
    sa.tight_hist = x->tune.tighten && (top_k <= 512 || x->level_cnt.size() <= 2) ? x->d_overflow + x->ws_slots : nullptr;
    sa.tight_k = top_k;

    dpq::SelectArgs se{};
    se.threads = x->tune.select_threads;
    se.fast_final = x->tune.select_fast;
    se.cand_count = x->d_cand_count;
    se.cand_key = x->d_cand_key;
    se.cand_stride = stride;
    se.region_off = top_k;
    se.scratch = x->d_scratch;
    se.lut32 = x->d_lut32;
    se.slot_query = nullptr;
    se.top_k = top_k;
    se.thr_key = x->d_thr_key;
    se.overflow = x->d_overflow;
    se.any_overflow = x->d_any + flag_slot;
    se.out_ids = d_ids;
    se.out_dists = d_dists;
    se.n_codes_total = x->plain ? -1 : x->img.n_codes_total;
    se.fp32_accum = x->plain ? 1 : 0;
    se.keep_thr = x->boot ? 1 : 0;
    se.stamps = x->d_boot_stamps ? x->d_boot_stamps + (size_t)kMaxBatchQueries * 8 : nullptr;

    const int64_t S = (int64_t)dpq::kChunk * x->img.chunks_per_segment;
    const size_t n_levels = x->img.n_segments > 0 ? x->level_cnt.size() : 0;
    if (n_levels == 0) {  // empty shard: every row is padding
        se.final_pass = 1;
        DPQ_HIP(dpq::launch_select(se, x->M, nq, stream));
        return DPQ_OK;
    }
    bool boot_built_tables = false;
    bool scan_gate_passed = false;
    for (size_t l = 0; l < n_levels; ++l) {
        const bool final_pass = l + 1 == n_levels;
        if (l == 0 && x->boot) {
            // level 0 with a multi-index: the nodes of the query's best cells, evaluated exactly -> first threshold
            const int cap_env = x->tune.boot_cap;
            dpq::BootArgs ba{};
            ba.cell_start = x->d_mi_cell;
            const bool full_sort = x->tune.boot_fullsort;  // developer A/B
            ba.nbr = full_sort ? nullptr : x->d_nbr;
            ba.n_classes = x->boot_classes;
            ba.mi_code = x->d_mi_code;
            ba.mi_id = x->d_mi_id;
            ba.lut32 = x->d_lut32;
            ba.slot_query = nullptr;
            ba.top_k = top_k;
            // 4-byte keys: up to 6144 of them keep the block at 40 KB of LDS (four blocks per CU).  Measured at top-100,
            // M = 8 (scripts/gpu_boot_ab.sh): 3072 / 4096 / 6144 nodes -> 727 / 559 / 404 candidates per query and the
            // same step time within 1.5 % (what the scan saves the bootstrap spends); 3072 is the shortest critical path.
            // M = 16 (a class sees 2 of 16 sub-spaces: weaker cells; a check costs 16 gathers): 3072 / 6144 / 8192 ->
            // 3003 / 1373 / 1021 candidates, 1.77 / 1.99 / 2.00 M q/s (scripts/gpu_m16_boot.sh).  top_k > 256
            // (scripts/gpu_boot_cap1000.sh, top-1000): 8192 / 12288 / 16384 -> M = 8 1.81 / 1.83 / 1.81, M = 16
            // 1.05 / 1.10 / 1.00 M q/s.  Round 3, one level + in-scan tightening (scripts/gpu_boot_cap_large_k.sh),
            // 4096 / 6144 / 8192 / 12288: top-512 3.11 / 3.27 / 3.25 / 3.16, top-1000 1.76 / 1.96 / 2.10 / 2.28,
            // top-2048 0.74 / 0.97 / 1.06 / 1.20, M = 16 top-1000 1.04 / 1.17 / 1.22 / 1.27 M q/s.
            // (5888, not 6144, keys where the block is meant to stay at four per CU: 39 KB is what four blocks per CU take --
            // see select_list_keys; M = 16 holds 16 KB of tables and runs three per CU either way)
            const int cap_auto = top_k <= 256 ? (x->M <= 8 ? 3072 : 6144) : (top_k <= 640 && x->M <= 8) ? 5888 : 12288;
            ba.cap = std::max(std::min(cap_env > 0 ? cap_env : cap_auto, 16384), std::max(top_k, 2048));
            ba.cap = (ba.cap + 63) / 64 * 64;
            const int target_env = x->tune.boot_target;
            // Cells are walked in rounds until `target` nodes are evaluated.  At top_k <= 256 two thirds of the key list are
            // enough: the 1 % of the queries whose first round of cells brings fewer than `cap` nodes (sparse neighbourhoods:
            // 9 of 1000 on the bench index) then stop there instead of walking a second round -- they were the launch's
            // last blocks (dev_boot_stamps.py: span 25.4 -> 22.4 us), their thresholds come from >= 2048 nodes instead of
            // 3072 and the scan's own tightening does the rest (exact checks and candidates per query unchanged).
            const int target_auto = top_k <= 256 ? std::max(top_k, ba.cap * 2 / 3) : ba.cap;
            ba.target = target_env > 0 ? std::min(ba.cap, std::max(target_env, top_k)) : target_auto;
            ba.thr_key = x->d_thr_key;
            ba.cand_count = x->d_cand_count;
            ba.fp32_accum = x->plain ? 1 : 0;
            ba.stamps = x->d_boot_stamps;
            ba.variant = x->tune.boot_variant;
            ba.n_queries = nq;
            // DPQ_OPT_NO_FUSE_QUANTISE: the first level's tables from quantise_kernel, as for every later level
            const bool fuse = x->tune.fuse_quantise;
            ba.qtab = fuse && !direct ? x->d_qtab : nullptr;
            ba.relabel = scratch ? x->d_relabel : nullptr;
            ba.lut_min = x->d_lut_min;
            boot_built_tables = fuse && !direct;
            {
                Timer t(x, stream, 5);
                DPQ_HIP(dpq::launch_bootstrap(ba, x->M, fuse && !direct ? nqp : nq, stream));
            }
            if (x->prof) x->prof_acc.bootstrap_launches++;
            x->dbg_ba = ba;
            x->dbg_boot_slots = fuse && !direct ? nqp : nq;
            continue;
        }
        if (l == 0) {
            // level 0: the pre-decoded, query-independent list; every query evaluates it exactly
            se.shared_id = x->d_l0_id;
            se.shared_code = x->d_l0_code;
            se.shared_n = (int)(x->level_cnt[0] * S);
        } else {
            se.shared_id = nullptr;
            se.shared_code = nullptr;
            se.shared_n = 0;
            // a laned batch's first scan lets the other lane's previous batch finish its select first (see lane_select_done)
            if (x->run_lane >= 0 && !scan_gate_passed && (x->tune.lane_gate == 1 ? x->gate_this_batch : x->tune.lane_gate == 2)) {
                const int other = x->run_lane ^ 1;
                if (x->lane_select_recorded[other]) DPQ_HIP(hipStreamWaitEvent(stream, x->lane_select_done[other], 0));
                scan_gate_passed = true;
            }
            // filter scan of the next slice of segments; appends behind the carried winners
            sa.seg_list = x->d_order + x->level_off[l];
            sa.n_seg_pass = x->level_cnt[l];
            const Regions rg = regions_for(x, sa.n_seg_pass, ngroups, top_k, cap);
            if (direct) {
                // one region per slot behind the carried winners, filled through a global counter
                sa.region_cap = se.region_cap = (int32_t)std::min<int64_t>(stride - top_k, INT32_MAX);
                se.n_regions = 2;
                DPQ_HIP(hipMemset2DAsync(x->d_cand_count + 1, sizeof(uint32_t) * dpq::kRegionStride, 0, sizeof(uint32_t),
                                         (size_t)nq, stream));
                // Which stream pass: the strand image (a lane per run of 64 nodes) has 64 x fewer, 64 x longer work items than
                // the chunk-per-wavefront decode, so it wants a big shard.  Measured, us per call (strand / chunk), one query:
                // 1 M codes 35 / 37 pipelined but 89 / 59 as a single synchronous call, 4 M 50 / 49, 12.5 M 72 / 97, 32 M
                // 113 / 172, 125 M 343 / 619; four queries per pass: 1 M 55 / 47, 4 M 76 / 61, 12.5 M 103 / 121, 32 M
                // 195 / 230, 125 M 609 / 841.  From 8 M codes.
                if (strands) {
                    // the pass over the strand image: the level's share of the strips (every strip exactly once over
                    // the levels, like the segments; the bootstrap consumed none)
                    const int64_t nseg = x->img.n_segments, ns = x->img.n_strips;
                    const int64_t lo = (int64_t)x->level_off[l] * ns / nseg;
                    const int64_t hi = final_pass ? ns : ((int64_t)x->level_off[l] + x->level_cnt[l]) * ns / nseg;
                    dpq::ScanArgs st = sa;
                    // strand1_kernel's candidate histogram: the first 256 words of the tightening counters (cleared by the
                    // table build); one launch per batch reads it (a later level would count in other units)
                    st.tight_hist = strand1_tight && x->level_cnt.size() == 2 ? x->d_overflow + x->ws_slots : nullptr;
                    st.debug_pass = strand1 ? (x->tune.s1_debug ? 16 + x->tune.s1_debug : 0) : 3;
                    st.stamps = strand1 ? x->d_s1_stamps : nullptr;
                    Timer t(x, stream, 1);
                    // A level of few strips (one strip per wavefront: the launch takes a strip's 64 dependent steps
                    // however few there are) goes through the chunk-per-wavefront pass over the same nodes: measured
                    // break-even at about 1500 strips (6 M nodes).
                    if (hi - lo < 1536 && x->d_strip_segs && !x->tune.force_strands && !strand1) {
                        st.seg_list = x->d_strip_segs + x->strip_seg_off[(size_t)lo];
                        st.n_seg_pass = (int32_t)(x->strip_seg_off[(size_t)hi] - x->strip_seg_off[(size_t)lo]);
                        DPQ_HIP(dpq::launch_stream(st, nq, stream));
                        if (x->prof) x->prof_acc.stream_launches++;
                    } else {
                        st.seg_list = x->d_strip_order + lo;
                        st.n_seg_pass = (int32_t)(hi - lo);
                        if (strand1) {  // a region per workgroup, counts written by the kernel
                            // one level: the strips in storage order (a workgroup sweeps a contiguous share)
                            if (x->level_cnt.size() == 2 && !x->tune.s1_scatter) st.seg_list = nullptr;
                            st.region_cap = se.region_cap = (int32_t)((stride - top_k) / dpq::kStrand1Regions);
                            se.n_regions = 1 + dpq::strand1_workgroups(st.n_seg_pass);
                        }
                        DPQ_HIP(dpq::launch_strand(st, nq, stream));
                        if (x->prof) (strand1 ? x->prof_acc.strand1_launches : x->prof_acc.strand_launches)++;
                    }
                } else {
                    Timer t(x, stream, 1);
                    sa.tight_hist = nullptr;
                    DPQ_HIP(dpq::launch_stream(sa, nq, stream));
                    if (x->prof) x->prof_acc.stream_launches++;
                }
                if (x->prof) {
                    x->prof_acc.scan_launches++;
                    x->prof_acc.scan_stream_bytes +=
                        (int64_t)nq * (int64_t)((double)x->info.device_bytes * sa.n_seg_pass / std::max(1, x->img.n_segments));
                }
            } else {
            sa.region_cap = se.region_cap = rg.region_cap;
            se.n_regions = 1 + rg.splits;
            if (!(boot_built_tables && l == 1)) {  // the bootstrap kernel wrote the first level's tables itself
                Timer t(x, stream, 3);
                DPQ_HIP(dpq::launch_quantise(sa, ngroups, stream));
            }
            if (scratch && !one_tile) {
                // the level's list in tiles: decode a tile into the scratch (list order), scan it with all groups
                const uint32_t* list = sa.seg_list;
                const int total = sa.n_seg_pass;
                for (int t0 = 0; t0 < total; t0 += (int)tile_segs) {
                    const int cnt = std::min<int>((int)tile_segs, total - t0);
                    {
                        Timer t(x, stream, 4);
                        DPQ_HIP(dpq::launch_decode_list(x->img, list + t0, cnt, x->d_relabel,
                                                        reinterpret_cast<uint32_t*>(x->d_batch_raw), stream));
                    }
                    sa.img.raw = x->d_batch_raw;
                    sa.raw_by_pos = 1;
                    sa.append = t0 > 0 ? 1 : 0;
                    sa.seg_list = list + t0;
                    sa.n_seg_pass = cnt;
                    {
                        Timer t(x, stream, 1);
                        DPQ_HIP(dpq::launch_scan(sa, ngroups, rg.splits, stream));
                    }
                    if (x->prof && t0 > 0) x->prof_acc.scan_launches++;
                }
                sa.img.raw = nullptr;
                sa.raw_by_pos = sa.append = 0;
                sa.seg_list = list;
                sa.n_seg_pass = total;
            } else {
                Timer t(x, stream, 1);
                DPQ_HIP(dpq::launch_scan(sa, ngroups, rg.splits, stream));
                if (l == 1) {
                    x->dbg_sa = sa;
                    x->dbg_groups = ngroups;
                    x->dbg_splits = rg.splits;
                }
            }
            if (x->prof) {
                x->prof_acc.scan_launches++;
                x->prof_acc.scan_stream_bytes +=
                    (int64_t)((double)x->info.device_bytes * sa.n_seg_pass / std::max(1, x->img.n_segments));
            }
            }
        }
        if (x->prof) x->prof_acc.scan_node_query_pairs += (int64_t)x->level_cnt[l] * S * nq;
        se.final_pass = final_pass ? 1 : 0;
        {
            Timer t(x, stream, 2);
            DPQ_HIP(dpq::launch_select(se, x->M, nq, stream));
        }
        if (x->prof) x->prof_acc.select_launches++;
    }

    // The only host synchronisation of the batch: did any query drop candidates
    // at some level (buffer overflow)?  Then its list may miss entries.
    if (flag_slot > 0) return DPQ_OK;  // asynchronous batch: dpq_finish checks the word
    DPQ_HIP(hipStreamSynchronize(stream));
    if (*reinterpret_cast<volatile uint32_t*>(x->h_any) == 0) return DPQ_OK;
    std::vector<int> over;
    for (int base = 0; base < nq; base += 4096) {
        const int n = std::min(4096, nq - base);
        DPQ_HIP(hipMemcpyAsync(x->h_overflow, x->d_overflow + base, sizeof(uint32_t) * n, hipMemcpyDeviceToHost,
                               stream));
        DPQ_HIP(hipStreamSynchronize(stream));
        for (int i = 0; i < n; ++i)
            if (x->h_overflow[i]) over.push_back(base + i);
    }
    if (over.empty()) return DPQ_OK;

    // Rerun the affected queries over the whole shard in ONE filter level.  The
    // k-th key of the incomplete list is still a valid upper bound (its entries
    // are real nodes) and it is tight, so the number of nodes under it is about
    // top_k; grow the buffer and repeat in the (pathological) case it is not.
    const int slots2 = ((int)over.size() + QG - 1) / QG * QG;
    const int ng2 = slots2 / QG;
    std::vector<int32_t> slot_query((size_t)slots2, -1);
    std::vector<uint64_t> h_key((size_t)nqp), k2((size_t)slots2, ~0ull);
    DPQ_HIP(hipMemcpy(h_key.data(), x->d_thr_key, sizeof(uint64_t) * nqp, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < over.size(); ++i) {
        slot_query[i] = over[i];
        k2[i] = h_key[(size_t)over[i]];
    }
    // few, large regions: with a tight threshold the candidates of a query may all sit in one workgroup's share
    const int splits2 = std::min(16, splits_for(x->img.n_segments, ng2));
    int64_t rcap2 = std::max<int64_t>(2 * (int64_t)top_k, 1024);
    for (int attempt = 0;; ++attempt) {
        const int64_t stride2 = (int64_t)top_k + (int64_t)splits2 * rcap2;
        int32_t* d_slot_query = nullptr;
        uint32_t *c_count = nullptr, *c_over = nullptr;
        uint64_t *c_keys = nullptr, *c_scratch = nullptr, *c_tk = nullptr;
        uint4* c_qtab = nullptr;
        auto cleanup = [&]() {
            hipFree(d_slot_query); hipFree(c_count); hipFree(c_over); hipFree(c_keys); hipFree(c_scratch); hipFree(c_tk);
            hipFree(c_qtab);
        };
        rc = stride2 > INT32_MAX ? fail(DPQ_ERR_NOMEM, "candidate buffer too large") : DPQ_OK;
        if (!rc) rc = dev_alloc(&d_slot_query, (size_t)slots2);
        if (!rc) rc = dev_alloc(&c_count, (size_t)slots2 * dpq::kRegionStride);
        if (!rc) rc = dev_alloc(&c_over, (size_t)slots2);
        if (!rc) rc = dev_alloc(&c_keys, (size_t)slots2 * stride2);
        if (!rc) rc = dev_alloc(&c_scratch, (size_t)slots2 * stride2);
        if (!rc) rc = dev_alloc(&c_tk, (size_t)slots2);
        if (!rc) rc = dev_alloc(&c_qtab, (size_t)ng2 * (dpq::qtab_bytes_per_group(x->M) / sizeof(uint4)));
        if (rc) {
            cleanup();
            return rc;
        }
        hipError_t e = hipSuccess;
        auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
        chk(hipMemcpy(d_slot_query, slot_query.data(), sizeof(int32_t) * slots2, hipMemcpyHostToDevice));
        chk(hipMemcpy(c_tk, k2.data(), sizeof(uint64_t) * slots2, hipMemcpyHostToDevice));
        chk(hipMemsetAsync(c_count, 0, sizeof(uint32_t) * slots2 * dpq::kRegionStride, stream));  // region 0: no carried winners
        chk(hipMemsetAsync(c_over, 0, sizeof(uint32_t) * slots2, stream));
        sa.seg_list = nullptr;
        sa.n_seg_pass = x->img.n_segments;
        sa.tight_hist = nullptr;  // the rerun starts from final thresholds
        if (!one_tile) {  // a tiled scratch holds the last tile only: the rerun decodes inside the scan, code values as labels
            sa.img.raw = x->img.raw;
            sa.lut32 = x->d_lut32;
        }
        sa.thr_key = c_tk;
        sa.slot_query = d_slot_query;
        sa.n_queries = slots2;
        sa.cand_count = c_count;
        sa.cand_key = c_keys;
        sa.cand_stride = stride2;
        sa.region_cap = (int32_t)rcap2;
        sa.qtab = c_qtab;
        chk(dpq::launch_quantise(sa, ng2, stream));
        chk(dpq::launch_scan(sa, ng2, splits2, stream));
        std::vector<uint32_t> h_cnt((size_t)slots2 * dpq::kRegionStride, 0);
        chk(hipMemcpyAsync(h_cnt.data(), c_count, sizeof(uint32_t) * h_cnt.size(), hipMemcpyDeviceToHost, stream));
        chk(hipStreamSynchronize(stream));
        uint32_t max_cnt = 0;
        for (uint32_t c : h_cnt) max_cnt = std::max(max_cnt, c);
        if (e == hipSuccess && (int64_t)max_cnt > rcap2 && attempt < 4) {
            cleanup();
            rcap2 = (int64_t)max_cnt + 64;
            continue;
        }
        se.shared_id = nullptr;
        se.shared_code = nullptr;
        se.shared_n = 0;
        se.cand_count = c_count;
        se.cand_key = c_keys;
        se.cand_stride = stride2;
        se.region_cap = (int32_t)rcap2;
        se.n_regions = 1 + splits2;
        se.scratch = c_scratch;
        se.slot_query = d_slot_query;
        se.thr_key = c_tk;
        se.overflow = c_over;
        se.any_overflow = nullptr;
        se.final_pass = 1;
        chk(dpq::launch_select(se, x->M, slots2, stream));
        chk(hipStreamSynchronize(stream));
        cleanup();
        if (e != hipSuccess) return fail(DPQ_ERR_HIP, std::string("overflow rerun: ") + hipGetErrorString(e));
        if ((int64_t)max_cnt > rcap2) return fail(DPQ_ERR_NOMEM, "candidate overflow persisted after reruns");
        break;
    }
    if (x->prof) x->prof_acc.overflow_reruns += (int64_t)over.size();
    return DPQ_OK;
}

int open_from_payload(const uint8_t* payload, int64_t n_bytes, int64_t n_codes, int M, int K,
                      const dpq_open_opts* opts, dpq_index** out) {
    if (!out) return fail(DPQ_ERR_ARG, "out is NULL");
    *out = nullptr;
    dpq_open_opts o{};
    if (opts) o = *opts;
    if (M != 8 && M != 16)
        return fail(DPQ_ERR_ARG, "this build has scan kernels for M = 8 (reference format) and M = 16 (own extension)");
    if (K < 1 || K > 256) return fail(DPQ_ERR_ARG, "K must be in 1..256 (one byte per sub-code)");
    if (o.global_offset < 0 || o.global_n_codes < 0 || (o.global_offset != 0 && o.global_n_codes == 0) ||
        (o.global_n_codes > 0 && o.global_offset + n_codes > o.global_n_codes))
        return fail(DPQ_ERR_ARG, "global_offset / global_n_codes do not enclose this payload (a part of a larger index "
                                 "needs global_n_codes > 0)");
    if (n_codes >= (int64_t)INT32_MAX || o.global_n_codes >= (int64_t)INT32_MAX || o.global_offset + n_codes >= (int64_t)INT32_MAX)
        return fail(DPQ_ERR_ARG, "ids beyond 2^31 - 1: results carry int32 DFS positions (h:2979)");
    if (o.chunks_per_segment > dpq::kSortMax / dpq::kChunk)
        return fail(DPQ_ERR_ARG, "chunks_per_segment must be <= 64 (a segment is the cascade's level-0 unit)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(DPQ_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    if (o.device < 0 || o.device >= ndev) return fail(DPQ_ERR_NO_DEVICE, "device ordinal out of range");

    dpq::SoA soa;
    std::string err;
    // threshold bootstrap: auto = on from 64 K nodes per shard (estimated before the byte-balanced cut)
    const int64_t n_scan = o.num_codes > 0 ? std::min<int64_t>(o.num_codes, n_codes) : n_codes;
    const int64_t per_shard = n_scan / std::max(1, o.shard_count);
    int mi_stride = o.bootstrap < 0 ? 0 : dpq::bootstrap_stride_for(per_shard);
    if (o.bootstrap > 0 && mi_stride == 0 && per_shard >= 16384) mi_stride = 1;  // forced on (tests, experiments)
    // the strand image (the stream pass's own layout, ~1.2 x the payload in host RAM and HBM) only where run_batch
    // will read it: big shards (cut by bytes: a margin on the estimate), or forced (tests, experiments)
    const Tuning tune0 = resolve_tuning(o);
    const bool want_strands = tune0.strands && (tune0.force_strands || per_shard >= dpq::kStrandMinNodes / 4 * 3);
    int rc = dpq::transcode(payload, n_bytes, n_codes, M, o.shard_rank, o.shard_count, o.chunks_per_segment, &soa,
                            &err, o.num_codes, mi_stride, 0, want_strands ? 1 : 0);
    if (rc) return fail(rc, err);

    DPQ_HIP(hipSetDevice(o.device));
    dpq_index* x = new dpq_index();
    x->device = o.device;
    x->M = M;
    x->K = K;
    x->cap_auto = o.cand_capacity <= 0;
    x->cap = o.cand_capacity;
    x->batch_decode = o.batch_decode;
    x->tune = tune0;
    auto up = [&](auto** dptr, const void* src, size_t bytes) -> int {
        using T = std::remove_pointer_t<std::remove_pointer_t<decltype(dptr)>>;
        int r = dev_alloc(dptr, (bytes + sizeof(T) - 1) / sizeof(T) + 64 / sizeof(T));
        if (r) return r;
        if (bytes) {
            hipError_t e = hipMemcpy(*dptr, src, bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) return fail(DPQ_ERR_HIP, std::string("upload: ") + hipGetErrorString(e));
        }
        return DPQ_OK;
    };
    rc = up(&x->d_nib, soa.nib.data(), soa.nib.size());
    if (!rc) rc = up(&x->d_par, soa.par.data(), soa.par.size());
    if (!rc) rc = up(&x->d_carry, soa.carry.data(), soa.carry.size());
    if (!rc) rc = up(&x->d_mask, soa.mask.data(), soa.mask.size());
    if (!rc) rc = up(&x->d_delta, soa.delta.data(), soa.delta.size());
    if (!rc) rc = up(&x->d_seg_off, soa.seg_delta_off.data(), soa.seg_delta_off.size() * 8);
    if (!rc) rc = up(&x->d_ckpt, soa.seg_ckpt.data(), soa.seg_ckpt.size());
    if (o.global_offset != 0)
        for (uint32_t& id : soa.mi_id) id += (uint32_t)o.global_offset;
    if (!rc && soa.mi_stride > 0 && (int64_t)soa.mi_id.size() >= 16384) {
        rc = up(&x->d_mi_cell, soa.mi_cell_start.data(), soa.mi_cell_start.size() * 4);
        if (!rc) rc = up(&x->d_mi_code, soa.mi_code.data(), soa.mi_code.size() * 4);
        if (!rc) rc = up(&x->d_mi_id, soa.mi_id.data(), soa.mi_id.size() * 4);
        x->boot = !rc;
        x->boot_classes = soa.mi_classes;
    }
    if (!rc && x->tune.relabel && soa.relabel.size() == (size_t)M * 256) {
        rc = up(&x->d_relabel, soa.relabel.data(), soa.relabel.size());
    }
    if (!rc && x->boot && x->tune.strands && soa.n_strips > 0 && soa.n_strips < INT32_MAX) {
        rc = up(&x->d_st_ckpt, soa.st_ckpt.data(), soa.st_ckpt.size() * 8);
        if (!rc) rc = up(&x->d_st_mask, soa.st_mask.data(), soa.st_mask.size() * 4);
        if (!rc) rc = up(&x->d_st_depth, soa.st_depth.data(), soa.st_depth.size() * 2);
        // (st_poff stays on the host: the kernel computes a lane's offset inside a phase as a wave prefix sum of the lanes'
        // byte counts; the array exists for the CPU-side checks of the image)
        if (!rc) rc = up(&x->d_st_pbase, soa.st_pbase.data(), soa.st_pbase.size() * 4);
        if (!rc) rc = up(&x->d_st_delta, soa.st_delta.data(), soa.st_delta.size());
        if (!rc) {
            // strips are visited in a low-discrepancy order too (every prefix a spread sample of the shard)
            const int64_t ns = soa.n_strips;
            std::vector<uint32_t> order((size_t)ns);
            int64_t P = std::max<int64_t>(1, (int64_t)((double)ns * 0.6180339887498949));
            auto gcd = [](int64_t a, int64_t b) { while (b) { int64_t t = a % b; a = b; b = t; } return a; };
            while (gcd(P, ns) != 1) ++P;
            for (int64_t j = 0; j < ns; ++j) order[(size_t)j] = (uint32_t)((j * P) % ns);
            rc = up(&x->d_strip_order, order.data(), order.size() * 4);
            const int64_t S = soa.nodes_per_segment();
            if (!rc && dpq::kStripNodes % S == 0) {
                std::vector<uint32_t> segs;
                x->strip_seg_off.assign((size_t)ns + 1, 0);
                for (int64_t j = 0; j < ns; ++j) {
                    const int64_t s0 = (int64_t)order[(size_t)j] * (dpq::kStripNodes / S);
                    for (int64_t t = s0; t < std::min(s0 + dpq::kStripNodes / S, soa.n_segments); ++t) segs.push_back((uint32_t)t);
                    x->strip_seg_off[(size_t)j + 1] = (int64_t)segs.size();
                }
                rc = up(&x->d_strip_segs, segs.data(), segs.size() * 4);
            }
        }
        if (!rc) {
            x->img.st_ckpt = x->d_st_ckpt;
            x->img.st_mask = x->d_st_mask;
            x->img.st_depth = x->d_st_depth;
            x->img.st_pbase = x->d_st_pbase;
            x->img.st_delta = x->d_st_delta;
            x->img.n_strips = (int32_t)soa.n_strips;
            x->strand_bytes = soa.strand_bytes();
        }
    }
    if (rc) {
        dpq_close(x);
        return rc;
    }
    x->img.nib = x->d_nib;
    x->img.par = x->d_par;
    x->img.carry = x->d_carry;
    x->img.mask = x->d_mask;
    x->img.delta = x->d_delta;
    x->img.seg_delta_off = x->d_seg_off;
    x->img.seg_ckpt = x->d_ckpt;
    x->img.n_local = soa.node_hi - soa.node_lo;
    x->img.n_codes_total = o.global_n_codes > 0 ? o.global_n_codes : soa.n_codes_total;
    x->img.id_base = (uint32_t)(o.global_offset + soa.node_lo);
    x->img.n_segments = (int32_t)soa.n_segments;
    x->img.chunks_per_segment = soa.chunks_per_segment;
    x->img.M = M;
    x->img.K = K;

    dpq_info& inf = x->info;
    inf.n_codes_total = x->img.n_codes_total;
    inf.n_bytes_total = soa.n_bytes_total;
    inf.node_lo = o.global_offset + soa.node_lo;
    inf.node_hi = o.global_offset + soa.node_hi;
    inf.algorithmic_bytes = soa.algorithmic_bytes;
    inf.device_bytes = soa.device_bytes();
    inf.bootstrap_bytes = x->boot ? soa.bootstrap_bytes() : 0;
    inf.bootstrap_stride = x->boot ? soa.mi_stride : 0;
    inf.batch_decode_mb = batch_decode_possible(x) ? (int32_t)((batch_raw_bytes(x) + (1 << 20) - 1) >> 20) : 0;
    inf.strand_bytes = x->strand_bytes;
    inf.n_diffs = soa.n_diffs;
    inf.M = M;
    inf.K = K;
    inf.Ds = 0;
    inf.n_segments = (int32_t)soa.n_segments;
    inf.chunks_per_segment = soa.chunks_per_segment;
    inf.max_depth = soa.max_depth;
    inf.device = o.device;
    inf.cand_capacity = x->cap;
    *out = x;
    return DPQ_OK;
}

int open_plain(const uint8_t* codes, int64_t n_codes, int M, int K, const dpq_open_opts* opts, dpq_index** out) {
    if (!out) return fail(DPQ_ERR_ARG, "out is NULL");
    *out = nullptr;
    dpq_open_opts o{};
    if (opts) o = *opts;
    if (!codes || n_codes < 1 || n_codes >= (int64_t)INT32_MAX) return fail(DPQ_ERR_ARG, "bad codes / n_codes");
    if (M != 8 && M != 16) return fail(DPQ_ERR_ARG, "M must be 8 or 16");
    if (K < 1 || K > 256) return fail(DPQ_ERR_ARG, "K must be in 1..256 (one byte per sub-code)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(DPQ_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    if (o.device < 0 || o.device >= ndev) return fail(DPQ_ERR_NO_DEVICE, "device ordinal out of range");
    int cps = o.chunks_per_segment <= 0 ? dpq::kDefaultChunksPerSegment : o.chunks_per_segment;
    if (cps > dpq::kSortMax / dpq::kChunk) return fail(DPQ_ERR_ARG, "chunks_per_segment must be <= 64");
    int count = o.shard_count <= 0 ? 1 : o.shard_count;
    if (o.shard_rank < 0 || o.shard_rank >= count) return fail(DPQ_ERR_ARG, "bad shard_rank / shard_count");
    if (o.num_codes < 0 || (int64_t)o.num_codes > n_codes) return fail(DPQ_ERR_ARG, "num_codes outside 0..n_codes");
    if (o.num_codes > 0) n_codes = o.num_codes;  // scan only the first num_codes codes (h:2625-2629)
    const int64_t S = (int64_t)dpq::kChunk * cps;
    const int64_t nseg_total = (n_codes + S - 1) / S;
    const int64_t seg_lo = nseg_total * o.shard_rank / count, seg_hi = nseg_total * (o.shard_rank + 1) / count;
    const int64_t lo = std::min(seg_lo * S, n_codes), hi = std::min(seg_hi * S, n_codes);
    DPQ_HIP(hipSetDevice(o.device));
    dpq_index* x = new dpq_index();
    x->device = o.device;
    x->M = M;
    x->K = K;
    x->plain = true;
    x->tune = resolve_tuning(o);
    x->cap_auto = o.cand_capacity <= 0;
    x->cap = o.cand_capacity;
    const size_t padded = (size_t)(seg_hi - seg_lo) * S * M;
    int rc = dev_alloc(&x->d_raw, padded + 64);
    if (!rc) {
        hipError_t e = hipMemset(x->d_raw, 0, padded + 64);
        if (e == hipSuccess && hi > lo)
            e = hipMemcpy(x->d_raw, codes + (size_t)lo * M, (size_t)(hi - lo) * M, hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail(DPQ_ERR_HIP, std::string("upload: ") + hipGetErrorString(e));
    }
    if (rc) {
        dpq_close(x);
        return rc;
    }
    {
        int mi_stride = o.bootstrap < 0 ? 0 : dpq::bootstrap_stride_for(hi - lo);
        if (o.bootstrap > 0 && mi_stride == 0 && hi - lo >= 16384) mi_stride = 1;
        if (mi_stride > 0) {
            dpq::SoA mi;
            std::vector<uint32_t> ids;
            std::vector<uint8_t> cds;
            for (int64_t i = lo; i < hi; i += mi_stride) {
                ids.push_back((uint32_t)i);
                cds.insert(cds.end(), codes + (size_t)i * M, codes + (size_t)(i + 1) * M);
            }
            dpq::build_multi_index(ids, cds, M, mi_stride, dpq::bootstrap_classes_for((int64_t)ids.size()), &mi);
            auto upl = [&](uint32_t** d, const std::vector<uint32_t>& v) -> int {
                int r = dev_alloc(d, v.size() + 16);
                if (r) return r;
                if (!v.empty() && hipMemcpy(*d, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
                    return fail(DPQ_ERR_HIP, "upload of the bootstrap multi-index failed");
                return DPQ_OK;
            };
            rc = upl(&x->d_mi_cell, mi.mi_cell_start);
            if (!rc) rc = upl(&x->d_mi_code, mi.mi_code);
            if (!rc) rc = upl(&x->d_mi_id, mi.mi_id);
            if (rc) {
                dpq_close(x);
                return rc;
            }
            x->boot = true;
            x->boot_classes = mi.mi_classes;
            x->info.bootstrap_bytes = mi.bootstrap_bytes();
            x->info.bootstrap_stride = mi_stride;
        }
    }
    x->img.raw = x->d_raw;
    x->img.n_local = hi - lo;
    x->img.n_codes_total = n_codes;
    x->img.id_base = (uint32_t)lo;
    x->img.n_segments = (int32_t)(seg_hi - seg_lo);
    x->img.chunks_per_segment = cps;
    x->img.M = M;
    x->img.K = K;
    dpq_info& inf = x->info;
    inf.n_codes_total = n_codes;
    inf.n_bytes_total = n_codes * M;
    inf.node_lo = lo;
    inf.node_hi = hi;
    inf.algorithmic_bytes = (hi - lo) * M;
    inf.device_bytes = (int64_t)padded;
    inf.M = M;
    inf.K = K;
    inf.n_segments = x->img.n_segments;
    inf.chunks_per_segment = cps;
    inf.device = o.device;
    inf.cand_capacity = x->cap;
    *out = x;
    return DPQ_OK;
}

void fill_info_from_soa(const dpq::SoA& s, dpq_info* inf) {
    memset(inf, 0, sizeof *inf);
    inf->n_codes_total = s.n_codes_total;
    inf->n_bytes_total = s.n_bytes_total;
    inf->node_lo = s.node_lo;
    inf->node_hi = s.node_hi;
    inf->algorithmic_bytes = s.algorithmic_bytes;
    inf->device_bytes = s.device_bytes();
    inf->bootstrap_bytes = s.bootstrap_bytes();
    inf->bootstrap_stride = s.mi_stride;
    inf->strand_bytes = s.n_strips > 0 ? s.strand_bytes() : 0;
    inf->n_diffs = s.n_diffs;
    inf->M = s.M;
    inf->n_segments = (int32_t)s.n_segments;
    inf->chunks_per_segment = s.chunks_per_segment;
    inf->max_depth = s.max_depth;
    inf->device = -1;
}

}  // namespace

extern "C" {

int dpq_version(void) { return DPQ_VERSION; }

const char* dpq_strerror(int status) {
    switch (status) {
        case DPQ_OK: return "ok";
        case DPQ_ERR_ARG: return "bad argument";
        case DPQ_ERR_IO: return "i/o error";
        case DPQ_ERR_FORMAT: return "malformed DTC stream";
        case DPQ_ERR_NO_DEVICE: return "no usable GPU";
        case DPQ_ERR_HIP: return "HIP runtime error";
        case DPQ_ERR_NOMEM: return "out of memory";
        case DPQ_ERR_STATE: return "call out of order";
        case DPQ_ERR_TOPK: return "top_k exceeds the number of codes";
        default: return "unknown status";
    }
}

const char* dpq_last_error(void) { return g_last_error.c_str(); }

int dpq_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n < 0 ? 0 : n;
}

int dpq_read_dtc_header(const char* path, int64_t* n_codes, int64_t* n_bytes) {
    return guarded([&]() -> int {
    if (!path || !n_codes || !n_bytes) return fail(DPQ_ERR_ARG, "NULL argument");
    std::string err;
    int rc = dpq::read_dtc_header(path, n_codes, n_bytes, &err);
    return rc ? fail(rc, err) : DPQ_OK;
    });
}

int dpq_read_codewords(const char* path, int32_t* M, int32_t* K, int32_t* Ds, float* out) {
    return guarded([&]() -> int {
    if (!path || !M || !K || !Ds) return fail(DPQ_ERR_ARG, "NULL argument");
    std::string err;
    std::vector<float> v;
    int m, k, ds;
    int rc = dpq::read_codewords(path, &m, &k, &ds, out ? &v : nullptr, &err);
    if (rc) return fail(rc, err);
    *M = m;
    *K = k;
    *Ds = ds;
    if (out) memcpy(out, v.data(), v.size() * sizeof(float));
    return DPQ_OK;
    });
}

int dpq_read_vecs(const char* path, int is_bvecs, int64_t* n, int32_t* D, float* out, int64_t cap) {
    return guarded([&]() -> int {
    if (!path || !n || !D) return fail(DPQ_ERR_ARG, "NULL argument");
    std::string err;
    std::vector<float> v;
    int d = 0;
    int rc = dpq::read_vecs(path, is_bvecs != 0, n, &d, out ? &v : nullptr, cap, &err);
    if (rc) return fail(rc, err);
    *D = d;
    if (out) memcpy(out, v.data(), v.size() * sizeof(float));
    return DPQ_OK;
    });
}

int dpq_dtc_file_name(const char* dataset_dir, int M, int K, int64_t N, char* out, int64_t out_len) {
    return guarded([&]() -> int {
    if (!dataset_dir || !out) return fail(DPQ_ERR_ARG, "NULL argument");
    std::string s = dpq::dtc_file_name(dataset_dir, M, K, N);
    if ((int64_t)s.size() + 1 > out_len) return fail(DPQ_ERR_ARG, "output buffer too small");
    memcpy(out, s.c_str(), s.size() + 1);
    return DPQ_OK;
    });
}

int dpq_dtc_validate(const uint8_t* payload, int64_t n_bytes, int64_t n_codes, int M, dpq_dtc_stats* stats) {
    return guarded([&]() -> int {
    std::string err;
    int rc = dpq::validate(payload, n_bytes, n_codes, M, stats, &err);
    return rc ? fail(rc, err) : DPQ_OK;
    });
}

int dpq_soa_build(const uint8_t* payload, int64_t n_bytes, int64_t n_codes, int M, const dpq_open_opts* opts,
                  dpq_soa** out) {
    return guarded([&]() -> int {
    if (!out) return fail(DPQ_ERR_ARG, "out is NULL");
    dpq_open_opts o{};
    if (opts) o = *opts;
    dpq_soa* s = new dpq_soa();
    std::string err;
    int rc = dpq::transcode(payload, n_bytes, n_codes, M, o.shard_rank, o.shard_count, o.chunks_per_segment, &s->soa,
                            &err, o.num_codes, o.bootstrap > 0 && M % 4 == 0 ? o.bootstrap : 0);
    if (rc) {
        delete s;
        *out = nullptr;
        return fail(rc, err);
    }
    *out = s;
    return DPQ_OK;
    });
}

int dpq_soa_info(const dpq_soa* soa, dpq_info* info) {
    return guarded([&]() -> int {
    if (!soa || !info) return fail(DPQ_ERR_ARG, "NULL argument");
    fill_info_from_soa(soa->soa, info);
    return DPQ_OK;
    });
}

int dpq_soa_array(const dpq_soa* soa, int which, const void** ptr, int64_t* n_bytes) {
    return guarded([&]() -> int {
    if (!soa || !ptr || !n_bytes) return fail(DPQ_ERR_ARG, "NULL argument");
    const dpq::SoA& s = soa->soa;
    switch (which) {
        case 0: *ptr = s.nib.data(); *n_bytes = (int64_t)s.nib.size(); break;
        case 1: *ptr = s.mask.data(); *n_bytes = (int64_t)s.mask.size(); break;
        case 2: *ptr = s.delta.data(); *n_bytes = (int64_t)s.delta.size(); break;
        case 3: *ptr = s.seg_delta_off.data(); *n_bytes = (int64_t)s.seg_delta_off.size() * 8; break;
        case 4: *ptr = s.seg_ckpt.data(); *n_bytes = (int64_t)s.seg_ckpt.size(); break;
        case 5: *ptr = s.mi_cell_start.data(); *n_bytes = (int64_t)s.mi_cell_start.size() * 4; break;
        case 6: *ptr = s.mi_code.data(); *n_bytes = (int64_t)s.mi_code.size() * 4; break;
        case 7: *ptr = s.mi_id.data(); *n_bytes = (int64_t)s.mi_id.size() * 4; break;
        case 8: *ptr = s.par.data(); *n_bytes = (int64_t)s.par.size(); break;
        case 9: *ptr = s.carry.data(); *n_bytes = (int64_t)s.carry.size(); break;
        case 10: *ptr = s.st_ckpt.data(); *n_bytes = (int64_t)s.st_ckpt.size() * 8; break;
        case 11: *ptr = s.st_mask.data(); *n_bytes = (int64_t)s.st_mask.size() * 4; break;
        case 12: *ptr = s.st_poff.data(); *n_bytes = (int64_t)s.st_poff.size() * 2; break;
        case 13: *ptr = s.st_pbase.data(); *n_bytes = (int64_t)s.st_pbase.size() * 4; break;
        case 14: *ptr = s.st_delta.data(); *n_bytes = (int64_t)s.st_delta.size(); break;
        case 15: *ptr = s.st_depth.data(); *n_bytes = (int64_t)s.st_depth.size() * 2; break;
        default: return fail(DPQ_ERR_ARG, "which must be 0..15");
    }
    return DPQ_OK;
    });
}

void dpq_soa_free(dpq_soa* soa) { delete soa; }

int dpq_dtc_encode(const uint8_t* root_code, const uint8_t* depths, const uint16_t* masks, const uint8_t* deltas,
                   int64_t n_codes, int M, uint8_t* out, int64_t* n_bytes) {
    return guarded([&]() -> int {
    std::string err;
    int rc = dpq::encode(root_code, depths, masks, deltas, n_codes, M, out, n_bytes, &err);
    return rc ? fail(rc, err) : DPQ_OK;
    });
}

int dpq_tree_build(const uint8_t* codes, int64_t n_codes, int M, int K, int max_height_folds, const float* codewords,
                   int Ds, dpq_tree** out) {
    return guarded([&]() -> int {
    if (!out) return fail(DPQ_ERR_ARG, "out is NULL");
    *out = nullptr;
    dpq_tree* t = new dpq_tree();
    std::string err;
    int rc = dpq::build_tree(codes, n_codes, M, K, max_height_folds, codewords, Ds, &t->tree, &err);
    if (rc) {
        delete t;
        return fail(rc, err);
    }
    *out = t;
    return DPQ_OK;
    });
}

int dpq_tree_build_gpu(const uint8_t* codes, int64_t n_codes, int M, int K, int max_height_folds,
                       const float* codewords, int Ds, int device, dpq_tree** out) {
    return guarded([&]() -> int {
    if (!out) return fail(DPQ_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!codes || n_codes < 1 || n_codes >= (int64_t)INT32_MAX || M < 1 || M > 16 || K < 1 || K > 256 ||
        max_height_folds < 1)
        return fail(DPQ_ERR_ARG, "bad argument to dpq_tree_build_gpu");
    std::string err;
    std::vector<uint32_t> finalists;
    std::vector<std::pair<uint32_t, uint32_t>> edges;
    // DPQ_DEV=1 DPQ_BUILD_PREFILTER=0: every position subset sorted whole, as rounds 2 - 3 did (developer A/B; same tree)
    const bool prefilter = !(dev_mode() && getenv("DPQ_BUILD_PREFILTER") && atoi(getenv("DPQ_BUILD_PREFILTER")) == 0);
    int rc = dpq::find_edges_gpu(codes, n_codes, M, max_height_folds, device, &finalists, &edges, &err, prefilter);
    if (rc) return fail(rc, err);
    dpq_tree* t = new dpq_tree();
    // DPQ_DEV=1 DPQ_BUILD_LAYOUT=host keeps the layout on the host (developer A/B; same tree either way)
    static const bool host_layout = dev_mode() && getenv("DPQ_BUILD_LAYOUT") &&
                                    std::string(getenv("DPQ_BUILD_LAYOUT")) == "host";
    rc = host_layout ? dpq::layout_tree(codes, n_codes, M, K, max_height_folds, codewords, Ds, finalists, &edges, &t->tree, &err)
                     : dpq::layout_tree_gpu(codes, n_codes, M, K, max_height_folds, codewords, Ds, finalists, &edges, device,
                                            &t->tree, &err);
    if (rc) {
        delete t;
        return fail(rc, err);
    }
    *out = t;
    return DPQ_OK;
    });
}

int dpq_tree_stats(const dpq_tree* t, dpq_dtc_stats* stats) {
    return guarded([&]() -> int {
    if (!t || !stats) return fail(DPQ_ERR_ARG, "NULL argument");
    const dpq::Tree& tr = t->tree;
    memset(stats, 0, sizeof *stats);
    stats->n_codes = tr.n;
    stats->n_diffs = tr.n_diffs;
    stats->n_bytes = tr.M + tr.n_diffs + (tr.n - 1) * dpq::mask_bytes_for(tr.M) + tr.n / 2;  // h:1765 for M = 8
    for (int d = 0; d < 16; ++d) stats->depth_hist[d] = tr.depth_hist[d];
    stats->max_depth = tr.max_depth;
    stats->M = tr.M;
    return DPQ_OK;
    });
}

int dpq_tree_array(const dpq_tree* t, int which, const void** ptr, int64_t* n_bytes) {
    return guarded([&]() -> int {
    if (!t || !ptr || !n_bytes) return fail(DPQ_ERR_ARG, "NULL argument");
    const dpq::Tree& tr = t->tree;
    switch (which) {
        case 0: *ptr = tr.vec_id.data(); *n_bytes = (int64_t)tr.vec_id.size() * 4; break;
        case 1: *ptr = tr.parent_pos.data(); *n_bytes = (int64_t)tr.parent_pos.size() * 4; break;
        case 2: *ptr = tr.depth.data(); *n_bytes = (int64_t)tr.depth.size(); break;
        case 3: *ptr = tr.mask.data(); *n_bytes = (int64_t)tr.mask.size() * 2; break;
        case 4: *ptr = tr.deltas.data(); *n_bytes = (int64_t)tr.deltas.size(); break;
        case 5: *ptr = tr.root_code.data(); *n_bytes = (int64_t)tr.root_code.size(); break;
        case 6: *ptr = tr.edges.data(); *n_bytes = (int64_t)tr.edges.size() * 8; break;
        default: return fail(DPQ_ERR_ARG, "which must be 0..6");
    }
    return DPQ_OK;
    });
}

int dpq_tree_encode(const dpq_tree* t, uint8_t* out, int64_t* n_bytes) {
    return guarded([&]() -> int {
    if (!t || !n_bytes) return fail(DPQ_ERR_ARG, "NULL argument");
    const dpq::Tree& tr = t->tree;
    std::string err;
    int rc = dpq::encode(tr.root_code.data(), tr.depth.data(), tr.mask.data(), tr.deltas.data(), tr.n, tr.M, out,
                         n_bytes, &err);
    return rc ? fail(rc, err) : DPQ_OK;
    });
}

int dpq_tree_write_files(const dpq_tree* t, const char* dataset_dir) {
    return guarded([&]() -> int {
    if (!t || !dataset_dir) return fail(DPQ_ERR_ARG, "NULL argument");
    std::string err;
    int rc = dpq::tree_write_files(t->tree, dataset_dir, &err);
    return rc ? fail(rc, err) : DPQ_OK;
    });
}

void dpq_tree_free(dpq_tree* t) { delete t; }

int dpq_read_qnode_ids(const char* path, int64_t n_codes, uint32_t* vec_ids) {
    return guarded([&]() -> int {
    if (!path || !vec_ids || n_codes < 0) return fail(DPQ_ERR_ARG, "bad argument");
    std::vector<uint32_t> ids;
    std::string err;
    int rc = dpq::read_qnode_ids(path, n_codes, &ids, &err);
    if (rc) return fail(rc, err);
    memcpy(vec_ids, ids.data(), ids.size() * 4);
    return DPQ_OK;
    });
}

int dpq_read_codes_plain(const char* path, int M, int64_t* n_codes, uint8_t* out) {
    return guarded([&]() -> int {
    if (!path || !n_codes || M < 1) return fail(DPQ_ERR_ARG, "bad argument");
    std::vector<uint8_t> codes;
    std::string err;
    int rc = dpq::read_codes_plain(path, M, n_codes, out ? &codes : nullptr, &err);
    if (rc) return fail(rc, err);
    if (out) memcpy(out, codes.data(), codes.size());
    return DPQ_OK;
    });
}

int dpq_read_codes_plain_ex(const char* path, int M, int K, int with_id, int64_t* n_codes, uint8_t* codes_out,
                            int32_t* ids_out) {
    return guarded([&]() -> int {
    if (!path || !n_codes || M < 1 || K < 1) return fail(DPQ_ERR_ARG, "bad argument");
    if (K > 256 && with_id) return fail(DPQ_ERR_ARG, "K > 256 with ids is not implemented in the reference either (pq_tree.cpp:1051-1054)");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(DPQ_ERR_IO, std::string("cannot open ") + path);
    int64_t n = 0;
    if (fread(&n, sizeof(int64_t), 1, f) != 1 || n < 0 || n > (int64_t)INT32_MAX) {
        fclose(f);
        return fail(DPQ_ERR_FORMAT, std::string("bad header in ") + path);
    }
    *n_codes = n;
    int rc = DPQ_OK;
    if (codes_out || ids_out) {
        const size_t cb = (size_t)M * (K > 256 ? 2 : 1), rec = cb + (with_id ? 4 : 0);
        std::vector<uint8_t> buf((size_t)std::min<int64_t>(n, 1 << 16) * rec);
        for (int64_t base = 0; base < n && !rc; base += 1 << 16) {
            const size_t m = (size_t)std::min<int64_t>(1 << 16, n - base);
            if (fread(buf.data(), rec, m, f) != m) rc = fail(DPQ_ERR_IO, std::string("short read on ") + path);
            for (size_t i = 0; i < m && !rc; ++i) {
                if (codes_out) memcpy(codes_out + ((size_t)base + i) * cb, buf.data() + i * rec, cb);
                if (ids_out && with_id) memcpy(ids_out + (size_t)base + i, buf.data() + i * rec + cb, 4);
            }
        }
    }
    fclose(f);
    return rc;
    });
}

int dpq_write_codes_plain(const char* path, const uint8_t* codes, int64_t n_codes, int M) {
    return guarded([&]() -> int {
    if (!path || (!codes && n_codes > 0) || n_codes < 0 || M < 1) return fail(DPQ_ERR_ARG, "bad argument");
    std::string err;
    int rc = dpq::write_codes_plain(path, codes, n_codes, M, &err);
    return rc ? fail(rc, err) : DPQ_OK;
    });
}

int dpq_encode_pq(const float* vectors, int64_t n, int D, const float* codewords, int M, int K, int Ds, int device,
                  uint8_t* codes_out) {
    return guarded([&]() -> int {
    if (!vectors || !codewords || !codes_out || n < 0 || D < 1 || M < 1 || K < 1 || K > 256 || Ds < 1)
        return fail(DPQ_ERR_ARG, "bad argument to dpq_encode_pq");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(DPQ_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(DPQ_ERR_NO_DEVICE, "device ordinal out of range");
    if (n == 0) return DPQ_OK;
    DPQ_HIP(hipSetDevice(device));
    float *d_v = nullptr, *d_c = nullptr;
    uint8_t* d_o = nullptr;
    const int64_t tile = 1 << 20;  // vectors per upload
    int rc = dev_alloc(&d_v, (size_t)std::min(n, tile) * D);
    if (!rc) rc = dev_alloc(&d_c, (size_t)M * K * Ds);
    if (!rc) rc = dev_alloc(&d_o, (size_t)std::min(n, tile) * M);
    hipError_t e = hipSuccess;
    if (!rc) {
        e = hipMemcpy(d_c, codewords, (size_t)M * K * Ds * sizeof(float), hipMemcpyHostToDevice);
        for (int64_t base = 0; base < n && e == hipSuccess; base += tile) {
            const int64_t m = std::min(tile, n - base);
            e = hipMemcpy(d_v, vectors + (size_t)base * D, (size_t)m * D * sizeof(float), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = dpq::launch_encode_pq(d_v, m, D, d_c, M, K, Ds, d_o, nullptr);
            if (e == hipSuccess) e = hipMemcpy(codes_out + (size_t)base * M, d_o, (size_t)m * M, hipMemcpyDeviceToHost);
        }
    }
    hipFree(d_v);
    hipFree(d_c);
    hipFree(d_o);
    if (rc) return rc;
    if (e != hipSuccess) return fail(DPQ_ERR_HIP, std::string("dpq_encode_pq: ") + hipGetErrorString(e));
    return DPQ_OK;
    });
}

int dpq_open_file(const char* path, int M, int K, const dpq_open_opts* opts, dpq_index** out) {
    return guarded([&]() -> int {
    if (!path || !out) return fail(DPQ_ERR_ARG, "NULL argument");
    std::vector<uint8_t> buf;
    std::string err;
    int rc = dpq::read_file(path, &buf, &err);
    if (rc) return fail(rc, err);
    if (buf.size() < 16) return fail(DPQ_ERR_FORMAT, std::string("file shorter than its header: ") + path);
    int64_t h[2];
    memcpy(h, buf.data(), 16);  // h:2823-2824
    if (h[1] < 0 || (uint64_t)h[1] > buf.size() - 16)
        return fail(DPQ_ERR_FORMAT, "n_bytes in the header exceeds the file size");
    return open_from_payload(buf.data() + 16, h[1], h[0], M, K, opts, out);
    });
}

int dpq_open_memory(const uint8_t* payload, int64_t n_bytes, int64_t n_codes, int M, int K,
                    const dpq_open_opts* opts, dpq_index** out) {
    return guarded([&]() -> int {
    return open_from_payload(payload, n_bytes, n_codes, M, K, opts, out);
    });
}

int dpq_open_plain_memory(const uint8_t* codes, int64_t n_codes, int M, int K, const dpq_open_opts* opts,
                          dpq_index** out) {
    return guarded([&]() -> int {
    return open_plain(codes, n_codes, M, K, opts, out);
    });
}

int dpq_open_plain_file(const char* path, int M, int K, const dpq_open_opts* opts, dpq_index** out) {
    return guarded([&]() -> int {
    if (!path || !out) return fail(DPQ_ERR_ARG, "NULL argument");
    std::vector<uint8_t> codes;
    int64_t n = 0;
    std::string err;
    int rc = dpq::read_codes_plain(path, M, &n, &codes, &err);
    if (rc) return fail(rc, err);
    return open_plain(codes.data(), n, M, K, opts, out);
    });
}

int dpq_set_codebook(dpq_index* x, const float* codewords, int Ds) {
    return guarded([&]() -> int {
    if (!x || !codewords || Ds < 1 || Ds > 4096) return fail(DPQ_ERR_ARG, "bad codebook argument");
    if (int rc = dpq_finish(x)) return rc;  // batches in flight still read the old codebook
    DPQ_HIP(hipSetDevice(x->device));
    hipFree(x->d_codebook);
    x->d_codebook = nullptr;
    const size_t n = (size_t)x->M * x->K * Ds;
    int rc = dev_alloc(&x->d_codebook, n);
    if (rc) return rc;
    DPQ_HIP(hipMemcpy(x->d_codebook, codewords, n * sizeof(float), hipMemcpyHostToDevice));
    x->Ds = Ds;
    x->info.Ds = Ds;
    if (x->boot) {
        // neighbour lists of the bootstrap's 8 sort slots (slot s: sub-space 2 (s / 2) (M / 8) + (s & 1)): for every
        // centroid all centroids of its sub-space, nearest first (ties by index; centroids beyond K last)
        std::vector<uint8_t> nbr((size_t)8 * 256 * 256);
        std::vector<std::pair<double, int>> row(256);
        for (int sl = 0; sl < 8; ++sl) {
            const int sub = 2 * (sl >> 1) * (x->M / 8) + (sl & 1);
            const float* cw = codewords + (size_t)sub * x->K * Ds;
            for (int c = 0; c < 256; ++c) {
                for (int o = 0; o < 256; ++o) {
                    double d2 = 1e300;  // beyond K: last
                    if (c < x->K && o < x->K) {
                        d2 = 0;
                        for (int d = 0; d < Ds; ++d) {
                            const double df = (double)cw[(size_t)c * Ds + d] - (double)cw[(size_t)o * Ds + d];
                            d2 += df * df;
                        }
                    }
                    row[(size_t)o] = {d2, o};
                }
                std::sort(row.begin(), row.end());
                for (int o = 0; o < 256; ++o) nbr[((size_t)sl * 256 + c) * 256 + o] = (uint8_t)row[(size_t)o].second;
            }
        }
        hipFree(x->d_nbr);
        x->d_nbr = nullptr;
        rc = dev_alloc(&x->d_nbr, nbr.size());
        if (rc) return rc;
        DPQ_HIP(hipMemcpy(x->d_nbr, nbr.data(), nbr.size(), hipMemcpyHostToDevice));
    }
    return DPQ_OK;
    });
}

int dpq_get_info(const dpq_index* x, dpq_info* info) {
    return guarded([&]() -> int {
    if (!x || !info) return fail(DPQ_ERR_ARG, "NULL argument");
    *info = x->info;
    info->cand_capacity = x->cap_auto ? 0 : x->cap;
    return DPQ_OK;
    });
}

int dpq_close(dpq_index* x) {
    return guarded([&]() -> int {
    if (!x) return DPQ_OK;
    hipSetDevice(x->device);
    if (!x->pending.empty()) {  // batches still in flight: let them drain before their buffers go
        hipDeviceSynchronize();
        x->pending.clear();
    }
    for (auto& ep : x->events) {
        hipEventDestroy(ep.a);
        hipEventDestroy(ep.b);
    }
    for (auto e : x->ev_pool) hipEventDestroy(e);
    free_workspace(x);
    free_parked_lane(x);
    for (int l = 0; l < 2; ++l) {
        if (x->lane_stream[l]) hipStreamDestroy(x->lane_stream[l]);
        if (x->lane_ready[l]) hipEventDestroy(x->lane_ready[l]);
        if (x->lane_select_done[l]) hipEventDestroy(x->lane_select_done[l]);
    }
    hipFree(x->d_st_ckpt);
    hipFree(x->d_st_mask);
    hipFree(x->d_st_depth);
    hipFree(x->d_st_pbase);
    hipFree(x->d_st_delta);
    hipFree(x->d_s1_stamps);
    hipFree(x->d_strip_order);
    hipFree(x->d_strip_segs);
    hipFree(x->d_nib);
    hipFree(x->d_par);
    hipFree(x->d_carry);
    hipFree(x->d_mask);
    hipFree(x->d_delta);
    hipFree(x->d_ckpt);
    hipFree(x->d_raw);
    hipFree(x->d_seg_off);
    hipFree(x->d_boot_stamps);
    hipFree(x->d_nbr);
    hipFree(x->d_batch_raw);  // the active lane's plain-code scratch (the parked lane's goes with free_parked_lane)
    hipFree(x->d_relabel);
    hipFree(x->d_mi_cell);
    hipFree(x->d_mi_code);
    hipFree(x->d_mi_id);
    hipFree(x->d_codebook);
    hipFree(x->d_order);
    hipFree(x->d_l0_id);
    hipFree(x->d_l0_code);
    hipFree(x->d_q_stage);
    hipFree(x->d_ids_stage);
    hipFree(x->d_dists_stage);
    for (auto& hs : x->host_slots) {
        hipFree(hs.d_q);
        hipFree(hs.d_ids);
        hipFree(hs.d_d);
        if (hs.kernels_done) hipEventDestroy(hs.kernels_done);
    }
    if (x->copy_in) hipStreamDestroy(x->copy_in);
    if (x->copy_out) hipStreamDestroy(x->copy_out);
    if (x->h_overflow) hipHostFree(x->h_overflow);
    if (x->h_any) hipHostFree(x->h_any);
    hipFree(x->d_counters);
    delete x;
    return DPQ_OK;
    });
}

namespace {

int check_batch_args(dpq_index* x, const float* d_queries, int nq, int top_k, int32_t* d_ids, float* d_dists) {
    if (!x || !d_queries || !d_ids || !d_dists || nq < 0) return fail(DPQ_ERR_ARG, "NULL argument or nq < 0");
    if (!x->d_codebook) return fail(DPQ_ERR_STATE, "dpq_set_codebook has not been called");
    if (top_k < 1 || top_k > dpq::kMaxTopK) return fail(DPQ_ERR_ARG, "top_k must be in 1..2048");
    if ((int64_t)top_k > x->img.n_codes_total)
        return fail(DPQ_ERR_TOPK, "top_k exceeds the number of codes in the index");
    return DPQ_OK;
}

}  // namespace

int dpq_finish(dpq_index* x) {
    return guarded([&]() -> int {
    if (!x) return fail(DPQ_ERR_ARG, "NULL index");
    if (x->pending.empty()) return DPQ_OK;
    DPQ_HIP(hipSetDevice(x->device));
    std::vector<dpq_index::Pending> todo;
    todo.swap(x->pending);
    {
        std::vector<hipStream_t> seen;
        hipError_t e = hipSuccess;
        for (const auto& p : todo)
            if (std::find(seen.begin(), seen.end(), p.stream) == seen.end()) {
                const hipError_t r = hipStreamSynchronize(p.stream);
                if (e == hipSuccess) e = r;
                seen.push_back(p.stream);
            }
        if (e != hipSuccess) {  // the device is in an error state: nothing can be rerun; the batches stay pending
            x->pending.insert(x->pending.begin(), todo.begin(), todo.end());
            return fail(DPQ_ERR_HIP, std::string("dpq_finish: ") + hipGetErrorString(e));
        }
    }
    // Every stream a pending batch ran on is idle from here on (synchronised above; nothing is enqueued while this
    // function runs: a handle belongs to one thread at a time), so a rerun may use whichever lane's workspace is
    // active.  Every batch is settled even if one fails: its callers' buffers must not keep an incomplete list behind
    // a later dpq_finish that has nothing left to report.  The first error is returned.
    int first_rc = DPQ_OK;
    std::string first_msg;
    for (const auto& p : todo) {
        if (*reinterpret_cast<volatile uint32_t*>(x->h_any + p.flag_slot) == 0) continue;
        // a query of this batch dropped candidates: answer the batch again, synchronously (it reruns what overflows)
        x->finish_reruns++;
        int rc = run_batch(x, p.d_queries, p.nq, p.top_k, p.d_ids, p.d_dists, p.stream);
        if (p.host_slot >= 0) x->host_slots[p.host_slot].redo = true;  // its results went down before this
        if (rc && !first_rc) {
            first_rc = rc;
            first_msg = g_last_error;
        }
    }
    // host-to-host batches: their results are on the way down (copy_out), or go down again after a rerun
    bool any_host = false;
    for (auto& hs : x->host_slots) any_host = any_host || hs.busy;
    if (any_host) {
        hipError_t e = x->copy_out ? hipStreamSynchronize(x->copy_out) : hipSuccess;
        for (auto& hs : x->host_slots) {
            if (hs.busy && hs.redo && !hs.direct && e == hipSuccess) {
                e = hipMemcpy(hs.h_ids, hs.d_ids, hs.n_out * sizeof(int32_t), hipMemcpyDeviceToHost);
                if (e == hipSuccess) e = hipMemcpy(hs.h_d, hs.d_d, hs.n_out * sizeof(float), hipMemcpyDeviceToHost);
            }
            hs.busy = hs.redo = false;
        }
        if (e != hipSuccess && !first_rc) {
            first_rc = DPQ_ERR_HIP;
            first_msg = std::string("dpq_finish (results to the host): ") + hipGetErrorString(e);
        }
    }
    if (first_rc) return fail(first_rc, first_msg);
    return DPQ_OK;
    });
}

namespace {
int enqueue_async(dpq_index* x, const float* d_queries, int nq, int top_k, int32_t* d_ids, float* d_dists, void* hip_stream,
                  bool allow_lanes, int host_slot = -1);
}

int dpq_query_batch_device_async(dpq_index* x, const float* d_queries, int nq, int top_k, int32_t* d_ids,
                                 float* d_dists, void* hip_stream) {
    return guarded([&]() -> int { return enqueue_async(x, d_queries, nq, top_k, d_ids, d_dists, hip_stream, true); });
}

int dpq_query_batch_device_ordered(dpq_index* x, const float* d_queries, int nq, int top_k, int32_t* d_ids,
                                   float* d_dists, void* hip_stream) {
    return guarded([&]() -> int { return enqueue_async(x, d_queries, nq, top_k, d_ids, d_dists, hip_stream, false); });
}

int dpq_finish_count(dpq_index* x, int32_t* rerun_batches) {
    return guarded([&]() -> int {
    if (!x) return fail(DPQ_ERR_ARG, "NULL index");
    const int64_t before = x->finish_reruns;
    int rc = dpq_finish(x);
    if (rerun_batches) *rerun_batches = (int32_t)(x->finish_reruns - before);
    return rc;
    });
}

namespace {
int enqueue_async(dpq_index* x, const float* d_queries, int nq, int top_k, int32_t* d_ids, float* d_dists, void* hip_stream,
                  bool allow_lanes, int host_slot) {
    {
    int rc = check_batch_args(x, d_queries, nq, top_k, d_ids, d_dists);
    if (rc || nq == 0) return rc;
    DPQ_HIP(hipSetDevice(x->device));
    hipStream_t user = reinterpret_cast<hipStream_t>(hip_stream);
    // DPQ_OPT_NO_ASYNC_OVERLAP: every batch on the caller's stream with one workspace (round 1's behaviour)
    const bool overlap = x->tune.async_overlap && allow_lanes;
    auto in_flight_on_other_stream = [&]() {
        for (const auto& p : x->pending)
            if (p.user_stream != user) return true;
        return false;
    };
    if (overlap) {
        // Laned batches are ordered by the caller's stream: a batch for another stream first settles what is in flight.
        if (in_flight_on_other_stream() && (rc = dpq_finish(x))) return rc;
    } else {
        // A stream-ordered batch runs on the caller's stream.  Batches still running on a lane's own stream are
        // settled first.  Up to TWO caller streams may have stream-ordered batches in flight: each is given one of the
        // two workspaces (a stream's batches follow each other, so its workspace is never shared) -- a caller that
        // alternates its steps between two streams gets the overlap of two lanes with every step still consumable in
        // stream order (the sharded driver: select -> pack -> all-gather -> merge behind each batch).  A third stream
        // settles everything first.
        bool laned = false;
        for (const auto& p : x->pending) laned = laned || p.stream != p.user_stream;
        if (laned && (rc = dpq_finish(x))) return rc;
        int lane = -1;
        for (int k = 0; k < 2; ++k)
            if (x->ordered_stream_set[k] && x->ordered_stream[k] == user) lane = k;
        if (lane < 0) {
            for (int k = 0; k < 2 && lane < 0; ++k) {
                bool busy = false;
                for (const auto& p : x->pending) busy = busy || (x->ordered_stream_set[k] && p.user_stream == x->ordered_stream[k]);
                if (!x->ordered_stream_set[k] || !busy) lane = k;
            }
            if (lane < 0) {
                if ((rc = dpq_finish(x))) return rc;
                lane = 0;
            }
            x->ordered_stream[lane] = user;
            x->ordered_stream_set[lane] = true;
        }
        switch_lane(x, lane);
    }
    const int D = x->M * x->Ds;
    for (int base = 0; base < nq; base += kMaxBatchQueries) {
        const int n = std::min(kMaxBatchQueries, nq - base);
        if ((int)x->pending.size() >= dpq_index::kFlagSlots - 1 && (rc = dpq_finish(x))) return rc;
        const int slot = 1 + (int)x->pending.size();
        hipStream_t stream = user;
        if (overlap) {
            // the batch runs on its lane's stream once the caller's stream has reached this point (its inputs are
            // there); a lane's batches follow each other on the lane's stream, so its workspace is never shared
            const int lane = (int)(x->async_seq++ & 1);
            if (!x->lane_stream[lane]) {
                DPQ_HIP(hipStreamCreateWithFlags(&x->lane_stream[lane], hipStreamNonBlocking));
                DPQ_HIP(hipEventCreateWithFlags(&x->lane_ready[lane], hipEventDisableTiming));
            }
            switch_lane(x, lane);
            stream = x->lane_stream[lane];
            DPQ_HIP(hipEventRecord(x->lane_ready[lane], user));
            DPQ_HIP(hipStreamWaitEvent(stream, x->lane_ready[lane], 0));
            if (!x->lane_select_done[lane]) DPQ_HIP(hipEventCreateWithFlags(&x->lane_select_done[lane], hipEventDisableTiming));
            x->run_lane = lane;
            x->gate_this_batch = x->pending.size() == 1;
        }
        rc = run_batch(x, d_queries + (size_t)base * D, n, top_k, d_ids + (size_t)base * top_k,
                       d_dists + (size_t)base * top_k, stream, slot);
        if (x->run_lane >= 0 && !rc) {
            hipError_t e = hipEventRecord(x->lane_select_done[x->run_lane], stream);
            x->lane_select_recorded[x->run_lane] = e == hipSuccess;
        }
        x->run_lane = -1;
        if (rc) return rc;
        x->pending.push_back({d_queries + (size_t)base * D, n, top_k, d_ids + (size_t)base * top_k,
                              d_dists + (size_t)base * top_k, stream, user, slot, host_slot});
    }
    if (x->prof) {
        x->prof_acc.query_batches++;
        x->prof_acc.queries += nq;
    }
    return DPQ_OK;
    }
}
}  // namespace

int dpq_query_batch_device(dpq_index* x, const float* d_queries, int nq, int top_k, int32_t* d_ids, float* d_dists,
                           void* hip_stream) {
    return guarded([&]() -> int {
    if (x && !x->pending.empty()) {  // keep the order of the batches on this index
        int rc = dpq_finish(x);
        if (rc) return rc;
    }
    if (!x || !d_queries || !d_ids || !d_dists || nq < 0) return fail(DPQ_ERR_ARG, "NULL argument or nq < 0");
    if (!x->d_codebook) return fail(DPQ_ERR_STATE, "dpq_set_codebook has not been called");
    if (top_k < 1 || top_k > dpq::kMaxTopK) return fail(DPQ_ERR_ARG, "top_k must be in 1..2048");
    if ((int64_t)top_k > x->img.n_codes_total)
        return fail(DPQ_ERR_TOPK, "top_k exceeds the number of codes in the index");
    if (nq == 0) return DPQ_OK;
    DPQ_HIP(hipSetDevice(x->device));
    hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
    const int D = x->M * x->Ds;
    for (int base = 0; base < nq; base += kMaxBatchQueries) {
        const int n = std::min(kMaxBatchQueries, nq - base);
        int rc = run_batch(x, d_queries + (size_t)base * D, n, top_k, d_ids + (size_t)base * top_k,
                           d_dists + (size_t)base * top_k, stream);
        if (rc) return rc;
    }
    if (x->prof) {
        x->prof_acc.query_batches++;
        x->prof_acc.queries += nq;
    }
    return DPQ_OK;
    });
}

int dpq_query_batch(dpq_index* x, const float* queries, int nq, int top_k, int32_t* ids, float* dists) {
    return guarded([&]() -> int {
    if (!x || !queries || !ids || !dists || nq < 0) return fail(DPQ_ERR_ARG, "NULL argument or nq < 0");
    if (!x->d_codebook) return fail(DPQ_ERR_STATE, "dpq_set_codebook has not been called");
    if (nq == 0) return DPQ_OK;
    if (top_k < 1 || top_k > dpq::kMaxTopK) return fail(DPQ_ERR_ARG, "top_k must be in 1..2048");
    DPQ_HIP(hipSetDevice(x->device));
    const size_t qf = (size_t)nq * x->M * x->Ds, oe = (size_t)nq * top_k;
    if (qf > x->q_stage_floats) {
        hipFree(x->d_q_stage);
        x->d_q_stage = nullptr;
        int rc = dev_alloc(&x->d_q_stage, qf);
        if (rc) return rc;
        x->q_stage_floats = qf;
    }
    if (oe > x->out_stage_elems) {
        hipFree(x->d_ids_stage);
        hipFree(x->d_dists_stage);
        x->d_ids_stage = nullptr;
        x->d_dists_stage = nullptr;
        int rc = dev_alloc(&x->d_ids_stage, oe);
        if (!rc) rc = dev_alloc(&x->d_dists_stage, oe);
        if (rc) return rc;
        x->out_stage_elems = oe;
    }
    DPQ_HIP(hipMemcpy(x->d_q_stage, queries, qf * sizeof(float), hipMemcpyHostToDevice));
    int rc = dpq_query_batch_device(x, x->d_q_stage, nq, top_k, x->d_ids_stage, x->d_dists_stage, nullptr);
    if (rc) return rc;
    DPQ_HIP(hipDeviceSynchronize());
    DPQ_HIP(hipMemcpy(ids, x->d_ids_stage, oe * sizeof(int32_t), hipMemcpyDeviceToHost));
    DPQ_HIP(hipMemcpy(dists, x->d_dists_stage, oe * sizeof(float), hipMemcpyDeviceToHost));
    return DPQ_OK;
    });
}

// The reference's interface is host vectors in, host results out (h:2805-2810), one call per query (main:328-339).  Pipelined:
// the queries of batch i + 1 go up and the results of batch i - 1 come down (two copy streams) beside batch i's kernels
// (the two lanes of dpq_query_batch_device_async); dpq_finish settles everything and answers again, synchronously, any
// batch in which a query overflowed its candidate buffers.
int dpq_query_batch_host_async(dpq_index* x, const float* queries, int nq, int top_k, int32_t* ids, float* dists) {
    return guarded([&]() -> int {
    if (!x || !queries || !ids || !dists || nq < 0) return fail(DPQ_ERR_ARG, "NULL argument or nq < 0");
    int rc = check_batch_args(x, queries, nq, top_k, ids, dists);
    if (rc || nq == 0) return rc;
    DPQ_HIP(hipSetDevice(x->device));
    if (!x->copy_in) {
        DPQ_HIP(hipStreamCreateWithFlags(&x->copy_in, hipStreamNonBlocking));
        DPQ_HIP(hipStreamCreateWithFlags(&x->copy_out, hipStreamNonBlocking));
    }
    // batches enqueued through another entry point (another caller stream) are settled first: enqueue_async does that
    int slot = (int)(x->host_seq % dpq_index::kHostSlots);
    if (x->host_slots[slot].busy) {  // every slot in flight: settle them all (the oldest is the one wanted)
        if ((rc = dpq_finish(x))) return rc;
    }
    x->host_seq++;
    dpq_index::HostSlot& hs = x->host_slots[slot];
    const size_t qf = (size_t)nq * x->M * x->Ds, oe = (size_t)nq * top_k;
    if (qf > hs.qf) {
        hipFree(hs.d_q);
        hs.d_q = nullptr;
        hs.qf = 0;
        if ((rc = dev_alloc(&hs.d_q, qf))) return rc;
        hs.qf = qf;
    }
    // Page-locked result buffers (dpq_pin_host / hipHostRegister / hipHostMalloc) are mapped into the device's address
    // space: the select kernel then writes the lists straight into them (posted writes over PCIe) and no copy is
    // enqueued at all -- a hipMemcpyAsync costs the host ~10 us, three of them per batch held the pipelined rate at the
    // synchronous call's.  Pageable buffers take staging buffers and copies on copy_out.
    void *m_ids = nullptr, *m_d = nullptr;
    const bool direct = hipHostGetDevicePointer(&m_ids, ids, 0) == hipSuccess && hipHostGetDevicePointer(&m_d, dists, 0) == hipSuccess;
    if (!direct) (void)hipGetLastError();  // (an unregistered pointer is not an error of this call)
    if (!direct && oe > hs.oe) {
        hipFree(hs.d_ids);
        hipFree(hs.d_d);
        hs.d_ids = nullptr;
        hs.d_d = nullptr;
        hs.oe = 0;
        if ((rc = dev_alloc(&hs.d_ids, oe)) || (rc = dev_alloc(&hs.d_d, oe))) return rc;
        hs.oe = oe;
    }
    if (!hs.kernels_done) DPQ_HIP(hipEventCreateWithFlags(&hs.kernels_done, hipEventDisableTiming));
    // (pageable caller memory makes this copy synchronous with the host; pinned memory -- dpq_pin_host -- lets it overlap)
    DPQ_HIP(hipMemcpyAsync(hs.d_q, queries, qf * sizeof(float), hipMemcpyHostToDevice, x->copy_in));
    const size_t first = x->pending.size();
    int32_t* const out_ids = direct ? static_cast<int32_t*>(m_ids) : hs.d_ids;
    float* const out_d = direct ? static_cast<float*>(m_d) : hs.d_d;
    if ((rc = enqueue_async(x, hs.d_q, nq, top_k, out_ids, out_d, x->copy_in, true, slot))) return rc;
    // (enqueue_async may have settled older batches: the entries of this one are the pending tail)
    const size_t begin = std::min(first, x->pending.size());
    hs.h_ids = ids;
    hs.h_d = dists;
    hs.n_out = oe;
    hs.busy = true;
    hs.redo = false;
    hs.direct = direct;
    if (direct) return DPQ_OK;
    // results down once the batch's last kernel is through: the copy waits on every stream the batch's parts ran on
    std::vector<hipStream_t> seen;
    for (size_t i = begin; i < x->pending.size(); ++i) {
        const auto& p = x->pending[i];
        if (p.host_slot != slot || std::find(seen.begin(), seen.end(), p.stream) != seen.end()) continue;
        seen.push_back(p.stream);
        DPQ_HIP(hipEventRecord(hs.kernels_done, p.stream));
        DPQ_HIP(hipStreamWaitEvent(x->copy_out, hs.kernels_done, 0));
    }
    DPQ_HIP(hipMemcpyAsync(ids, hs.d_ids, oe * sizeof(int32_t), hipMemcpyDeviceToHost, x->copy_out));
    DPQ_HIP(hipMemcpyAsync(dists, hs.d_d, oe * sizeof(float), hipMemcpyDeviceToHost, x->copy_out));
    return DPQ_OK;
    });
}

// Page-locks / releases caller memory (hipHostRegister) so that dpq_query_batch_host_async's copies run beside the kernels;
// for callers that do not link the HIP runtime themselves.
int dpq_pin_host(void* ptr, int64_t bytes) {
    return guarded([&]() -> int {
    if (!ptr || bytes <= 0) return fail(DPQ_ERR_ARG, "NULL pointer or no bytes");
    DPQ_HIP(hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault));
    return DPQ_OK;
    });
}

int dpq_unpin_host(void* ptr) {
    return guarded([&]() -> int {
    if (!ptr) return fail(DPQ_ERR_ARG, "NULL pointer");
    DPQ_HIP(hipHostUnregister(ptr));
    return DPQ_OK;
    });
}

int dpq_merge_topk_host(const int32_t* ids, const float* dists, int n_lists, int nq, int top_k, int32_t* out_ids,
                        float* out_dists) {
    return guarded([&]() -> int {
    if (!ids || !dists || !out_ids || !out_dists || n_lists < 1 || nq < 0 || top_k < 1)
        return fail(DPQ_ERR_ARG, "bad merge argument");
    std::vector<uint64_t> keys;
    for (int q = 0; q < nq; ++q) {
        keys.clear();
        for (int l = 0; l < n_lists; ++l)
            for (int r = 0; r < top_k; ++r) {
                const size_t o = ((size_t)l * nq + q) * top_k + r;
                if (ids[o] < 0) continue;
                uint32_t bits;
                memcpy(&bits, &dists[o], 4);
                keys.push_back(((uint64_t)bits << 32) | (uint32_t)ids[o]);
            }
        const size_t kk = std::min((size_t)top_k, keys.size());
        std::partial_sort(keys.begin(), keys.begin() + kk, keys.end());
        for (int r = 0; r < top_k; ++r) {
            const size_t o = (size_t)q * top_k + r;
            if ((size_t)r < kk) {
                out_ids[o] = (int32_t)(keys[r] & 0xffffffffu);
                uint32_t bits = (uint32_t)(keys[r] >> 32);
                memcpy(&out_dists[o], &bits, 4);
            } else {
                out_ids[o] = -1;
                out_dists[o] = INFINITY;
            }
        }
    }
    return DPQ_OK;
    });
}

int dpq_merge_topk_device(const int32_t* d_ids, const float* d_dists, int n_lists, int nq, int top_k,
                          int32_t* d_out_ids, float* d_out_dists, int device, void* hip_stream) {
    return guarded([&]() -> int {
    if (!d_ids || !d_dists || !d_out_ids || !d_out_dists || n_lists < 1 || nq < 0 || top_k < 1)
        return fail(DPQ_ERR_ARG, "bad merge argument");
    if ((int64_t)n_lists * top_k > 16384) return fail(DPQ_ERR_ARG, "n_lists * top_k exceeds 16384");
    DPQ_HIP(hipSetDevice(device));
    DPQ_HIP(dpq::launch_merge(d_ids, d_dists, n_lists, nq, top_k, top_k, d_out_ids, d_out_dists,
                              reinterpret_cast<hipStream_t>(hip_stream)));
    return DPQ_OK;
    });
}

int dpq_merge_topk_device_packed(const int32_t* d_packed, int n_lists, int nq, int top_k, int32_t* d_out_ids,
                                 float* d_out_dists, int device, void* hip_stream) {
    return guarded([&]() -> int {
    if (!d_packed || !d_out_ids || !d_out_dists || n_lists < 1 || nq < 0 || top_k < 1)
        return fail(DPQ_ERR_ARG, "bad merge argument");
    if ((int64_t)n_lists * top_k > 16384) return fail(DPQ_ERR_ARG, "n_lists * top_k exceeds 16384");
    DPQ_HIP(hipSetDevice(device));
    DPQ_HIP(dpq::launch_merge(d_packed, reinterpret_cast<const float*>(d_packed + top_k), n_lists, nq, top_k, 2 * top_k,
                              d_out_ids, d_out_dists, reinterpret_cast<hipStream_t>(hip_stream)));
    return DPQ_OK;
    });
}

// Developer hook: time `reps` full-index filter-scan
// launches for nq query slots with the filter pinned (pass_all == 0: nothing
// survives; 1: everything survives), to separate decode/ADC cost from
// candidate handling.  Needs a prior dpq_query_batch* call with >= nq queries.
int dpq_debug_scan_time(dpq_index* x, int nq, int pass_all, int reps, int splits, float* ms_out) {
    return guarded([&]() -> int {
    DPQ_DEV_ONLY();
    if (!x || !ms_out || !x->d_lut32) return fail(DPQ_ERR_STATE, "run a query batch first");
    DPQ_HIP(hipSetDevice(x->device));
    if (pass_all == 3) {
        // the last batch's first filter level exactly as it ran: its bootstrap again (thresholds + tables), then `reps`
        // timed launches of its scan
        if (x->dbg_groups <= 0 || x->dbg_boot_slots <= 0) return fail(DPQ_ERR_STATE, "the last batch ran no bootstrap + scan");
        hipEvent_t a, b;
        DPQ_HIP(hipEventCreate(&a));
        DPQ_HIP(hipEventCreate(&b));
        DPQ_HIP(dpq::launch_bootstrap(x->dbg_ba, x->M, x->dbg_boot_slots, nullptr));
        DPQ_HIP(dpq::launch_scan(x->dbg_sa, x->dbg_groups, x->dbg_splits, nullptr));
        DPQ_HIP(hipEventRecord(a, nullptr));
        for (int r = 0; r < reps; ++r) {
            // every launch starts its in-scan tightening from empty histograms, as a batch's first level does (the fill
            // is a 1-2 us kernel inside the timed region)
            if (x->dbg_sa.tight_hist)
                DPQ_HIP(hipMemsetAsync(x->dbg_sa.tight_hist, 0, sizeof(uint32_t) * dpq::kTightWords * (size_t)x->dbg_boot_slots, nullptr));
            DPQ_HIP(dpq::launch_scan(x->dbg_sa, x->dbg_groups, x->dbg_splits, nullptr));
        }
        DPQ_HIP(hipEventRecord(b, nullptr));
        DPQ_HIP(hipEventSynchronize(b));
        float ms = 0;
        DPQ_HIP(hipEventElapsedTime(&ms, a, b));
        *ms_out = ms / reps;
        hipEventDestroy(a);
        hipEventDestroy(b);
        return DPQ_OK;
    }
    const int QG = dpq::queries_per_group(x->M);
    const int nqp = (nq + QG - 1) / QG * QG;
    if (nqp > x->ws_slots) return fail(DPQ_ERR_ARG, "nq exceeds the workspace");
    dpq::ScanArgs sa{};
    sa.img = x->img;
    sa.fp32_accum = x->plain ? 1 : 0;
    sa.lut32 = x->d_lut32;
    sa.lut_min = x->d_lut_min;
    sa.thr_key = x->d_thr_key;
    sa.slot_query = nullptr;
    sa.n_queries = nq;
    sa.debug_pass = pass_all == 2 ? 0 : (pass_all ? 2 : 1);  // 2: the thresholds the last batch left behind
    sa.seg_list = nullptr;
    sa.n_seg_pass = x->img.n_segments;
    // what the last batch ran: its plain-code scratch, if it decoded the whole shard into one
    if (!x->plain && x->d_batch_raw && batch_tile_segments(x) >= x->img.n_segments && !getenv("DPQ_DEBUG_FUSED")) {
        sa.img.raw = x->d_batch_raw;
        if (x->d_relabel) sa.lut32 = x->d_lut32r;
    }
    if (const char* e = getenv("DPQ_DEBUG_NSEG")) sa.n_seg_pass = std::min(x->img.n_segments, atoi(e));
    if (splits <= 0) splits = splits_for(sa.n_seg_pass, nqp / QG);
    sa.cand_count = x->d_cand_count;
    sa.cand_key = x->d_cand_key;
    sa.cand_stride = x->ws_cap;
    sa.region_off = 0;
    sa.region_cap = std::max(1, x->ws_cap / splits);
    sa.qtab = x->d_qtab;
    hipEvent_t a, b;
    DPQ_HIP(hipEventCreate(&a));
    DPQ_HIP(hipEventCreate(&b));
    DPQ_HIP(dpq::launch_quantise(sa, nqp / QG, nullptr));
    DPQ_HIP(dpq::launch_scan(sa, nqp / QG, splits, nullptr));
    DPQ_HIP(hipEventRecord(a, nullptr));
    for (int r = 0; r < reps; ++r) DPQ_HIP(dpq::launch_scan(sa, nqp / QG, splits, nullptr));
    DPQ_HIP(hipEventRecord(b, nullptr));
    DPQ_HIP(hipEventSynchronize(b));
    float ms = 0;
    DPQ_HIP(hipEventElapsedTime(&ms, a, b));
    *ms_out = ms / reps;
    hipEventDestroy(a);
    hipEventDestroy(b);
    if (getenv("DPQ_DEBUG_WG_TIMES")) {  // when do the workgroups of one launch start and end?
        const int nwg = splits * (nqp / QG);
        unsigned long long* d_t = nullptr;
        int rc = dev_alloc(&d_t, (size_t)nwg * 2);
        if (rc) return rc;
        sa.wg_times = d_t;
        DPQ_HIP(dpq::launch_scan(sa, nqp / QG, splits, nullptr));
        DPQ_HIP(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)nwg * 2);
        DPQ_HIP(hipMemcpy(h.data(), d_t, h.size() * 8, hipMemcpyDeviceToHost));
        hipFree(d_t);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < nwg; ++i) t0 = std::min(t0, h[2 * (size_t)i]), t1 = std::max(t1, h[2 * (size_t)i + 1]);
        std::vector<double> st, en, life;
        for (int i = 0; i < nwg; ++i) {
            st.push_back((double)(h[2 * (size_t)i] - t0) / 100.0);
            en.push_back((double)(h[2 * (size_t)i + 1] - t0) / 100.0);
            life.push_back(en.back() - st.back());
        }
        std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end()); std::sort(life.begin(), life.end());
        auto pct = [&](const std::vector<double>& v, double p) { return v[(size_t)(p * (v.size() - 1))]; };
        fprintf(stderr, "scan workgroups (%d, %s codes): span %.1f us; start p50 %.1f max %.1f; end min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f; "
                        "lifetime min %.1f p50 %.1f max %.1f us; busy %.1f %%\n", nwg, sa.img.raw ? "plain" : "compressed",
                (double)(t1 - t0) / 100.0, pct(st, 0.5), st.back(), en.front(), pct(en, 0.1), pct(en, 0.5), pct(en, 0.9), en.back(),
                life.front(), pct(life, 0.5), life.back(),
                100.0 * std::accumulate(life.begin(), life.end(), 0.0) / (nwg * (double)(t1 - t0) / 100.0));
    }
    return DPQ_OK;
    });
}

// Developer hook (not in the public header): one full-index filter-scan launch of the STAMPS build of the
// scan kernel (s_memtime brackets around the sections of the loop) with the thresholds the last batch
// left behind; out[0..n) = per-section cycle sums over all wavefronts (order: enum in dpq_kernels.hip).
int dpq_debug_scan_stamps(dpq_index* x, int nq, int splits, unsigned long long* out, int n_out, float* ms_out) {
    return guarded([&]() -> int {
    DPQ_DEV_ONLY();
    if (!x || !out || !x->d_lut32) return fail(DPQ_ERR_STATE, "run a query batch first");
    if (x->M != 8) return fail(DPQ_ERR_ARG, "the STAMPS build exists for M = 8");
    DPQ_HIP(hipSetDevice(x->device));
    const int QG = dpq::queries_per_group(x->M);
    const int nqp = (nq + QG - 1) / QG * QG;
    if (nqp > x->ws_slots) return fail(DPQ_ERR_ARG, "nq exceeds the workspace");
    const int n = std::min(n_out, dpq::scan_stamp_count());
    unsigned long long* d_st = nullptr;
    int rc = dev_alloc(&d_st, (size_t)dpq::scan_stamp_count());
    if (rc) return rc;
    DPQ_HIP(hipMemset(d_st, 0, sizeof(unsigned long long) * dpq::scan_stamp_count()));
    dpq::ScanArgs sa{};
    sa.img = x->img;
    sa.fp32_accum = x->plain ? 1 : 0;
    sa.lut32 = x->d_lut32;
    sa.lut_min = x->d_lut_min;
    sa.thr_key = x->d_thr_key;
    sa.n_queries = nq;
    sa.n_seg_pass = x->img.n_segments;
    if (const char* e = getenv("DPQ_DEBUG_NSEG")) sa.n_seg_pass = std::min(x->img.n_segments, atoi(e));
    if (splits <= 0) splits = splits_for(sa.n_seg_pass, nqp / QG);
    sa.cand_count = x->d_cand_count;
    sa.cand_key = x->d_cand_key;
    sa.cand_stride = x->ws_cap;
    sa.region_off = 0;
    sa.region_cap = std::max(1, x->ws_cap / splits);
    sa.qtab = x->d_qtab;
    sa.stamps = d_st;
    hipEvent_t a, b;
    DPQ_HIP(hipEventCreate(&a));
    DPQ_HIP(hipEventCreate(&b));
    DPQ_HIP(dpq::launch_quantise(sa, nqp / QG, nullptr));
    DPQ_HIP(hipEventRecord(a, nullptr));
    DPQ_HIP(dpq::launch_scan(sa, nqp / QG, splits, nullptr));
    DPQ_HIP(hipEventRecord(b, nullptr));
    DPQ_HIP(hipEventSynchronize(b));
    float ms = 0;
    DPQ_HIP(hipEventElapsedTime(&ms, a, b));
    if (ms_out) *ms_out = ms;
    DPQ_HIP(hipMemcpy(out, d_st, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
    hipEventDestroy(a);
    hipEventDestroy(b);
    hipFree(d_st);
    return DPQ_OK;
    });
}

// Developer hook: per-wavefront marks of strand1_kernel (100 MHz clock): [256 workgroups][16 wavefronts][16 marks] =
// start, end of the prologue, end of each of the wavefront's first strips.  First call arms it; later calls copy out
// the marks of the last one-query call on a strand image.
int dpq_debug_strand1_stamps(dpq_index* x, unsigned long long* out, int n_words) {
    return guarded([&]() -> int {
    DPQ_DEV_ONLY();
    if (!x || !out) return fail(DPQ_ERR_ARG, "NULL argument");
    DPQ_HIP(hipSetDevice(x->device));
    const size_t total = (size_t)256 * 16 * 16;
    if (!x->d_s1_stamps) {
        int rc = dev_alloc(&x->d_s1_stamps, total);
        if (rc) return rc;
        DPQ_HIP(hipMemset(x->d_s1_stamps, 0, total * 8));
        return DPQ_OK;
    }
    DPQ_HIP(hipDeviceSynchronize());
    DPQ_HIP(hipMemcpy(out, x->d_s1_stamps, std::min((size_t)std::max(n_words, 0), total) * 8, hipMemcpyDeviceToHost));
    DPQ_HIP(hipMemset(x->d_s1_stamps, 0, total * 8));
    return DPQ_OK;
    });
}

// Developer hook: phase marks of the bootstrap kernel.  First call arms it
// (allocates [2048][8] marks); later calls return the mean cycles between consecutive marks over `nq` slots
// of the last batch: out[0] rank, [1] cell counts + prefix, [2] node evaluation, [3] k-th key select.
int dpq_debug_boot_stamps(dpq_index* x, int nq, double* out) {
    return guarded([&]() -> int {
    DPQ_DEV_ONLY();
    if (!x || !out) return fail(DPQ_ERR_ARG, "NULL argument");
    DPQ_HIP(hipSetDevice(x->device));
    if (!x->d_boot_stamps) {
        int rc = dev_alloc(&x->d_boot_stamps, (size_t)kMaxBatchQueries * 16);
        if (rc) return rc;
        DPQ_HIP(hipMemset(x->d_boot_stamps, 0, sizeof(unsigned long long) * kMaxBatchQueries * 16));
        for (int i = 0; i < 8; ++i) out[i] = 0;
        return DPQ_OK;
    }
    DPQ_HIP(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)kMaxBatchQueries * 16);
    DPQ_HIP(hipMemcpy(h.data(), x->d_boot_stamps, h.size() * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) out[i] = 0;
    nq = std::min(nq, kMaxBatchQueries);
    // out[0..3]: bootstrap (rank, cells, evaluate, select); out[4..7]: the last select launch (gather, k-th key,
    // winners, sort + output)
    for (int half = 0; half < 2; ++half) {
        // blocks on the 100 MHz chip-wide clock: first start -> last end, distribution of lifetimes and start offsets
        const size_t o = (size_t)half * kMaxBatchQueries * 8;
        unsigned long long t0 = ~0ull, t1 = 0;
        std::vector<double> life, start;
        for (int q = 0; q < nq; ++q) {
            t0 = std::min(t0, h[o + (size_t)q * 8 + 6]);
            t1 = std::max(t1, h[o + (size_t)q * 8 + 7]);
        }
        for (int q = 0; q < nq; ++q) {
            life.push_back((double)(h[o + (size_t)q * 8 + 7] - h[o + (size_t)q * 8 + 6]) / 100.0);
            start.push_back((double)(h[o + (size_t)q * 8 + 6] - t0) / 100.0);
        }
        std::sort(life.begin(), life.end());
        std::sort(start.begin(), start.end());
        auto pct = [&](const std::vector<double>& v, double p) { return v[(size_t)(p * (v.size() - 1))]; };
        fprintf(stderr, "%s blocks: span %.2f us; lifetime min %.2f median %.2f p90 %.2f max %.2f us; start offset median %.2f p90 %.2f max %.2f us\n",
                half ? "select" : "bootstrap", (double)(t1 - t0) / 100.0, life.front(), pct(life, 0.5), pct(life, 0.9), life.back(),
                pct(start, 0.5), pct(start, 0.9), start.back());
        if (!half) {  // bootstrap blocks by the rounds of cells they walked (slot 5 of a block's stamps)
            double sum[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0};
            int cnt[4] = {0, 0, 0, 0};
            for (int q = 0; q < nq; ++q) {
                const int r = (int)std::min<unsigned long long>(h[o + (size_t)q * 8 + 5], 4) - 1;
                if (r < 0) continue;
                const double l = (double)(h[o + (size_t)q * 8 + 7] - h[o + (size_t)q * 8 + 6]) / 100.0;
                sum[r] += l, mx[r] = std::max(mx[r], l), cnt[r]++;
            }
            for (int r = 0; r < 4; ++r)
                if (cnt[r]) fprintf(stderr, "  %d%s round(s): %d blocks, lifetime mean %.2f max %.2f us\n", r + 1, r == 3 ? "+" : "", cnt[r], sum[r] / cnt[r], mx[r]);
        }
    }
    for (int half = 0; half < 2; ++half)
        for (int q = 0; q < nq; ++q)
            for (int i = 0; i < 4; ++i) {
                const size_t o = (size_t)half * kMaxBatchQueries * 8 + (size_t)q * 8;
                out[4 * half + i] += (double)(h[o + i + 1] - h[o + i]) / nq;
            }
    return DPQ_OK;
    });
}

// Developer hook: time the level-0 select (shared, query-independent candidate list).
int dpq_debug_select_time(dpq_index* x, int nq, int top_k, int flags, int reps, float* ms_out) {
    return guarded([&]() -> int {
    DPQ_DEV_ONLY();
    if (!x || !ms_out || !x->d_lut32 || !x->d_l0_id) return fail(DPQ_ERR_STATE, "run a query batch first");
    DPQ_HIP(hipSetDevice(x->device));
    dpq::SelectArgs se{};
    se.shared_id = x->d_l0_id;
    se.shared_code = x->d_l0_code;
    se.shared_n = x->l0_segments * dpq::kChunk * x->img.chunks_per_segment;
    se.cand_count = x->d_cand_count;
    se.cand_key = x->d_cand_key;
    se.cand_stride = x->ws_cap;
    se.region_off = top_k;
    se.scratch = x->d_scratch;
    se.lut32 = x->d_lut32;
    se.top_k = top_k;
    se.final_pass = 0;
    se.thr_key = x->d_thr_key;
    se.overflow = x->d_overflow;
    se.n_codes_total = x->img.n_codes_total;
    (void)flags;
    hipEvent_t a, b;
    DPQ_HIP(hipEventCreate(&a));
    DPQ_HIP(hipEventCreate(&b));
    DPQ_HIP(dpq::launch_select(se, x->M, nq, nullptr));
    DPQ_HIP(hipEventRecord(a, nullptr));
    for (int r = 0; r < reps; ++r) DPQ_HIP(dpq::launch_select(se, x->M, nq, nullptr));
    DPQ_HIP(hipEventRecord(b, nullptr));
    DPQ_HIP(hipEventSynchronize(b));
    float ms = 0;
    DPQ_HIP(hipEventElapsedTime(&ms, a, b));
    *ms_out = ms / reps;
    hipEventDestroy(a);
    hipEventDestroy(b);
    return DPQ_OK;
    });
}

int dpq_profile_enable(dpq_index* x, int on) {
    return guarded([&]() -> int {
    if (!x) return fail(DPQ_ERR_ARG, "NULL index");
    x->prof = on != 0;
    x->prof_scan_only = on == 2;
    return DPQ_OK;
    });
}

int dpq_profile_reset(dpq_index* x) {
    return guarded([&]() -> int {
    if (!x) return fail(DPQ_ERR_ARG, "NULL index");
    hipSetDevice(x->device);
    for (auto& ep : x->events) {
        x->ev_pool.push_back(ep.a);
        x->ev_pool.push_back(ep.b);
    }
    x->events.clear();
    memset(&x->prof_acc, 0, sizeof x->prof_acc);
    if (x->d_counters) hipMemset(x->d_counters, 0, 16);
    return DPQ_OK;
    });
}

int dpq_profile_read(dpq_index* x, dpq_profile* out) {
    return guarded([&]() -> int {
    if (!x || !out) return fail(DPQ_ERR_ARG, "NULL argument");
    DPQ_HIP(hipSetDevice(x->device));
    if (x->prof_failed) {
        x->prof_failed = false;
        return fail(DPQ_ERR_HIP, "a profiling event could not be created or recorded; timings are incomplete");
    }
    for (auto& ep : x->events) {
        DPQ_HIP(hipEventSynchronize(ep.b));
        float ms = 0.f;
        DPQ_HIP(hipEventElapsedTime(&ms, ep.a, ep.b));
        if (ep.kind == 0) x->prof_acc.lut_ms += ms;
        if (ep.kind == 1) x->prof_acc.scan_ms += ms;
        if (ep.kind == 2) x->prof_acc.select_ms += ms;
        if (ep.kind == 3) x->prof_acc.quantise_ms += ms;
        if (ep.kind == 4) x->prof_acc.decode_ms += ms;
        if (ep.kind == 5) x->prof_acc.bootstrap_ms += ms;
        x->ev_pool.push_back(ep.a);
        x->ev_pool.push_back(ep.b);
    }
    x->events.clear();
    if (x->d_counters) {
        unsigned long long h[2] = {0, 0};
        DPQ_HIP(hipMemcpy(h, x->d_counters, 16, hipMemcpyDeviceToHost));
        x->prof_acc.exact_checks = (int64_t)h[0];
        x->prof_acc.candidates = (int64_t)h[1];
    }
    *out = x->prof_acc;
    return DPQ_OK;
    });
}

}  // extern "C"
