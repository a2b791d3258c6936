// dpq_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the DeltaPQ
// query path.  Reference being replaced: the per-query loop of
// query_processing_scan_compressed_codes_opt_o_direct
// (/root/reference/deltapq_create_approx_tree.h:2805-2984; "h:" below,
// "main:" = deltapq_approx_tree_main.cpp).
//
//   lut_build_kernel        a3: T[m][k] = sum_d (c[m][k][d] - q[m*Ds+d])^2 with the
//                           reference's mixed fp32/fp64 arithmetic (h:2841-2849).
//   bootstrap_kernel        the first threshold of a query (shards >= 64 K nodes): the nodes of
//                           the query's best cells of an inverted multi-index over sub-space
//                           pairs, evaluated in fp32; an upper bound of the k-th key; also writes
//                           the slot's fields of the first filter level's tables.
//   quantise_kernel         conservative 8-bit lower-bound tables of a query group for one filter
//                           level, once per group (later levels, reruns).
//   decode_list_kernel      a5, batches of >= 3 query groups: a tile of a filter level's segment list decoded
//                           ONCE into a plain-code scratch (one wavefront per segment) that every group's
//                           scan pass reads through L2 / Infinity Cache; the bytes it writes are bank-aware
//                           LABELS of the code values (a permutation per sub-space, found at load).
//   scan_kernel             a5 + most of a6: delta decode + ADC filter + exact check.
//                           One wavefront = one 64-node chunk per step, three chunks in flight
//                           (software pipeline over the decode's two global round trips); child
//                           codes are rebuilt from parent + packed deltas by pointer jumping over
//                           ds_bpermute along parent lanes resolved at load; lower-bound
//                           distances are LDS table gathers (a lookup workload: no MFMA).  A
//                           workgroup keeps the filter tables of 64 queries in LDS (32 at
//                           M = 16) and decodes every chunk once for all of them -- or, behind
//                           decode_list_kernel, reads the chunk's plain codes (the <M, PLAIN>
//                           instantiation, also the -task pqscan comparator).  Nodes the
//                           filter lets through for some query are queued per wavefront and
//                           checked exactly (the reference's distance, whole (distance, id)
//                           keys) against the query's threshold key; what passes is a candidate
//                           key in the workgroup's own region of the query's buffer (no global
//                           atomics).  On one-level plans the LAST wavefront of a workgroup is a
//                           helper that lowers the slots' thresholds from the candidates the query
//                           group's workgroups have found so far (in-scan tightening).
//   stream_kernel           a5 + a6 for batches of up to four queries (the reference's own call shape and its
//                           neighbours): 1, 2 or 4 queries per pass over the compressed image, every decoded node
//                           against the queries' exact tables in LDS (no filter tables); a wavefront per chunk.
//   strand_kernel           the same pass over the STRAND image (M = 8, big shards): a LANE decodes a run of 64
//                           consecutive nodes with the reference's own stack machine (h:2888-2905), 64 of them
//                           side by side, ancestor stacks in LDS -- ~70 instructions per 64 nodes instead of ~300;
//                           bound by the exact-table gathers in LDS, 24 % of the HBM peak in payload bytes.
//   decode_segments_kernel  the same decode, writing plain codes (small shards: cascade level 0
//                           is a query-independent spread sample).
//   select_kernel           the rest of a6: candidate regions gathered; k-th smallest
//                           (distance, id) key = next threshold, winners carried to the next
//                           level; on the last level the sorted top-k.
//   merge_kernel            8e: merge of per-shard partial top-k lists.
//
// Top-k strategy (replaces the sequential size-k max-heap, h:2851-2853,
// 2909-2914): thresholds that are valid upper bounds of the final k-th key.
// The first one comes from the bootstrap (or, on small shards, from a spread
// sample evaluated exactly); filter levels then visit the segments in a
// low-discrepancy order, every segment exactly once, and keep the nodes whose
// key is <= the current threshold; a level's k-th best key tightens it.  The
// in-scan filter is a CONSERVATIVE LOWER BOUND of the distance in 8-bit fixed
// point (byte entries summed in 16-bit fields at M = 16; scaled to the query's
// threshold, rounded down, saturated), so it never drops a node the exact rule
// would keep.
#include "dpq_kernels.h"

#include <atomic>
#include <type_traits>
#include <cfloat>
#include <cmath>

namespace dpq {

static hipError_t ensure_dynamic_lds(const void* fn, size_t bytes, std::atomic<bool>* done);

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------

// (a & mask) | c in one VALU op; the mask travels in an SGPR (a VOP3 cannot carry a 32-bit literal, and
// without the asm the compiler emits v_and_b32 + v_or_b32)
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask, uint32_t c) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask), "v"(c));
    return r;
}

__device__ __forceinline__ uint32_t bperm(int src_lane, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}

// number of set bits of a wave-uniform 64-bit mask strictly below this lane
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m, uint32_t acc) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, acc));
}

__device__ __forceinline__ uint64_t make_key(float d, uint32_t id) {
    return ((uint64_t)__float_as_uint(d) << 32) | (uint64_t)id;
}

// ---------------------------------------------------------------------------
// a3: LUT build.  grid = (slots / kLutQueries, M), block = 256 (one thread per
// centroid; its codebook row is loaded once and reused for kLutQueries queries,
// whose sub-vectors are wave-uniform scalar loads).
// Output layout [query][m][256] fp32 (exact tables, read by select_kernel and
// by the scan prologue) + the per-(query, m) minima that anchor the filter
// quantisation.  Also clears the per-slot candidate counters and overflow
// flags of the batch (saves two fill launches per batch).
// ---------------------------------------------------------------------------
constexpr int kLutQueries = 8;
constexpr int kLutMinParts = 4;  // per (query, m): one partial minimum per wavefront of the 256-thread block

__device__ __forceinline__ float lut_min_of(const float* __restrict__ lut_min, size_t q, int M, int m) {
    const float4 p = *reinterpret_cast<const float4*>(lut_min + (q * M + m) * kLutMinParts);
    return fminf(fminf(p.x, p.y), fminf(p.z, p.w));
}

__global__ __launch_bounds__(256) void lut_build_kernel(const float* __restrict__ codebook,
                                                         const float* __restrict__ queries, int nq, int n_slots, int M,
                                                         int K, int Ds, float* __restrict__ lut,
                                                         float* __restrict__ lut_min, uint32_t* __restrict__ cand_count,
                                                         uint32_t* __restrict__ overflow, const uint8_t* __restrict__ relabel,
                                                         float* __restrict__ lut_labels, uint32_t* __restrict__ tight_hist) {
    // This kernel (and decode_list_kernel) runs UNDER the previous pipelined batch's scan: the scan's sixteen wavefronts per
    // CU take instruction issue by age, and what is left made a 15 us table build take 85 us (kernel-trace timeline,
    // profiles/r04_timeline_default.txt) -- on the critical path of the lane's next bootstrap.  Raised wave priority gives
    // these few wavefronts the issue slots they ask for; the scan loses as much as they use, no more.
#ifndef DPQ_UNDER_SCAN_PRIO
#define DPQ_UNDER_SCAN_PRIO 3
#endif
    __builtin_amdgcn_s_setprio(DPQ_UNDER_SCAN_PRIO);
    const int q0 = blockIdx.x * kLutQueries, m = blockIdx.y, k = threadIdx.x;
    if (m == 0 && k < kLutQueries && q0 + k < n_slots) {
        if (cand_count) cand_count[q0 + k] = 0;
        if (overflow) overflow[q0 + k] = 0;
    }
    // the first filter level's tightening histograms of these slots (scan_kernel)
    if (m == 0 && tight_hist)
        for (int i = k; i < kLutQueries * kTightWords; i += 256)
            if (q0 * kTightWords + i < n_slots * kTightWords) tight_hist[(size_t)q0 * kTightWords + i] = 0;
    if (q0 >= nq) return;  // padding slots only
    float acc[kLutQueries];
#pragma unroll
    for (int j = 0; j < kLutQueries; ++j) acc[j] = 0.0f;
    if (k < K) {
        const float* c = codebook + ((size_t)m * K + k) * Ds;
        const float* qv = queries + (size_t)q0 * M * Ds + (size_t)m * Ds;  // query j: qv + j * M * Ds (clamped to nq - 1)
        const size_t qstride = (size_t)M * Ds;
        // h:2845-2846: `float += pow(float - float, 2)`, d ascending
        auto step = [&](float& a, float cv, float qd) {
            const float diff = __fsub_rn(cv, qd);                      // fp32 subtract
            const double sq = __dmul_rn((double)diff, (double)diff);   // pow(.,2): exact in fp64
            a = (float)__dadd_rn((double)a, sq);                       // float += double
        };
        if ((Ds & 3) == 0) {  // rows are 16-byte aligned: one dwordx4 load per 4 dimensions
            const float4* c4 = reinterpret_cast<const float4*>(c);
            for (int d = 0; d < Ds; d += 4) {
                const float4 v = c4[d >> 2];
#pragma unroll
                for (int j = 0; j < kLutQueries; ++j) {
                    const float* qj = qv + (size_t)min(j, nq - 1 - q0) * qstride + d;
                    step(acc[j], v.x, qj[0]);
                    step(acc[j], v.y, qj[1]);
                    step(acc[j], v.z, qj[2]);
                    step(acc[j], v.w, qj[3]);
                }
            }
        } else {
            for (int d = 0; d < Ds; ++d) {
                const float cv = c[d];
#pragma unroll
                for (int j = 0; j < kLutQueries; ++j) step(acc[j], cv, qv[(size_t)min(j, nq - 1 - q0) * qstride + d]);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < kLutQueries; ++j) acc[j] = INFINITY;  // centroids beyond K do not exist; no valid code points at them
    }
#pragma unroll
    for (int j = 0; j < kLutQueries; ++j) {
        if (q0 + j < nq) lut[((size_t)(q0 + j) * M + m) * 256 + k] = acc[j];
        // second copy, rows by the LABELS of the per-batch plain-code scratch (the scan's exact checks and its filter
        // tables index it with the scratch's codes as they are)
        if (lut_labels && q0 + j < nq) lut_labels[((size_t)(q0 + j) * M + m) * 256 + relabel[m * 256 + k]] = acc[j];
        uint32_t v = __float_as_uint(acc[j]);  // acc >= 0: uint order == float order
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o));
        // one partial minimum per wavefront: the readers take the least of the four (lut_min_of).  No LDS in this
        // kernel: the scan fills every CU's LDS to the last KB, and a workgroup that wants even 128 B of it only
        // starts where a scan workgroup has retired -- the table build of the next pipelined batch then ran in the
        // scan's tail instead of under it.
        if ((k & 63) == 0 && q0 + j < nq) lut_min[((size_t)(q0 + j) * M + m) * kLutMinParts + (k >> 6)] = __uint_as_float(v);
    }
}

// ---------------------------------------------------------------------------
// a5: delta decode of one 64-node chunk by one wavefront.
// M = 8: the reference format (a code is 2 dwords, 8 stack levels).
// M = 16: this build's extension (4 dwords, 2-byte masks, 16 levels); the
// reference format stops at M = 8 (h:1765, 1791-1795, 2883).
// ---------------------------------------------------------------------------

template <int M>
struct Cfg {
    static constexpr int W = M / 4;                   // dwords per code
    static constexpr int LEVELS = M <= 8 ? 8 : 16;    // ancestor stack entries (h:2858-2864: M of them)
    static constexpr int JUMPS = M <= 8 ? 3 : 4;      // pointer-jumping rounds: 2^JUMPS > deepest in-chunk chain
    static constexpr int PLANES = M <= 8 ? 4 : 5;     // bits of popcount(mask)
    // ---- filter tables of the scan ----
    static constexpr int EB = 8;                      // bits per table entry (4 queries per table dword)
    static constexpr int AB = M <= 8 ? 8 : 16;        // bits per field of the accumulators the entries are summed in
    static constexpr int R = AB / EB;                 // accumulator dwords per table dword
    static constexpr int F = 32 / AB;                 // fields (= queries) per accumulator dword
    static constexpr int NG = M <= 8 ? 4 : 2;         // 16-byte entries per (m, code): NG * M * 256 * 16 B = 128 KB
    static constexpr int NA = NG * 4 * R;             // accumulator dwords per node
    static constexpr int QG = NA * F;                 // queries per scan workgroup: 64 (M = 8), 32 (M = 16)
    static constexpr int J = NA < AB ? NA : AB;       // accumulator dwords folded into one survivor-mask dword
    static constexpr int MD = (NA + AB - 1) / AB;     // survivor-mask dwords per lane
    // 8-bit geometry re-swept on the GPU with bootstrap thresholds (QT/SAT -> exact checks per query, scan ms per
    // 1000 queries): 32/20 5644 0.198, 40/21 3848 0.178, 48/22 3004 0.169, 56/23 2585 0.165, 64/24 2413 0.163,
    // 80/26 2474 0.164, 90/26 3129 0.170, 100/27 3756 0.178, 120/30 4789 0.189 -- saturation costs more than resolution
    // M = 16: sixteen byte entries at that resolution do not fit a byte sum (QT 110 / SAT 31 with the top bits set
    // aside every four sub-spaces: 3.3 x the exact checks of 16-bit entries and a slower scan), and 16-bit ENTRIES
    // (round 1) serve 16 queries per decode pass at 32 LDS bytes per pair.  So the entries are bytes, summed four
    // sub-spaces at a time as bytes (4 * SAT <= 255) and then widened into 16-bit accumulator fields: 32 queries
    // per pass, 16 LDS bytes per pair.  Swept on the GPU (entries summed as bytes / SAT / QT -> exact checks per
    // query, M q/s at top-1000 and top-100; 16-bit entries: 7775, 0.995, 1.32): 4/63/160 17766 0.91 1.67,
    // 4/63/200 15134 0.99 1.77, 4/63/250 13776 1.04 1.79, 4/63/300 14271 1.02 1.75, 4/63/400 22002 0.81 1.46,
    // 2/127/200 15075 0.95 1.58, 1/255/320 11686 0.97 1.47, 1/255/400 10710 0.99 1.47 -- every unit of bound an
    // entry loses to rounding lets more of a concentrated 16-sub-space distance distribution through, but the
    // widening instructions of the finer geometries cost as much as the checks they save.
#ifndef DPQ_QT16
#define DPQ_QT16 250
#define DPQ_SAT16 63
#define DPQ_PRE16 4
#endif
    // M = 8 with the in-scan tightening's headroom (8 SAT + BIAS + XMAX XU <= 255), scripts/sim_filter_nibbles.py, survivors
    // per query at a threshold of rank 800 / 400 / 200: 80/26 (no headroom) 2241 / 1139 / 550, 64/24 (none) 2156 / 1066 /
    // 501, 64/22 2414 / 1211 / 574, 72/23 2515 / 1284 / 620, 80/23 3113 / 1639 / 818
    // Round 3, last step: tightening steps of ONE unit (seven steps = 11 % of the span; finer steps locate the cut better
    // than a longer range reaches, as at M = 16) free a unit of saturation: 64/23 with XU 1 against 64/22 with XU 2 on the
    // GPU (scripts/gpu_xu8.sh): exact checks per query 1676 against 1810 at top-100 (candidates 400 / 401), 12 514 against
    // 13 344 at top-1000; scan launch 0.1148 / 0.1178 ms and 0.284 / 0.298 ms.
#ifndef DPQ_QT8
#define DPQ_QT8 64
#define DPQ_SAT8 23
#endif
    static constexpr int QT = M <= 8 ? DPQ_QT8 : DPQ_QT16;   // filter units that span (tau - sum of minima)
    static constexpr int SAT = M <= 8 ? DPQ_SAT8 : DPQ_SAT16; // entry saturation
    // In-scan tightening: a slot's cut can be lowered by e * XU filter units, e = 1 .. XMAX, while a scan launch runs
    // (the additive term of its accumulator field grows by as much): the field sums need that much headroom.
    static constexpr int XMAX = kTightBuckets - 1;
#ifndef DPQ_XU8
#define DPQ_XU8 1
#endif
#ifndef DPQ_XU16
#define DPQ_XU16 6  // swept on the GPU at top-1000 (scripts/gpu_xu16.sh; exact checks per query, ms per step): 4: 10823, 0.754; 6: 10295, 0.742; 8: 10806, 0.757; 12: 11957, 0.798; 16: 13133, 0.849; 24: 15919, 0.959 (top-100: 0.366 - 0.376)
#endif
    static constexpr int XU = M <= 8 ? DPQ_XU8 : DPQ_XU16;  // about a quarter of QT at XMAX steps: the k-th distance of a bootstrap
                                               // threshold of rank 8 k lies 17 % of (tau - minima) above the final one
    // field sum >= 2^(AB-1) (its top bit) <=> sum of entries > QT + 1.  R = 1: added to every m = 0 entry;
    // R = 2: the accumulator fields start from it.
    static constexpr int BIAS = (1 << (AB - 1)) - 1 - (QT + 1);
    static constexpr int FIELD_MAX = (1 << EB) - 1;
    static_assert(BIAS >= 0 && M * SAT + BIAS + XMAX * XU < (1 << AB), "a field sum must not carry into its neighbour");
    static_assert(XMAX * XU < QT, "cuts stay above zero");
    static constexpr int PRE = R == 1 ? M : DPQ_PRE16;  // entries summed as bytes before they are widened
    static_assert(R == 1 || (PRE * SAT <= FIELD_MAX && M * SAT > QT + 1 && M % PRE == 0), "byte sums of PRE entries; all-SAT rejects");
    static constexpr uint32_t LOW = AB == 8 ? 0x01010101u : 0x00010001u;  // bit 0 of every accumulator field
    // local slot of field f of accumulator dword acc: survivor-mask dword acc / AB, bit AB * f + acc % AB
    __host__ __device__ static constexpr int slot_of(int acc, int f) { return (acc / AB) * (J * F) + f * J + acc % AB; }
    // byte tb of dword c of 16-byte entry g of the tables <-> local slot: its accumulator is (4 g + c) R + tb % R
    // (R = 2: even bytes widen into one accumulator, odd bytes into the next), field tb / R
    __host__ __device__ static constexpr int slot_of_table(int g, int c, int tb) {
        return slot_of((4 * g + c) * R + tb % R, tb / R);
    }
    // the table entry of a slot nobody asks for: its field sum rejects every node
    __host__ __device__ static constexpr uint32_t reject_entry(int m) {
        return R == 1 ? (m == 0 ? (uint32_t)FIELD_MAX : 0u) : (uint32_t)SAT;
    }
    // refine queue of a wavefront: one entry per node with filter survivors = (code, id, survivor mask);
    // (16 wavefronts share what the 128 KB of tables and the tightening's 1-2 KB histogram leave of the 160 KB)
    static constexpr int QE_BYTES = 4 * W + 4 + 4 * MD;
    static constexpr int QCAP = M <= 8 ? 88 : 80;
    static_assert(QCAP >= 64 + 16, "a step pushes up to 64 entries");
};

// a7: the reference's decoder[256] (main:312-325) as byte-permute selectors.
// entry[0]/[1]: v_perm_b32 selectors that move a node's packed changed bytes to
// positions 0..3 / 4..7 of an 8-position group (0x0c = constant zero).
// A compile-time table in global memory (4 KB, lives in the vector L1): the
// scan is bound by LDS cycles, so the decode keeps its table out of the LDS.
struct DecodeTable {
    uint32_t e[256][2];
    constexpr DecodeTable() : e() {
        for (int b = 0; b < 256; ++b) {
            uint32_t sel[2] = {0, 0};
            uint32_t rank = 0;
            for (int m = 0; m < 8; ++m) {
                const bool set = (b >> m) & 1;
                sel[m >> 2] |= (set ? rank : 0x0cu) << (8 * (m & 3));
                rank += set ? 1u : 0u;
            }
            e[b][0] = sel[0];
            e[b][1] = sel[1];
        }
    }
};
__device__ const DecodeTable g_dtab{};

// bit i (0..3) of `nib` -> v_perm_b32 selector byte i: i (take the own byte) where the bit is set,
// 4 + i (take the other operand's byte) elsewhere.  perm(other, own, sel).
__device__ __forceinline__ uint32_t own_sel(uint32_t nib) {
    return 0x07060504u - ((nib * 0x00810204u) & 0x04040404u);  // nib < 16: bit i lands on bit 8 i + 2, no carries
}

template <int M>
struct WaveDecoder {
    static constexpr int W = Cfg<M>::W;
    static constexpr int LEVELS = Cfg<M>::LEVELS;
    uint32_t stk[W];  // ancestor stack (vecs_stack, h:2858-2862) in lanes 0..LEVELS-1
    uint64_t doff;    // offset of the chunk's first changed byte

    __device__ __forceinline__ void begin_segment(const DeviceImage& img, uint32_t seg, int lane) {
        doff = img.seg_delta_off[seg];
#pragma unroll
        for (int w = 0; w < W; ++w) stk[w] = 0;
        if (lane < LEVELS) {
            const uint32_t* ck = reinterpret_cast<const uint32_t*>(img.seg_ckpt) + ((size_t)seg * LEVELS + lane) * W;
#pragma unroll
            for (int w = 0; w < W; ++w) stk[w] = ck[w];
        }
    }

    // The decode of a 64-node chunk runs in three stages so that the scan can keep the stages of three
    // consecutive chunks in flight (each of the first two ends in a global-memory round trip):
    //   load_in   depth nibble and diff mask of the lane's node (coalesced)
    //   load_delta wave scan of popcount(mask) -> the node's changed bytes (3 aligned dwords) + permute selectors
    //   finish    scatter to positions, pointer jumping over the in-chunk ancestor chain, apply to the stack
    struct In {
        uint32_t nb, mk, par;
    };
    struct Ld {
        uint32_t level, mk, par;
        uint32_t w[W / 2][3], sh[W / 2];
        uint2 t[W / 2];
    };
    __device__ __forceinline__ static In load_in(const DeviceImage& img, int64_t node) {
        In r;
        r.nb = img.nib[node >> 1];
        r.mk = M <= 8 ? (uint32_t)img.mask[node] : (uint32_t)reinterpret_cast<const uint16_t*>(img.mask)[node];
        r.par = img.par[node];
        return r;
    }
    // `at`: running offset of the chunk's first changed byte (advanced past the chunk)
    __device__ __forceinline__ static Ld load_delta(const DeviceImage& img, const In& in, int64_t node, uint64_t& at) {
        Ld r;
        r.level = (node & 1) ? (in.nb >> 4) : (in.nb & 15u);
        r.mk = in.mk;
        r.par = in.par;
        const uint32_t pc = __popc(in.mk);
        // wave exclusive scan of pc by bit planes: v_mbcnt, no LDS traffic
        uint32_t excl = 0, total = 0;
#pragma unroll
        for (int b = Cfg<M>::PLANES - 1; b >= 0; --b) {
            const uint64_t plane = __ballot((pc >> b) & 1u);
            excl = mbcnt64(plane, excl << 1);
            total = (total << 1) + (uint32_t)__popcll(plane);
        }
        // changed bytes at byte granularity, one 8-position group at a time: 3 aligned dwords (+ funnel shift later)
#pragma unroll
        for (int h = 0; h < W / 2; ++h) {
            const uint32_t skip = h == 0 ? 0u : (uint32_t)__popc(in.mk & 0xffu);
            const uint8_t* dp = img.delta + at + excl + skip;
            const uintptr_t ua = reinterpret_cast<uintptr_t>(dp);
            const uint32_t* wp = reinterpret_cast<const uint32_t*>(ua & ~(uintptr_t)3);
            r.sh[h] = (uint32_t)(ua & 3);
            r.w[h][0] = wp[0];
            r.w[h][1] = wp[1];
            r.w[h][2] = wp[2];
            r.t[h] = *reinterpret_cast<const uint2*>(g_dtab.e[(in.mk >> (8 * h)) & 0xffu]);
        }
        at += total;
        return r;
    }

    // Decode node `node` (= this lane's node of the chunk) in one go.  `carry`: update the
    // stack for the next chunk of the segment.
    __device__ __forceinline__ void step(const DeviceImage& img, int64_t node, int lane, bool carry, uint32_t (&code)[W]) {
        const In in = load_in(img, node);
        const Ld ld = load_delta(img, in, node, doff);
        uint32_t cl = 0xffu;
        if (carry && lane < LEVELS) cl = img.carry[(size_t)(node >> 6) * LEVELS + lane];
        finish(ld, lane, cl, code);
    }

    // carry_lane (lanes 0..LEVELS-1): lane of the chunk's last node of depth `lane`, 0xFF = none / no carry wanted
    __device__ __forceinline__ void finish(const Ld& ld, int lane, uint32_t carry_lane, uint32_t (&code)[W]) {
        uint32_t mk = ld.mk;
        uint32_t pv[W];
#pragma unroll
        for (int h = 0; h < W / 2; ++h) {  // scatter the packed changed bytes to their positions (a7)
            const uint32_t raw_lo = __builtin_amdgcn_alignbyte(ld.w[h][1], ld.w[h][0], ld.sh[h]);
            const uint32_t raw_hi = __builtin_amdgcn_alignbyte(ld.w[h][2], ld.w[h][1], ld.sh[h]);
            pv[2 * h] = __builtin_amdgcn_perm(raw_hi, raw_lo, ld.t[h].x);
            pv[2 * h + 1] = __builtin_amdgcn_perm(raw_hi, raw_lo, ld.t[h].y);
        }
        // parent = nearest preceding node with depth - 1 (h:2888: stack[depth-1]); which lane that is was
        // resolved when the image was built (DeviceImage::par), as was the stack level the chain ends on
        uint32_t P = ld.par;  // 0xFF: the parent precedes the chunk
        // Pointer jumping: compose patches along the in-chunk ancestor chain.  A patch travels as its
        // W value dwords plus ONE dword (position mask | parent lane): the byte selectors that merge two
        // patches are rebuilt from the position mask (VALU) instead of being carried through the LDS
        // crossbar -- the scan is bound by LDS cycles.
#pragma unroll
        for (int s = 0; s < Cfg<M>::JUMPS; ++s) {
            if (__ballot(P != 0xffu) == 0) break;  // every chain is resolved (wave-uniform)
            const int src = P == 0xffu ? lane : (int)P;
            uint32_t q_pv[W];
#pragma unroll
            for (int w = 0; w < W; ++w) q_pv[w] = bperm(src, pv[w]);
            const uint32_t q_meta = bperm(src, mk | (P << 16));
            if (P != 0xffu) {
#pragma unroll
                for (int w = 0; w < W; ++w) pv[w] = __builtin_amdgcn_perm(q_pv[w], pv[w], own_sel((mk >> (4 * w)) & 15u));
                mk |= q_meta & 0xffffu;
                P = q_meta >> 16;
            }
        }
        // apply to the ancestor that precedes the chunk
#pragma unroll
        for (int w = 0; w < W; ++w)
            code[w] = __builtin_amdgcn_perm(bperm((int)ld.level, stk[w]), pv[w], own_sel((mk >> (4 * w)) & 15u));
        // carry the stack: stack[D] = code of the last node with depth D
        if (__ballot(carry_lane != 0xffu)) {
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const uint32_t nv = bperm(carry_lane == 0xffu ? lane : (int)carry_lane, code[w]);
                if (carry_lane != 0xffu) stk[w] = nv;
            }
        }
    }
};

// Cascade level 0: plain codes of a list of segments.  grid = n_seg, block = 64.
template <int M>
__global__ __launch_bounds__(64) void decode_segments_kernel(const DeviceImage img,
                                                              const uint32_t* __restrict__ seg_list,
                                                              uint32_t* __restrict__ out_id,
                                                              uint32_t* __restrict__ out_code) {
    constexpr int W = Cfg<M>::W;
    const int lane = threadIdx.x;
    const uint32_t seg = seg_list ? seg_list[blockIdx.x] : blockIdx.x;
    const int cps = img.chunks_per_segment;
    WaveDecoder<M> dec;
    if (!img.raw) dec.begin_segment(img, seg, lane);
    for (int c = 0; c < cps; ++c) {
        const int64_t node = ((int64_t)seg * cps + c) * 64 + lane;
        uint32_t code[W];
        if (img.raw) {
#pragma unroll
            for (int w = 0; w < W; ++w) code[w] = reinterpret_cast<const uint32_t*>(img.raw)[(size_t)node * W + w];
        } else {
            dec.step(img, node, lane, c + 1 < cps, code);
        }
        const size_t o = ((size_t)blockIdx.x * cps + c) * 64 + lane;
        out_id[o] = node < img.n_local ? img.id_base + (uint32_t)node : 0xffffffffu;
#pragma unroll
        for (int w = 0; w < W; ++w) out_code[W * o + w] = code[w];
    }
}

// Segments decoded into plain codes, one wavefront per segment: entry j of seg_list (NULL: segment j) goes to
// position j of the scratch, [n_seg * S][M].  grid = ceil(n_seg / 4), block = 256.
template <int M>
__global__ __launch_bounds__(256) void decode_list_kernel(const DeviceImage img, const uint32_t* __restrict__ seg_list,
                                                           int n_seg, const uint8_t* __restrict__ relabel,
                                                           uint32_t* __restrict__ out_code) {
    constexpr int W = Cfg<M>::W;
    // relabel[m][code value] = the label the scratch (and the batch's tables) use; read from global memory (2-4 KB, L1):
    // this kernel must not ask for LDS -- a workgroup that does is only placed where a scan workgroup has retired, and
    // the decode of the next pipelined batch then runs in the scan's tail instead of under it (as lut_build_kernel did)
    __builtin_amdgcn_s_setprio(DPQ_UNDER_SCAN_PRIO);  // (see lut_build_kernel)
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n_seg) return;
    const int64_t seg = seg_list ? (int64_t)seg_list[j] : j;
    const int cps = img.chunks_per_segment;
    WaveDecoder<M> dec;
    dec.begin_segment(img, (uint32_t)seg, lane);
    for (int c = 0; c < cps; ++c) {
        const int64_t node = (seg * cps + c) * 64 + lane;
        const int64_t out = (j * cps + c) * 64 + lane;
        uint32_t code[W];
        dec.step(img, node, lane, c + 1 < cps, code);
        if (relabel) {
#pragma unroll
            for (int w = 0; w < W; ++w) {
                uint32_t r = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) r |= (uint32_t)relabel[(4 * w + b) * 256 + ((code[w] >> (8 * b)) & 0xffu)] << (8 * b);
                code[w] = r;
            }
        }
        if constexpr (W == 2)
            reinterpret_cast<uint2*>(out_code)[out] = make_uint2(code[0], code[1]);
        else
            reinterpret_cast<uint4*>(out_code)[out] = make_uint4(code[0], code[1], code[2], code[3]);
    }
}

// ---------------------------------------------------------------------------
// scan: decode + ADC filter + exact check for QG queries per workgroup
// ---------------------------------------------------------------------------
// exact distance: fp64 sum of the M fp32 entries, rounded to fp32 == the
// reference's incremental fp64 stack (h:2889-2907), see DESIGN.md section 3
template <int M>
__device__ __forceinline__ float exact_dist(const float* __restrict__ T, const uint32_t* __restrict__ c, bool fp32_accum) {
    if (fp32_accum) {  // plain scan (h:2658-2662): `float dist += lut[m][code]`, m ascending
        float fsum = 0.0f;
#pragma unroll
        for (int m = 0; m < M; ++m) fsum = __fadd_rn(fsum, T[m * 256 + ((c[m >> 2] >> (8 * (m & 3))) & 0xffu)]);
        return fsum;
    }
    double dsum = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const uint32_t byte = (c[m >> 2] >> (8 * (m & 3))) & 0xffu;
        dsum = __dadd_rn(dsum, (double)T[m * 256 + byte]);
    }
    return (float)dsum;
}

// ---------------------------------------------------------------------------
// Filter tables of one cascade level, built ONCE per query group (the 16 scan
// workgroups of a group used to rebuild the same 128 KB each): conservative
// lower-bound tables, F queries per dword, in the LDS layout of the scan
// ([g][m][code] x 16 B), written to global memory; a scan workgroup copies them.
// For slot q with threshold tau:
//   entry[m][k] = min(floor((T[m][k] - min_m) * s), SAT) (+ BIAS for m = 0),  s = QT / (tau' - sum_m min_m)
//   tau' = tau * (1 + 2^-20): covers the fp32 rounding of the exact distance and, for the plain
//   scan, the M fp32 roundings of its accumulated distance (<= M * 2^-24 relative)
// floor and min only lower an entry, so sum_m entry <= (d - sum min) * s, which is <= QT for a node
// with exact distance d <= tau: its field sum stays <= QT + BIAS < 2^(EB-1).  The top bit of a field
// is therefore a safe reject flag (no per-query compare in the scan loop), M * SAT + BIAS <= 2^EB - 1
// keeps the fields of a dword from carrying into each other, and what the filter lets through is
// checked exactly before it becomes a candidate.
// Computed as fma(T, s32, -off) in fp32: off >= min * s32 (rounded up), one rounding of the
// result, and s32 carries a (1 - 2^-20) factor, so every entry is <= the exact real value.
// grid = (NG * M, groups), block = 256 threads: thread k builds the 16-byte entry (g, m, k).
// ---------------------------------------------------------------------------
// Scale and offsets of one slot's filter fields (see the comment above): s32 = QT / (tau' - sum of minima),
// the additive term of sub-space m, rounded so that every field stays a lower bound.
struct FilterScale {
    float s32;      // 0: all fields 0, everything passes the filter
    uint32_t bias;  // added to the m = 0 fields
    double B;       // sum of the per-sub-space minima
};

template <int M>
__device__ __forceinline__ FilterScale filter_scale(uint64_t key, const float* __restrict__ lut_min, size_t q) {
    using C = Cfg<M>;
    FilterScale r{0.0f, 0u, 0.0};
    double B = 0.0;
#pragma unroll
    for (int mm = 0; mm < M; ++mm) B += (double)lut_min_of(lut_min, q, M, mm);
    const double taup = (double)__uint_as_float((uint32_t)(key >> 32)) * (1.0 + 0x1p-20);
    const double R = taup - B;
    r.B = B;
    if (key != ~0ull && R > 0.0 && R < 1e300) {  // else: no threshold yet (or degenerate), keep everything
        r.s32 = (float)((double)C::QT / R * (1.0 - 0x1p-20));
        r.bias = C::R == 1 ? (uint32_t)C::BIAS : 0u;  // R = 2: the accumulators carry it
    }
    return r;
}

// entry = fma(T, s32, -off) with off >= min * s32 (rounded up): never above (T - min) * s32.  EB = 8: returns the
// whole additive term of v_cvt_pk_u8_f32's input (bias of m = 0 and half a unit of slack folded in, rounded DOWN).
template <int M>
__device__ __forceinline__ float filter_offset(const FilterScale& fs, float min_m, int m) {
    const float mn = fs.s32 != 0.0f ? min_m : 0.0f;
    const double od = (double)mn * (double)fs.s32;
    float of = (float)od;
    if ((double)of < od) of = __uint_as_float(__float_as_uint(of) + 1u);  // od >= 0: next float up
    const double sd = (m == 0 ? (double)fs.bias : 0.0) - 0.5 - (double)of;
    float sf = (float)sd;
    if ((double)sf > sd) sf = __uint_as_float(__float_as_uint(sf) + (sf > 0.0f ? -1 : 1));  // next float down
    return sf;
}

template <int M>
__device__ __forceinline__ uint32_t filter_field(float tv, float sc, float of, uint32_t bias_m) {
    using C = Cfg<M>;
    // v_cvt_pk_u8_f32: float -> u8 clamped to [0, 255].  Half a unit is taken off first (in `of`), so whichever
    // way the conversion rounds, the byte is <= floor(value): the entry stays a lower bound (negative -> 0;
    // inf/NaN of centroids beyond K -> SAT by the min).
    const float fv = fminf(__fmaf_rn(tv, sc, of), (float)C::SAT + (float)bias_m);
    return __builtin_amdgcn_cvt_pk_u8_f32(fv, 0u, 0u);
}

// One slot's fields of its group's filter tables, written byte by byte (16 bytes apart):
// T = the slot's exact tables (NULL: an unused slot, its entries reject).
template <int M>
__device__ __forceinline__ void write_filter_fields(uint4* qtab, int slot, const float* T, float sc, const float* of_m,
                                                    uint32_t bias, const uint8_t* __restrict__ relabel, int tid, int nthreads) {
    using C = Cfg<M>;
    constexpr int F = C::F, AB = C::AB, R = C::R, QG = C::QG, NG = C::NG, J = C::J;
    const int group = slot / QG, ls = slot % QG;
    const int f = (ls % (J * F)) / J, acc = (ls / (J * F)) * AB + ls % J;  // inverse of Cfg::slot_of
    const int d = acc / R, tb = f * R + acc % R;                           // ... and of Cfg::slot_of_table
    const int g = d >> 2, c = d & 3;
    unsigned char* base = reinterpret_cast<unsigned char*>(qtab + ((size_t)group * NG + g) * M * 256) + c * 4 + tb;
    for (int e = tid; e < M * 256; e += nthreads) {
        const int m = e >> 8;
        const uint32_t v = T ? filter_field<M>(T[e], sc, of_m[m], m == 0 ? bias : 0u) : C::reject_entry(m);
        const int row = relabel ? (m << 8) + relabel[e] : e;  // the label this code value carries in the scratch
        base[(size_t)row * 16] = (unsigned char)v;
    }
}

template <int M>
__global__ __launch_bounds__(256) void quantise_kernel(const ScanArgs a) {
    using C = Cfg<M>;
    constexpr int QG = C::QG, NG = C::NG;
    constexpr int TE = M * 256;
    constexpr int NS = 16;  // slots served by one 16-byte entry: byte tb of dword c
    __shared__ float s_scale[NS], s_off[NS];
    __shared__ int32_t s_row[NS];
    __shared__ uint32_t s_bias[NS];
    const int g = blockIdx.x / M, m = blockIdx.x % M, group = blockIdx.y, k = threadIdx.x;
    if (blockIdx.x == 0 && a.tight_hist)  // this level's tightening histograms of the group's slots (scan_kernel)
        for (int i = k; i < QG * kTightWords; i += 256) a.tight_hist[(size_t)group * QG * kTightWords + i] = 0;
    if (k < NS) {
        const int ls = C::slot_of_table(g, k >> 2, k & 3);
        const int slot = group * QG + ls;
        int qq = a.slot_query ? a.slot_query[slot] : (slot < a.n_queries ? slot : -1);
        FilterScale fs{0.0f, 0u, 0.0};
        float mn_m = 0.0f;
        if (qq >= 0 && a.debug_pass != 1) {
            uint64_t key = ~0ull;
            if (a.debug_pass != 2) key = a.thr_key[slot];
            fs = filter_scale<M>(key, a.lut_min, (size_t)qq);
            mn_m = lut_min_of(a.lut_min, (size_t)qq, M, m);
        } else {
            qq = -1;
        }
        const float s32 = fs.s32;
        const uint32_t bias = fs.bias;
        const float of = filter_offset<M>(fs, mn_m, m);
        s_scale[k] = s32;
        s_off[k] = of;
        s_bias[k] = m == 0 ? bias : 0u;
        s_row[k] = qq >= 0 ? qq * TE + m * 256 : -1;
    }
    __syncthreads();
    uint32_t out[4] = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            const int i = c * 4 + tb;
            const int row = s_row[i];
            const float sc = s_scale[i], of = s_off[i];
            const uint32_t bias = s_bias[i];
            const float tv = row >= 0 ? a.lut32[(size_t)row + k] : 0.0f;
            // v_cvt_pk_u8_f32: float -> u8 clamped to [0, 255], written into byte tb of the dword.  Half a
            // unit is taken off first, so whichever way the conversion rounds, the byte is <= floor(value):
            // the entry stays a lower bound (negative -> 0; inf/NaN of centroids beyond K -> SAT by the min).
            const float fv = fminf(__fmaf_rn(tv, sc, of), (float)C::SAT + (float)bias);  // `of` holds the shift here
            out[c] = row < 0 ? out[c] | (C::reject_entry(m) << (8 * tb))  // unused slot: rejects everything
                             : __builtin_amdgcn_cvt_pk_u8_f32(fv, (uint32_t)tb, out[c]);
        }
    }
    a.qtab[((size_t)group * NG * M + blockIdx.x) * 256 + k] = make_uint4(out[0], out[1], out[2], out[3]);
}

// Read window of the scan's ADC gathers at M = 8 (ds_read_b128 in flight before the first pair is summed; 0 = leave
// the order to the compiler).  See scan_kernel.
#ifndef DPQ_TIGHT_SLEEP
#define DPQ_TIGHT_SLEEP 32  // s_sleep units of 64 cycles between two looks of the helper wavefront (16 / 32 / 64: the same scan, scripts/gpu_tight_sleep.sh)
#endif
#ifndef DPQ_GATHER_PIPE
#define DPQ_GATHER_PIPE 0
#endif

// LDS map of the scan workgroup
template <int M>
struct ScanLds {
    using C = Cfg<M>;
    static constexpr size_t kTables = (size_t)C::NG * M * 256 * 16;                  // filter tables
    static constexpr size_t kThr = kTables;                                           // [QG] u64 threshold keys
    static constexpr size_t kBase = kThr + (size_t)C::QG * 8;                         // [QG] i32 row of the exact tables
    static constexpr size_t kCount = kBase + (size_t)C::QG * 4;                       // [QG] candidates of this workgroup
    static constexpr size_t kChecks = kCount + (size_t)C::QG * 4;                     // [4]: pairs checked exactly, next list entry
    // per-wave ring of nodes with filter survivors awaiting the exact check
    // in-scan tightening: a slot's filter scale and (rounded-down) sum of minima, the additive terms of the NA
    // accumulator dwords (a field grows by the units its slot's cut was lowered by), their version
    static constexpr size_t kScale = kChecks + 16;                                    // [QG] f32
    static constexpr size_t kBdn = kScale + (size_t)C::QG * 4;                        // [QG] f32
    static constexpr size_t kExtra = kBdn + (size_t)C::QG * 4;                        // [NA] u32
    // candidates of this workgroup not yet added to the global histograms: 16-bit counters, two per word
    static constexpr size_t kHist = kExtra + (size_t)C::NA * 4;                       // [QG][kTightBuckets / 2] u32
    static constexpr size_t kQCode = kHist + (size_t)C::QG * kTightBuckets * 2;       // [waves][QCAP][W] dwords
    static constexpr size_t kQId = kQCode + (size_t)kScanWaves * C::QCAP * M;         // [waves][QCAP] u32
    static constexpr size_t kQMask = kQId + (size_t)kScanWaves * C::QCAP * 4;         // [waves][QCAP][MD] dwords
    static constexpr size_t kBytes = kQMask + (size_t)kScanWaves * C::QCAP * 4 * C::MD;
    static_assert(kBytes <= 160 * 1024, "one workgroup per CU: 160 KB of LDS");
};

// Diagnostic build of the scan (STAMPS): per-wavefront s_memtime brackets around the sections of the
// loop, summed into a.stamps[].  Reading the counter drains the wave's LDS/scalar queue, so sections
// do not overlap inside a wavefront as they do in the product kernel (about +10 % cycles); the split
// between sections is what it is for.  Never used by a query call.
enum { kStPrologue = 0, kStSegment, kStDecode, kStGather, kStFold, kStPush, kStRefine, kStTotal, kStSteps, kStRefines,
       kStWaves, kStEntries, kStCount };

// Register budget: the plain-code instantiations at M = 8 stay within 96 VGPRs (second launch bound: five wavefronts per
// SIMD), although a workgroup only ever brings four: the fifth slot's registers are what the NEXT pipelined batch's
// table build and decode run in, under this scan (at 121 VGPRs the pipelined step was 18 us longer).
template <int M, bool PLAIN, bool STAMPS, bool TIGHT>
__global__ __launch_bounds__(kScanThreads, (PLAIN && !STAMPS && M <= 8) ? 5 : 4) void scan_kernel(const ScanArgs a) {
    using C = Cfg<M>;
    constexpr int W = C::W, QG = C::QG, NG = C::NG, NA = C::NA, EB = C::AB, F = C::F, MD = C::MD, QCAP = C::QCAP;
    constexpr int TE = M * 256;  // table entries per query
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* lut = reinterpret_cast<uint4*>(smem);                                        // [NG][M][256] x 16 B = 128 KB
    uint64_t* s_thr = reinterpret_cast<uint64_t*>(smem + ScanLds<M>::kThr);             // [QG]
    int32_t* s_base = reinterpret_cast<int32_t*>(smem + ScanLds<M>::kBase);             // [QG], -1 = unused slot
    // candidates found by this workgroup per query; their keys go straight to the workgroup's own region
    // of the query's candidate buffer (plain stores, no global atomics), the counts in the epilogue
    uint32_t* wg_count = reinterpret_cast<uint32_t*>(smem + ScanLds<M>::kCount);        // [QG]
    uint32_t* wg_checks = reinterpret_cast<uint32_t*>(smem + ScanLds<M>::kChecks);      // [1]
    uint32_t* wg_next = wg_checks + 1;                                                  // [1] next list entry of this workgroup
    float* s_scale = reinterpret_cast<float*>(smem + ScanLds<M>::kScale);               // [QG] 0 = no tightening for the slot
    float* s_bdn = reinterpret_cast<float*>(smem + ScanLds<M>::kBdn);                   // [QG]
    uint32_t* s_extra = reinterpret_cast<uint32_t*>(smem + ScanLds<M>::kExtra);         // [NA]
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(smem + ScanLds<M>::kHist);           // [QG][kTightBuckets / 2]
    // refine ring of this wavefront
    uint32_t* rq_code = reinterpret_cast<uint32_t*>(smem + ScanLds<M>::kQCode) + (size_t)(threadIdx.x >> 6) * QCAP * W;
    uint32_t* rq_id = reinterpret_cast<uint32_t*>(smem + ScanLds<M>::kQId) + (size_t)(threadIdx.x >> 6) * QCAP;
    uint32_t* rq_mask = reinterpret_cast<uint32_t*>(smem + ScanLds<M>::kQMask) + (size_t)(threadIdx.x >> 6) * QCAP * MD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    uint64_t st[kStCount] = {};
    uint64_t t_mark = 0, t_start = 0;
    auto stamp = [&](int section) {
        if constexpr (STAMPS) {
            const uint64_t now = __builtin_amdgcn_s_memtime();
            st[section] += now - t_mark;
            t_mark = now;
        }
    };
    if constexpr (STAMPS) t_start = t_mark = __builtin_amdgcn_s_memtime();
    if (a.wg_times && tid == 0) a.wg_times[2 * (blockIdx.y * gridDim.x + blockIdx.x)] = __builtin_amdgcn_s_memrealtime();

    // Workgroups are dealt to the 8 XCDs round-robin in launch order (observed; speed only).  All
    // workgroups of a query group read the same filter tables and exact tables (QG * M KB), so give each
    // XCD a contiguous run of (group, split) pairs: its L2 then holds 1/8 of the batch's tables.
    int group, split;
    {
        const int total = (int)(gridDim.x * gridDim.y);
        const int linear = (int)(blockIdx.x + gridDim.x * blockIdx.y);
        const int xcd = linear & 7, r = linear >> 3;
        const int per = total >> 3, extra = total & 7;             // XCD x runs per + (x < extra) workgroups
        const int j = xcd * per + min(xcd, extra) + r;             // position in group-major order
        group = j / (int)gridDim.x;
        split = j % (int)gridDim.x;
    }
    const int slot0 = group * QG;

    // ---- prologue: this level's filter tables of the group (quantise_kernel) -> LDS ----
    if (tid < QG) {
        const int slot = slot0 + tid;
        int qq = a.slot_query ? a.slot_query[slot] : (slot < a.n_queries ? slot : -1);
        uint64_t key = ~0ull;
        if (qq >= 0 && a.debug_pass != 1) {
            if (a.debug_pass != 2) key = a.thr_key[slot];
        } else {
            qq = -1;
        }
        s_base[tid] = qq >= 0 ? qq * TE : -1;
        s_thr[tid] = key;
        // in-scan tightening works in the units of the slot's filter tables: the same scale the tables were built with
        float sc = 0.0f, bdn = 0.0f;
        if (TIGHT && qq >= 0 && key != ~0ull) {
            const FilterScale fs = filter_scale<M>(key, a.lut_min, (size_t)qq);
            sc = fs.s32;
            bdn = __double2float_rd(fs.B);
        }
        s_scale[tid] = sc;
        s_bdn[tid] = bdn;
        if (tid < NA) s_extra[tid] = C::R == 1 ? 0u : (uint32_t)C::BIAS * C::LOW;
#pragma unroll
        for (int w = 0; w < kTightBuckets / 2; ++w) s_hist[tid * (kTightBuckets / 2) + w] = 0;
        // a level scanned in several launches (tiles of its segment list) keeps appending to its regions
        wg_count[tid] = a.append ? a.cand_count[(size_t)slot * kRegionStride + 1 + split] : 0u;
        if (tid == 0) {
            *wg_checks = 0;
            *wg_next = 0;
            wg_checks[2] = 0;
        }
    }
    {
        const uint4* src = a.qtab + (size_t)group * (NG * TE);
        constexpr int PER = NG * TE / kScanThreads;  // 8
        static_assert(PER == 8, "eight 16-byte entries per thread");
        // (named values, not an array: next to the other local arrays of this kernel an array here was left in scratch)
        const uint4 v0 = src[tid], v1 = src[tid + kScanThreads], v2 = src[tid + 2 * kScanThreads], v3 = src[tid + 3 * kScanThreads];
        const uint4 v4 = src[tid + 4 * kScanThreads], v5 = src[tid + 5 * kScanThreads], v6 = src[tid + 6 * kScanThreads],
                    v7 = src[tid + 7 * kScanThreads];
        lut[tid] = v0, lut[tid + kScanThreads] = v1, lut[tid + 2 * kScanThreads] = v2, lut[tid + 3 * kScanThreads] = v3;
        lut[tid + 4 * kScanThreads] = v4, lut[tid + 5 * kScanThreads] = v5, lut[tid + 6 * kScanThreads] = v6,
                                 lut[tid + 7 * kScanThreads] = v7;
    }
    __syncthreads();
    stamp(kStPrologue);

    const int cps = a.img.chunks_per_segment;
    WaveDecoder<M> dec;
    // bits of a survivor-mask dword that stand for a slot of this configuration
    uint32_t live = 0;
#pragma unroll
    for (int f = 0; f < F; ++f) live |= ((1u << C::J) - 1u) << (EB * f);

    // Refine ring: one entry per node that has filter survivors.  refine(n) checks the n oldest entries
    // (lane = entry; a lane walks the set bits of its entry's survivor mask, two per round so that their
    // table gathers are in flight together) exactly -- same distance rule and key order as the select
    // kernel; what passes is a candidate.
    int rq_head = 0, rq_n = 0;  // wave-uniform
    uint32_t my_pairs = 0;      // (node, query) pairs this lane checked exactly
    const size_t region0 = (size_t)a.region_off + (size_t)split * a.region_cap;  // this workgroup's region in a slot's buffer
    auto bit_slot = [](int p) { return (p >> 5) * (C::J * F) + ((p & 31) / EB) * C::J + (p % EB); };  // bit -> local slot
    // survivor bits a lane checks per round (their table gathers, M each from L2, are in flight together); 3, 4 and 6
    // per round measured the same step time as 2 at top-100, and 3 / 4 the same or worse at top-1000 (M = 16: 0.755 /
    // 0.816 / 0.795 ms per step; scripts/gpu_refine_bits.sh): the checks are bound by cache lines per gather, not by latency
#ifndef DPQ_REFINE_BITS
#define DPQ_REFINE_BITS 2
#endif
    constexpr int RB = DPQ_REFINE_BITS;
    auto refine = [&](int n) {
        __builtin_amdgcn_wave_barrier();  // ring entries were written by other lanes of this wavefront
        int i = rq_head + lane;
        i = i >= QCAP ? i - QCAP : i;
        uint32_t c[W];
        uint32_t eid = 0;
        uint64_t pend = 0;
        if (lane < n) {
#pragma unroll
            for (int w = 0; w < W; ++w) c[w] = rq_code[i * W + w];
            eid = rq_id[i];
            pend = rq_mask[i * MD];
            if constexpr (MD == 2) pend |= (uint64_t)rq_mask[i * MD + 1] << 32;
        } else {
#pragma unroll
            for (int w = 0; w < W; ++w) c[w] = 0;
        }
        my_pairs += (uint32_t)__popcll(pend);  // statistics: summed over the wavefront once, after the loop
        while (__ballot(pend != 0)) {
            int ls[RB];
            bool has[RB];
            float d[RB];
#pragma unroll
            for (int e = 0; e < RB; ++e) {
                has[e] = pend != 0;
                ls[e] = has[e] ? bit_slot(__ffsll((unsigned long long)pend) - 1) : 0;
                pend &= pend - 1;  // 0 stays 0
            }
            float t_sc[RB], t_bd[RB];  // TIGHT: the slot's filter scale and minima, read while the gathers are in flight
#pragma unroll
            for (int e = 0; e < RB; ++e) {
                t_sc[e] = TIGHT && has[e] ? s_scale[ls[e]] : 0.0f;
                t_bd[e] = TIGHT && has[e] ? s_bdn[ls[e]] : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < RB; ++e) d[e] = has[e] ? exact_dist<M>(a.lut32 + s_base[ls[e]], c, PLAIN && a.fp32_accum != 0) : 0.0f;
#pragma unroll
            for (int e = 0; e < RB; ++e) {
                const uint64_t key = make_key(d[e], eid);
                if (has[e] && key <= s_thr[ls[e]]) {
                    const uint32_t li = atomicAdd(&wg_count[ls[e]], 1u);
                    if (li < (uint32_t)a.region_cap)
                        a.cand_key[(size_t)(slot0 + ls[e]) * a.cand_stride + region0 + li] = key;
                    const float sc = t_sc[e];
                    if (sc != 0.0f) {
                        // by how many steps of XU filter units the slot's cut could be lowered with this node still
                        // below it: u = (d (1 + 2^-20) - minima) * scale, its distance in units of the slot's tables,
                        // in fp32 with 2e-3 units of margin against the node (the roundings are below 1e-4 units)
                        const float u = __fmaf_rn(d[e] * sc, 0x1.000002p-20f, (d[e] - t_bd[e]) * sc);
                        const int steps = min((int)(((float)C::QT - u) * (1.0f / C::XU) - 2e-3f), C::XMAX);
                        // counted in LDS (a global atomic here sits in the wavefront's in-order memory queue in front of the
                        // next round's table gathers); wave 0 moves the counts to the global histograms
                        if (steps >= 1) atomicAdd(&s_hist[ls[e] * (kTightBuckets / 2) + (steps >> 1)], 1u << (16 * (steps & 1)));
                    }
                }
            }
        }
        rq_head += n;
        rq_head = rq_head >= QCAP ? rq_head - QCAP : rq_head;
        rq_n -= n;
        if constexpr (STAMPS) st[kStRefines] += 1;
        __builtin_amdgcn_wave_barrier();
    };

    // ---- in-scan threshold tightening ----
    // The level's thresholds come from the bootstrap (the k-th best of ~3 K sampled nodes: about rank 8 k of the
    // index) or from the previous level; while a launch runs, the candidates all workgroups of a query group have found
    // say more.  Every candidate is counted (refine above) under the number of steps its slot's cut could be lowered
    // with the node still below it; once the nodes that survive a cut of e steps number top_k, the final k-th key cannot
    // lie above that cut.  Wave 0 of every workgroup, every few wave steps: moves the workgroup's counts from LDS to the
    // group's global histograms (atomics), reads those back, lowers threshold keys (exact checks) and raises the additive
    // terms of the filter's accumulator fields (a field's top bit then rejects at QT + 1 - e * XU units instead of
    // QT + 1).  Thresholds only ever tighten and every one of them is an upper bound of the final k-th key, so the result
    // does not depend on when a workgroup looks or on how fresh the counts it sees are: a count that lags still counts
    // real nodes.  The reads are workgroup-scope loads (vector L1 bypassed, this XCD's L2): the workgroups of a query
    // group run on one XCD (the remap above), and agent-scope loads of lines that L2 atomics keep dirty were measured
    // at 3 x the whole scan.
    int my_steps = 0;    // lane = local slot (wave 0): steps its cut has been lowered by
    static_assert(kTightBuckets == 8, "a slot's LDS counters are one 16-byte read, its global row one 8-byte store");
    uint2 row = make_uint2(0, 0);  // helper, lane = slot: this workgroup's row of the slot's counters as last written
    bool frozen = false;
    auto tighten = [&]() {  // the helper wavefront
        const int ls = lane;
        uint32_t raise = 0;
        int field = 0;  // byte (R = 1) / halfword (R = 2) of s_extra that stands for the slot
        if (ls < QG && s_scale[ls] != 0.0f) {
            // This workgroup's running counts (LDS, 16 bits per bucket) go, saturated to bytes, to ITS row of the slot's
            // counters -- a plain 8-byte store, one writer per row -- and the rows of the group's workgroups are read back
            // and summed.  No atomics: agent-scope atomics on these lines from one wavefront per workgroup were measured
            // to double the scan time (they sit in the CU's memory pipeline in front of the exact checks' gathers).
            // layout [group][workgroup (split)][slot of the group] x 8 bytes: a row is what one workgroup writes, and the
            // lanes (slots) of a look read consecutive addresses of it
            uint2* h = reinterpret_cast<uint2*>(a.tight_hist + (size_t)slot0 * kTightWords) + ls;
            asm volatile("" ::: "memory");
            const uint4 now = reinterpret_cast<const uint4*>(s_hist)[ls];
            auto sat = [](uint32_t v) { return min(v & 0xffffu, 255u) | (min(v >> 16, 255u) << 8); };
            // (a 16-bit counter that passed 0x8000 could wrap into its neighbour before the next look: from then on this
            // workgroup's row stays as it is -- counts may lag, they must never run ahead)
            if ((now.x | now.y | now.z | now.w) & 0x80008000u) frozen = true;
            if (!frozen) row = make_uint2(sat(now.x) | sat(now.y) << 16, sat(now.z) | sat(now.w) << 16);
            const uint2 mine = row;
            if (!frozen) h[(size_t)split * QG] = mine;
            uint32_t cnt[kTightBuckets];
#pragma unroll
            for (int e = 0; e < kTightBuckets; ++e) cnt[e] = 0;
            const int n_rows = (int)gridDim.x;  // <= kTightSplits (launch_scan)
            for (int r0 = 0; r0 < n_rows; r0 += 4) {  // four rows per round: two 16-byte loads in flight
                uint32_t w[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint64_t g = r0 + j < n_rows ? __hip_atomic_load(reinterpret_cast<const uint64_t*>(h + (size_t)(r0 + j) * QG),
                                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
                    w[2 * j] = (uint32_t)g, w[2 * j + 1] = (uint32_t)(g >> 32);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t lo = r0 + j == split ? mine.x : w[2 * j], hi = r0 + j == split ? mine.y : w[2 * j + 1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) cnt[e] += (lo >> (8 * e)) & 0xffu, cnt[4 + e] += (hi >> (8 * e)) & 0xffu;
                }
            }
            uint32_t c = 0;
            int best = 0;
#pragma unroll
            for (int e = C::XMAX; e >= 1; --e) {
                c += cnt[e];
                if (best == 0 && c >= (uint32_t)a.tight_k) best = e;
            }
            if (best > my_steps) {
                raise = (uint32_t)((best - my_steps) * C::XU);
                my_steps = best;
                // the cut in distance terms: nodes counted under >= best steps have d <= t2 (see refine)
                const double t2 = ((double)s_bdn[ls] + (double)(C::QT - best * C::XU) / (double)s_scale[ls]) / (1.0 + 0x1p-20);
                const uint64_t key = ((uint64_t)__float_as_uint(__double2float_ru(t2)) << 32) | 0xffffffffull;
                if (key < s_thr[ls]) s_thr[ls] = key;
                // the slot's accumulator field (inverse of Cfg::slot_of)
                const int f = (ls % (C::J * F)) / C::J, acc = (ls / (C::J * F)) * EB + ls % C::J;
                field = C::R == 1 ? acc * 4 + f : acc * 2 + f;
            }
        }
        if constexpr (C::R == 1) {
            // M = 8: the additive term of a slot's field lives in its m = 0 table entries (that is where the bias is).  The
            // raises of this look are gathered into the 16 dwords of s_extra (byte = slot's field), and every entry of the
            // rows (g, m = 0) that have one takes its four dwords' worth in one 16-byte read-add-write: this wavefront is the
            // tables' only writer, byte fields cannot carry (Cfg: M SAT + BIAS + XMAX XU <= 255), and a gather that runs
            // meanwhile sees the old or the new entry -- both are valid cuts.
            if (__ballot(raise != 0)) {
                if (raise != 0) reinterpret_cast<volatile uint8_t*>(s_extra)[field] = (uint8_t)raise;
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const uint4 d = reinterpret_cast<const uint4*>(s_extra)[g];  // broadcast
                    if ((d.x | d.y | d.z | d.w) == 0) continue;  // wave-uniform
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        uint4 v = lut[(g * M) * 256 + lane + 64 * i];
                        v.x += d.x, v.y += d.y, v.z += d.z, v.w += d.w;
                        lut[(g * M) * 256 + lane + 64 * i] = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (lane < NA) s_extra[lane] = 0;
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            if (raise != 0) reinterpret_cast<volatile uint16_t*>(s_extra)[field] = (uint16_t)(C::BIAS + my_steps * C::XU);
        }
    };
    // The LAST wavefront of the workgroup does nothing but this (it draws no segments).  Measured alternatives, scan launch
    // at bootstrap thresholds (no tightening: 0.121 ms): this helper 0.117; the looks inlined into the last wavefront's
    // scan loop every 2 / 4 / 8 steps 0.128-0.130 (96 VGPRs with spills; without the register cap 121 VGPRs, which pushes
    // the next pipelined batch's table build out from under the scan); a second copy of the scan loop for the last
    // wavefront, with the looks, beside a clean copy for the other fifteen 0.122.  What the helper costs is its share of the
    // exact checks' latency hiding: fifteen scanning wavefronts without any tightening take 0.126.
    const bool helper = TIGHT && __builtin_amdgcn_readfirstlane(tid) >= kScanThreads - 64;  // scalar condition
    uint32_t* wg_done = wg_checks + 2;  // scanning wavefronts that have finished

    // list entry s -> workgroup s % splits (a short list still reaches every workgroup); inside the
    // workgroup the wavefronts draw their next entry from an LDS counter, so a wavefront that met
    // segments with many filter survivors does not hold the others back
    int entry_pos = 0;  // list position of the entry next_entry() returned last (the plain-code scratch is laid out by it)
    auto next_entry = [&]() -> int {  // wave-uniform: the next segment of this wavefront, -1 = the list is used up
        int j = 0;
        if (lane == 0) j = (int)atomicAdd(wg_next, 1u);
        j = __builtin_amdgcn_readfirstlane(j);
        const int s = split + (int)gridDim.x * j;
        if (s >= a.n_seg_pass) return -1;
        entry_pos = s;
        return __builtin_amdgcn_readfirstlane(a.seg_list ? (int)a.seg_list[s] : s);
    };
    // Software pipeline over the chunks this wavefront draws: while chunk A is decoded and filtered, the
    // changed bytes of chunk B and the nibbles/masks (and segment header) of chunk C are in flight.  Each
    // stage of the decode ends in a global round trip; with four wavefronts per SIMD they were what a
    // wavefront spent most of its time on (STAMPS build: decode 47 % of the wave time).
    struct Chunk {
        int seg, c;                       // wave-uniform; seg < 0: no chunk
        int pos;                          // PLAIN: list position of the segment
        typename WaveDecoder<M>::In in;   // stage 1
        uint32_t carry_lane;              // stage 1 (lanes 0..LEVELS-1): the chunk's last node of depth `lane`, 0xFF = none
        uint64_t h_doff;                  // stage 1, first chunk of a segment: offset of its first changed byte,
        uint32_t h_stk[W];                //          and the ancestor stack at its first node (lanes 0..LEVELS-1)
        typename WaveDecoder<M>::Ld ld;   // stage 2
        uint32_t raw[W];                  // PLAIN: the code itself (stage 1)
    };
    // Which node of its chunk a lane takes.  The decode ties lane i to node i (parents are reached by lane); plain
    // codes can be dealt freely, and ds_read_b128 serves a wavefront in four fixed 16-lane groups ({0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31}, the same + 32): giving each group 16 CONSECUTIVE nodes of the DFS order -- neighbours share
    // bytes, equal table rows are broadcast -- costs 1.52 instead of 1.60 LDS cycles per group read
    // (scripts/sim_lane_permutation.py).
    int nlane = lane;
    if constexpr (PLAIN) {
        const int h = lane & 31;
        const int g = (h >= 4 && h < 12) || (h >= 16 && h < 20) || h >= 28;
        const int pos = h < 4 ? h : h < 12 ? h - 4 : h < 16 ? h - 8 : h < 20 ? h - 8 : h < 28 ? h - 12 : h - 16;
        nlane = (lane & 32) + 16 * g + pos;
    }
    auto node_of = [&](const Chunk& k) { return ((int64_t)k.seg * cps + k.c) * 64 + nlane; };  // local position
    auto stage1 = [&](Chunk& k) {
        if (k.seg < 0) return;
        const int64_t node = node_of(k);
        if (PLAIN) {  // uncompressed comparator (h:2590-2678): the code is simply there
#pragma unroll
            for (int w = 0; w < W; ++w) {
                // the codes of a plain index lie in node order, the per-batch scratch in the order of the launch's list
                const int64_t at = a.raw_by_pos ? ((int64_t)k.pos * cps + k.c) * 64 + nlane : node;
                k.raw[w] = reinterpret_cast<const uint32_t*>(a.img.raw)[(size_t)at * W + w];
            }
        } else {
            k.in = WaveDecoder<M>::load_in(a.img, node);
            k.carry_lane = 0xffu;
            if (k.c + 1 < cps && lane < C::LEVELS) k.carry_lane = a.img.carry[(size_t)(node >> 6) * C::LEVELS + lane];
            if (k.c == 0) {
                k.h_doff = a.img.seg_delta_off[k.seg];
#pragma unroll
                for (int w = 0; w < W; ++w) k.h_stk[w] = 0;
                if (lane < C::LEVELS) {
                    const uint32_t* ck = reinterpret_cast<const uint32_t*>(a.img.seg_ckpt) + ((size_t)k.seg * C::LEVELS + lane) * W;
#pragma unroll
                    for (int w = 0; w < W; ++w) k.h_stk[w] = ck[w];
                }
            }
        }
    };
    uint64_t at = 0;  // running changed-byte offset of the chunk in stage 2
    auto stage2 = [&](Chunk& k) {
        if (k.seg < 0 || PLAIN) return;
        if (k.c == 0) at = k.h_doff;
        k.ld = WaveDecoder<M>::load_delta(a.img, k.in, node_of(k), at);
    };
    auto successor = [&](const Chunk& k) {
        Chunk n;
        n.seg = k.seg;
        n.pos = k.pos;
        n.c = k.c + 1;
        if (k.seg >= 0 && n.c == cps) {
            n.seg = next_entry();
            n.pos = entry_pos;
            n.c = 0;
        }
        return n;
    };
    Chunk A, B, Cn;
    A.seg = helper ? -1 : next_entry();  // the helper draws no segments: its scan loop below is empty
    A.pos = entry_pos;
    A.c = 0;
    stage1(A);
    B = successor(A);
    stage1(B);
    stage2(A);
    stamp(kStSegment);
    while (A.seg >= 0) {
        // The compiler waits for ALL outstanding loads (vmcnt(0): it cannot count them across the loop's branches) at
        // the first use of any loaded register, and in program order those waits sat right behind the prefetches of
        // the later stages -- every wave step paid one (plain codes: 27 % of the wave time in the STAMPS build) or two
        // (compressed) full memory round trips.  So everything in flight is settled HERE, before this step issues
        // its prefetches: each load has then been in flight for a whole wave step, and the decode and the filter of
        // chunk A below run without a memory wait.
        if constexpr (PLAIN) {
#pragma unroll
            for (int w = 0; w < W; ++w) asm volatile("" ::"v"(A.raw[w]), "v"(B.raw[w]));
        } else {
            asm volatile("" ::"v"(A.ld.level), "v"(A.ld.mk), "v"(A.ld.par), "v"(A.carry_lane), "v"(B.in.nb), "v"(B.in.mk),
                         "v"(B.in.par), "v"(B.carry_lane));
#pragma unroll
            for (int h = 0; h < W / 2; ++h)
                asm volatile("" ::"v"(A.ld.w[h][0]), "v"(A.ld.w[h][1]), "v"(A.ld.w[h][2]), "v"(A.ld.t[h].x), "v"(A.ld.t[h].y));
#pragma unroll
            for (int w = 0; w < W; ++w) asm volatile("" ::"v"(A.h_stk[w]), "v"(B.h_stk[w]));
        }
        Cn = successor(B);
        stage2(B);
        stage1(Cn);
        {
            const int64_t node = node_of(A);
            uint32_t code[W];
            if (PLAIN) {
#pragma unroll
                for (int w = 0; w < W; ++w) code[w] = A.raw[w];
            } else {
                if (A.c == 0) {
#pragma unroll
                    for (int w = 0; w < W; ++w) dec.stk[w] = A.h_stk[w];
                }
                dec.finish(A.ld, lane, A.carry_lane, code);
            }
            if constexpr (STAMPS) {
                st[kStSteps] += 1;
                // the decode's result must exist before its section ends
                asm volatile("" ::"v"(code[0]), "v"(code[W - 1]));
            }
            stamp(kStDecode);

            // ---- ADC lower bound: M LDS gathers per 4 * F queries; the fields of a dword are
            // summed with 3-input integer adds (no carry can cross a field) ----
            uint32_t acc[NA];
            uint32_t fold_h[MD] = {};  // R == 1 with the pipelined gathers: the fold below is done as the groups complete
            if constexpr (C::R == 1 && DPQ_GATHER_PIPE > 0) {
                // The NG * 8 gathers of a wave step as ONE rolling window over the LDS queue: DPQ_GATHER_PIPE reads are
                // issued up front, then every pair of reads that is summed into its accumulators is replaced by the next
                // pair, so the queue never drains inside the step (the compiler's own order issued 16 + 8 + 8 reads with
                // an s_waitcnt lgkmcnt(0) behind each group: three full drains per step at four wavefronts per SIMD).
                // The scheduling fences pin the program order; the waits are the compiler's counted lgkmcnt(N).  The
                // reject-bit fold of a group's four accumulators follows the group's last pair, under the later reads.
                static_assert(NG * 8 == 32 && (DPQ_GATHER_PIPE % 2) == 0 && DPQ_GATHER_PIPE >= 2 && DPQ_GATHER_PIPE <= 14, "window of read pairs");
                uint32_t off[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) off[m] = (code[m >> 2] >> (8 * (m & 3))) & 0xffu;
                uint4 gv[32];
#define DPQ_RD(r) gv[r] = lut[(((r) >> 3) * M + ((r) & 7)) * 256 + off[(r) & 7]]
#pragma unroll
                for (int r = 0; r < DPQ_GATHER_PIPE; r += 2) {
                    DPQ_RD(r);
                    DPQ_RD(r + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int p = 0; p < 16; ++p) {  // pair p = reads 2p, 2p + 1 = sub-spaces 2 (p & 3), + 1 of group p >> 2
                    if (2 * p + DPQ_GATHER_PIPE < 32) {
                        DPQ_RD(2 * p + DPQ_GATHER_PIPE);
                        DPQ_RD(2 * p + DPQ_GATHER_PIPE + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const int g = p >> 2;
                    const uint4 x = gv[2 * p], y = gv[2 * p + 1];
                    if ((p & 3) == 0) {
                        acc[4 * g + 0] = x.x + y.x;
                        acc[4 * g + 1] = x.y + y.y;
                        acc[4 * g + 2] = x.z + y.z;
                        acc[4 * g + 3] = x.w + y.w;
                    } else {
                        acc[4 * g + 0] += x.x + y.x;
                        acc[4 * g + 1] += x.y + y.y;
                        acc[4 * g + 2] += x.z + y.z;
                        acc[4 * g + 3] += x.w + y.w;
                    }
                    if ((p & 3) == 3) {
#pragma unroll
                        for (int j = 4 * g; j < 4 * g + 4; ++j)
                            fold_h[j / EB] = and_or(acc[j], C::LOW << (EB - 1), fold_h[j / EB] >> 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef DPQ_RD
            } else
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if constexpr (C::R == 1) {
                    uint4 v[8];
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const uint32_t byte = (code[m >> 2] >> (8 * (m & 3))) & 0xffu;
                        v[m] = lut[(g * M + m) * 256 + byte];
                    }
                    acc[4 * g + 0] = ((v[0].x + v[1].x + v[2].x) + v[3].x + v[4].x) + (v[5].x + v[6].x + v[7].x);
                    acc[4 * g + 1] = ((v[0].y + v[1].y + v[2].y) + v[3].y + v[4].y) + (v[5].y + v[6].y + v[7].y);
                    acc[4 * g + 2] = ((v[0].z + v[1].z + v[2].z) + v[3].z + v[4].z) + (v[5].z + v[6].z + v[7].z);
                    acc[4 * g + 3] = ((v[0].w + v[1].w + v[2].w) + v[3].w + v[4].w) + (v[5].w + v[6].w + v[7].w);
                } else {
                    // byte sums of four sub-spaces (no carry: 4 * SAT <= 255), widened into 16-bit fields:
                    // even bytes -> accumulator 2 (4 g + c), odd bytes -> the next one; the fields start at BIAS
                    uint32_t lo[4], hi[4];
                    if constexpr (TIGHT) {  // BIAS + what the slots' cuts were lowered by: broadcast reads of 16 words
                        asm volatile("" ::: "memory");  // wave 0 of the workgroup rewrites them: re-read every step
                        const uint4 e0 = reinterpret_cast<const uint4*>(s_extra)[2 * g], e1 = reinterpret_cast<const uint4*>(s_extra)[2 * g + 1];
                        lo[0] = e0.x, hi[0] = e0.y, lo[1] = e0.z, hi[1] = e0.w, lo[2] = e1.x, hi[2] = e1.y, lo[3] = e1.z, hi[3] = e1.w;
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) lo[c] = hi[c] = (uint32_t)C::BIAS * C::LOW;
                    }
#pragma unroll
                    for (int m0 = 0; m0 < M; m0 += C::PRE) {
                        uint32_t p[4] = {0, 0, 0, 0};
#pragma unroll
                        for (int mm = 0; mm < C::PRE; ++mm) {
                            const int m = m0 + mm;
                            const uint32_t byte = (code[m >> 2] >> (8 * (m & 3))) & 0xffu;
                            const uint4 v = lut[(g * M + m) * 256 + byte];
                            p[0] += v.x, p[1] += v.y, p[2] += v.z, p[3] += v.w;
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            lo[c] += __builtin_amdgcn_perm(0u, p[c], 0x0c020c00u);  // bytes 0, 2 -> 16-bit fields
                            hi[c] += __builtin_amdgcn_perm(0u, p[c], 0x0c030c01u);  // bytes 1, 3
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[(4 * g + c) * 2 + 0] = lo[c];
                        acc[(4 * g + c) * 2 + 1] = hi[c];
                    }
                }
            }
            if constexpr (STAMPS) {
#pragma unroll
                for (int i = 0; i < NA; ++i) asm volatile("" ::"v"(acc[i]));
            }
            stamp(kStGather);
            // ---- filter (replaces the heap test h:2909-2914): the top bit of a field rejects.
            // Fold the top bits of the NA accumulators into one pending mask per lane (bit = local slot):
            // shift-and-insert with one mask constant, accumulator j of a mask dword ends at bit j of
            // every field. ----
            const bool valid = node < a.img.n_local;
            uint64_t pend = 0;
#pragma unroll
            for (int h = 0; h < MD; ++h) {
                uint32_t fail = 0;
                if constexpr (C::R == 1 && DPQ_GATHER_PIPE > 0) {
                    fail = fold_h[h];
                } else {
#pragma unroll
                    for (int j = 0; j < C::J; ++j) fail = and_or(acc[EB * h + j], C::LOW << (EB - 1), fail >> 1);
                }
                fail >>= EB - C::J;
                pend |= (uint64_t)(valid ? (~fail & live) : 0u) << (32 * h);
            }
            const uint64_t pushing = __ballot(pend != 0);
            stamp(kStFold);
            if (pushing) {
                // ---- queue the nodes the filter let through for some query: (code, id, survivor mask) ----
                const int cnt = (int)__popcll(pushing);
                while (rq_n + cnt > QCAP) {
                    refine(min(rq_n, 64));
                    stamp(kStRefine);
                }
                if (pend != 0) {
                    int pos = rq_head + rq_n + (int)mbcnt64(pushing, 0);
                    pos = pos >= QCAP ? pos - QCAP : pos;
#pragma unroll
                    for (int w = 0; w < W; ++w) rq_code[pos * W + w] = code[w];
                    rq_id[pos] = a.img.id_base + (uint32_t)node;
                    rq_mask[pos * MD] = (uint32_t)pend;
                    if constexpr (MD == 2) rq_mask[pos * MD + 1] = (uint32_t)(pend >> 32);
                }
                rq_n += cnt;
                stamp(kStPush);
            }
        }
        A = B;
        B = Cn;
    }
    while (rq_n > 0) refine(min(rq_n, 64));
    stamp(kStRefine);
    if constexpr (TIGHT) {
        // the helper keeps looking (about every microsecond) until the other fifteen wavefronts are done
        if (helper) {
            for (;;) {
                tighten();
                if (__builtin_amdgcn_readfirstlane((int)*reinterpret_cast<volatile uint32_t*>(wg_done)) >= kScanWaves - 1) break;
                __builtin_amdgcn_s_sleep(DPQ_TIGHT_SLEEP);
            }
        } else if (lane == 0) {
            atomicAdd(wg_done, 1u);
        }
    }
    if (a.counters || STAMPS) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) my_pairs += (uint32_t)__shfl_xor((int)my_pairs, off, 64);
        if (a.counters && lane == 0) atomicAdd(wg_checks, my_pairs);
        if constexpr (STAMPS) st[kStEntries] += my_pairs;
    }

    // ---- epilogue: this workgroup's candidate counts (a count above region_cap tells the select
    // kernel that candidates were dropped) ----
    __syncthreads();
    if (a.counters && tid < 64) {  // statistics: one pair of global atomics per workgroup
        uint32_t c = 0;
        for (int q = tid; q < QG; q += 64)
            c += wg_count[q] - (a.append ? a.cand_count[(size_t)(slot0 + q) * kRegionStride + 1 + split] : 0u);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += (uint32_t)__shfl_xor((int)c, off, 64);
        if (tid == 0) {
            atomicAdd(&a.counters[0], (unsigned long long)*wg_checks);
            atomicAdd(&a.counters[1], (unsigned long long)c);
        }
    }
    __syncthreads();  // the statistics above read the counts this launch started from
    if (tid < QG) a.cand_count[(size_t)(slot0 + tid) * kRegionStride + 1 + split] = wg_count[tid];
    if (a.wg_times && tid == 0) a.wg_times[2 * (blockIdx.y * gridDim.x + blockIdx.x) + 1] = __builtin_amdgcn_s_memrealtime();
    if constexpr (STAMPS) {
        st[kStTotal] = __builtin_amdgcn_s_memtime() - t_start;
        st[kStWaves] = 1;
        if (a.stamps && lane == 0)
            for (int i = 0; i < kStCount; ++i) atomicAdd(&a.stamps[i], (unsigned long long)st[i]);
    }
}

// ---------------------------------------------------------------------------
// stream: Q queries per pass over the COMPRESSED image, Q = 1, 2, 4, 8 (batches of up to kStreamMaxQ queries -- Q = 1 is
// the reference's own call shape, main:328-339; the idea of one decode serving several queries is the reference's batch
// variant, h:3223-3447).  The 64-query filter machinery does not pay for a handful of queries (a 128 KB table set of
// which a few columns are used); here a wavefront decodes its chunks exactly as the scan does (same WaveDecoder, same
// three chunks in flight) and evaluates every node against the queries' EXACT tables in LDS, interleaved by query
// (T[m][code][Q]: one ds_read_b32 / _b64 / _b128 fetches a sub-space's entry for every query of the pass): an fp32
// sum of the M entries first, the reference's fp64 sum only for nodes within 2^-19 of the query's threshold; what
// passes the threshold key is a candidate, appended to the slot's single region through one global atomic per wave
// step and query that found any (a tight bootstrap threshold leaves a few per thousand nodes).  Q * M KB of LDS:
// this is the mode in which the path is bound by the decode's instruction count and by HBM, not by the LDS array.
// grid = (workgroups, ceil(slots / Q)), block = 256; the segments of the launch's list are dealt round-robin to the
// grid's wavefronts.
// ---------------------------------------------------------------------------
// Block size: the decode is a chain of global and crossbar round trips, so the kernel lives on wavefronts per CU.  Q <= 2:
// 256 threads, 8-16 KB of tables, 32 wavefronts per CU at <= 64 VGPRs.  Q = 4 (32 KB of tables at M = 8): 512 threads share
// one table set; the kernel wants 78 VGPRs: three blocks = 24 wavefronts per CU (squeezed into 64 VGPRs for 32 wavefronts
// it spills: four queries x 125 M codes 1.74 ms against 0.85; 256-thread blocks, 20 wavefronts: 1.00).
template <int Q>
constexpr int stream_threads() { return Q >= 4 ? 512 : 256; }

template <int M, int Q>
#ifndef DPQ_STREAM_Q4_WAVES
#define DPQ_STREAM_Q4_WAVES 6
#endif
__global__ __launch_bounds__(stream_threads<Q>(), M > 8 ? 4 : Q >= 4 ? DPQ_STREAM_Q4_WAVES : 8) void stream_kernel(const ScanArgs a) {
    constexpr int kStreamThreads = stream_threads<Q>();
    using C = Cfg<M>;
    constexpr int W = C::W, TE = M * 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* T = reinterpret_cast<float*>(smem);  // [M][256][Q]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot0 = blockIdx.y * Q;
    uint64_t thr[Q];
    float quick[Q];
    bool any = false;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int slot = slot0 + j;
        const int qq = a.slot_query ? (slot < a.n_queries ? a.slot_query[slot] : -1) : (slot < a.n_queries ? slot : -1);
        any |= qq >= 0;
        // a slot nobody asks for: nothing passes (-inf), its table column is never looked at
        thr[j] = qq >= 0 ? a.thr_key[slot] : 0ull;
        // fp32 sums are within M * 2^-24 (relative) of the exact distance: everything that can pass lies below this
        quick[j] = qq < 0 ? -INFINITY : thr[j] == ~0ull ? INFINITY : __uint_as_float((uint32_t)(thr[j] >> 32)) * (1.0f + 0x1p-19f);
        const float* src = a.lut32 + (size_t)(qq >= 0 ? qq : 0) * TE;
        for (int i = tid; i < TE; i += kStreamThreads) T[i * Q + j] = qq >= 0 ? src[i] : 0.0f;
    }
    if (!any) return;  // block-uniform
    __syncthreads();

    const int cps = a.img.chunks_per_segment;
    const int n_waves = (int)gridDim.x * (kStreamThreads / 64);
    int entry = (int)blockIdx.x * (kStreamThreads / 64) + wave;  // list position of this wavefront's next segment
    auto next_entry = [&]() -> int {
        if (entry >= a.n_seg_pass) return -1;
        const int s = entry;
        entry += n_waves;
        return __builtin_amdgcn_readfirstlane(a.seg_list ? (int)a.seg_list[s] : s);
    };
    WaveDecoder<M> dec;
    struct Chunk {
        int seg, c;
        typename WaveDecoder<M>::In in;
        uint32_t carry_lane;
        uint64_t h_doff;
        uint32_t h_stk[W];
        typename WaveDecoder<M>::Ld ld;
    };
    auto node_of = [&](const Chunk& k) { return ((int64_t)k.seg * cps + k.c) * 64 + lane; };
    auto stage1 = [&](Chunk& k) {
        if (k.seg < 0) return;
        const int64_t node = node_of(k);
        k.in = WaveDecoder<M>::load_in(a.img, node);
        k.carry_lane = 0xffu;
        if (k.c + 1 < cps && lane < C::LEVELS) k.carry_lane = a.img.carry[(size_t)(node >> 6) * C::LEVELS + lane];
        if (k.c == 0) {
            k.h_doff = a.img.seg_delta_off[k.seg];
#pragma unroll
            for (int w = 0; w < W; ++w) k.h_stk[w] = 0;
            if (lane < C::LEVELS) {
                const uint32_t* ck = reinterpret_cast<const uint32_t*>(a.img.seg_ckpt) + ((size_t)k.seg * C::LEVELS + lane) * W;
#pragma unroll
                for (int w = 0; w < W; ++w) k.h_stk[w] = ck[w];
            }
        }
    };
    uint64_t at = 0;
    auto stage2 = [&](Chunk& k) {
        if (k.seg < 0) return;
        if (k.c == 0) at = k.h_doff;
        k.ld = WaveDecoder<M>::load_delta(a.img, k.in, node_of(k), at);
    };
    auto successor = [&](const Chunk& k) {
        Chunk n;
        n.seg = k.seg;
        n.c = k.c + 1;
        if (k.seg >= 0 && n.c == cps) {
            n.seg = next_entry();
            n.c = 0;
        }
        return n;
    };
    Chunk A, B, Cn;
    A.seg = next_entry();
    A.c = 0;
    stage1(A);
    B = successor(A);
    stage1(B);
    stage2(A);
    while (A.seg >= 0) {
        // settle what is in flight before this step issues its prefetches (see scan_kernel)
        asm volatile("" ::"v"(A.ld.level), "v"(A.ld.mk), "v"(A.ld.par), "v"(A.carry_lane), "v"(B.in.nb), "v"(B.in.mk),
                     "v"(B.in.par), "v"(B.carry_lane));
#pragma unroll
        for (int h = 0; h < W / 2; ++h)
            asm volatile("" ::"v"(A.ld.w[h][0]), "v"(A.ld.w[h][1]), "v"(A.ld.w[h][2]), "v"(A.ld.t[h].x), "v"(A.ld.t[h].y));
        Cn = successor(B);
        stage2(B);
        stage1(Cn);
        const int64_t node = node_of(A);
        uint32_t code[W];
        if (A.c == 0) {
#pragma unroll
            for (int w = 0; w < W; ++w) dec.stk[w] = A.h_stk[w];
        }
        dec.finish(A.ld, lane, A.carry_lane, code);
        float d32[Q];
#pragma unroll
        for (int j = 0; j < Q; ++j) d32[j] = 0.0f;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const float* row = T + (size_t)(m * 256 + ((code[m >> 2] >> (8 * (m & 3))) & 0xffu)) * Q;
            if constexpr (Q == 1) {
                d32[0] += row[0];
            } else if constexpr (Q == 2) {
                const float2 v = *reinterpret_cast<const float2*>(row);
                d32[0] += v.x, d32[1] += v.y;
            } else {
#pragma unroll
                for (int j0 = 0; j0 < Q; j0 += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(row + j0);
                    d32[j0] += v.x, d32[j0 + 1] += v.y, d32[j0 + 2] += v.z, d32[j0 + 3] += v.w;
                }
            }
        }
        const bool valid = node < a.img.n_local;
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            bool pass = valid && d32[j] <= quick[j];
            if (__ballot(pass) == 0) continue;  // wave-uniform: nothing of this step can pass for query j
            uint64_t key = 0;
            if (pass) {  // the reference's distance (fp64 sum rounded once) and the whole (distance, id) key
                double dsum = 0.0;
#pragma unroll
                for (int m = 0; m < M; ++m)
                    dsum = __dadd_rn(dsum, (double)T[(size_t)(m * 256 + ((code[m >> 2] >> (8 * (m & 3))) & 0xffu)) * Q + j]);
                key = make_key((float)dsum, a.img.id_base + (uint32_t)node);
                pass = key <= thr[j];
            }
            const uint64_t found = __ballot(pass);
            if (found) {
                uint32_t* count = a.cand_count + (size_t)(slot0 + j) * kRegionStride + 1;  // the slot's single region
                uint64_t* region = a.cand_key + (size_t)(slot0 + j) * a.cand_stride + a.region_off;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(count, (uint32_t)__popcll(found));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                const uint32_t li = base + mbcnt64(found, 0);
                if (pass && li < (uint32_t)a.region_cap) region[li] = key;
            }
        }
        A = B;
        B = Cn;
    }
}

// ---------------------------------------------------------------------------
// strand: the stream pass over the STRAND image (dpq_format.h; M = 8, shards with a bootstrap): Q = 1, 2, 4 queries per
// pass like stream_kernel, but a LANE decodes a run of 64 consecutive nodes one after the other -- the reference's own
// stack machine (h:2876-2905: code = stack[depth - 1] with the masked positions replaced; stack[depth] = code), 64 of
// them side by side, the ancestor stacks in LDS ([level][lane]: conflict-free 8-byte rows) -- instead of a wavefront
// resolving a 64-node chunk cooperatively.  Per 64 decoded nodes that is ~45 instructions of decode instead of ~270 (no
// prefix scan over the masks, no pointer jumping over the LDS crossbar, no carry between chunks): the pass is bound by
// the exact-table gathers in LDS instead of by instruction issue.
// A wavefront takes a strip (4096 nodes) at a time: loads the 64 runs' checkpoints into its stack rows, then per group
// of four steps one coalesced header load (four mask bytes + four depth nibbles: 6 bytes), and the lane's
// changed bytes of the group (<= 32: two 16-byte loads at ITS byte offset inside the group's bytes, which lie lane after
// lane: the wavefront's loads touch a few hundred contiguous bytes), a group ahead.  They are parked in LDS as dword
// ROWS [k][lane] -- lane l's k-th dword in bank l mod 32 whatever k, so the three dwords a node reads around its byte
// pointer never conflict (an unaligned ds_read_b64 at the pointer, tried first, cost 40 % of the pass) -- funnel-shifted
// to the pointer, scattered by the mask's byte-permute selectors (the decoder[256] of main:312-325, here a 2 KB LDS
// table) and merged into the parent code with two v_perm.
// Same distances, keys and candidate handling as stream_kernel.  grid = (workgroups, ceil(slots / Q)), block = 256.
// ---------------------------------------------------------------------------
#ifndef DPQ_ADC_SPLIT
#define DPQ_ADC_SPLIT 4  // sub-spaces summed before a lane looks at its bound (one query per pass)
#endif
constexpr int kStrandThreads = 256;
constexpr int kStrandRows = kPhaseLen * 2 + 2;         // dword rows of a phase in LDS: a lane's <= 32 bytes + the overshoot of a 3-dword read
constexpr int kStrandBuf = kStrandRows * 64 * 4;
constexpr int kStrandLevels = 8;
static_assert(kPhaseLen == 4 && kRunLen == 64, "headers come four to an 8-byte load; a phase is two 16-byte loads per lane at most");

template <int Q>
struct StrandLds {
    static constexpr size_t kT = 0;                                        // [8][256][Q] f32
    static constexpr size_t kDtab = kT + (size_t)Q * 8 * 256 * 4;          // [256] x 8 B selectors
    static constexpr size_t kWave = kDtab + 256 * 8;                       // per wavefront: stack [8][64] x 8 B, then the phase buffer
    // 6.5 KB per wavefront: four blocks = 16 wavefronts per CU at Q = 1.  Squeezed to 5.5 KB (seven stack levels, eight rows
    // with wrap-around reads: exactly 32 KB per block, five blocks per CU) the big level of a 125 M-code call took 337 us
    // instead of 239 -- and 262 with the fifth block kept out by padding; three blocks (padding): 288, two: 352.  Sixteen
    // wavefronts per CU is the optimum.
    static constexpr size_t kPerWave = kStrandLevels * 64 * 8 + kStrandBuf;
    static constexpr size_t kBytes = kWave + (kStrandThreads / 64) * kPerWave;
};

template <int Q>
__global__ __launch_bounds__(kStrandThreads, Q >= 4 ? 2 : Q == 2 ? 3 : 4) void strand_kernel(const ScanArgs a) {
    constexpr int M = 8, TE = M * 256, LEVELS = 8, GROUPS = kRunLen / kPhaseLen;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* T = reinterpret_cast<float*>(smem + StrandLds<Q>::kT);
    uint2* dtab = reinterpret_cast<uint2*>(smem + StrandLds<Q>::kDtab);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint2* stk = reinterpret_cast<uint2*>(smem + StrandLds<Q>::kWave + (size_t)wave * StrandLds<Q>::kPerWave);  // [level][lane]
    uint32_t* drow = reinterpret_cast<uint32_t*>(stk + kStrandLevels * 64);  // [kStrandRows][64] dwords of the current group
    const int slot0 = blockIdx.y * Q;
    uint64_t thr[Q];
    float quick[Q], rest_min[Q][M];  // rest_min: minima of the sub-spaces' tables (wave-uniform)
    bool any = false;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int slot = slot0 + j;
        const int qq = a.slot_query ? (slot < a.n_queries ? a.slot_query[slot] : -1) : (slot < a.n_queries ? slot : -1);
        any |= qq >= 0;
        thr[j] = qq >= 0 ? a.thr_key[slot] : 0ull;
        quick[j] = qq < 0 ? -INFINITY : thr[j] == ~0ull ? INFINITY : __uint_as_float((uint32_t)(thr[j] >> 32)) * (1.0f + 0x1p-19f);
#pragma unroll
        for (int m = 0; m < M; ++m) rest_min[j][m] = qq >= 0 ? lut_min_of(a.lut_min, (size_t)qq, M, m) : 0.0f;
        const float* src = a.lut32 + (size_t)(qq >= 0 ? qq : 0) * TE;
        for (int i = tid; i < TE; i += kStrandThreads) T[i * Q + j] = qq >= 0 ? src[i] : 0.0f;
    }
    if (!any) return;  // block-uniform
    dtab[tid] = *reinterpret_cast<const uint2*>(g_dtab.e[tid]);
    __syncthreads();

    const int n_waves = (int)gridDim.x * (kStrandThreads / 64);
    for (int entry = (int)blockIdx.x * (kStrandThreads / 64) + wave; entry < a.n_seg_pass; entry += n_waves) {
        const int64_t sid = __builtin_amdgcn_readfirstlane(a.seg_list ? (int)a.seg_list[entry] : entry);
        // the runs' ancestor stacks
#pragma unroll
        for (int lv = 0; lv < kStrandLevels; ++lv) {
            const uint64_t c = a.img.st_ckpt[(sid * LEVELS + lv) * 64 + lane];
            stk[lv * 64 + lane] = make_uint2((uint32_t)c, (uint32_t)(c >> 32));
        }
        const uint32_t* pbase = a.img.st_pbase + sid * GROUPS;
        // Two-deep prefetch: group g + 2's header word and group g + 1's changed bytes are fetched while group g is decoded.
        // Where a lane's bytes start inside a group's bytes (which lie lane after lane) is the wave prefix sum of the
        // lanes' byte counts, i.e. of the popcounts of the header's four masks: computed (bit planes + v_mbcnt), not
        // loaded -- a loaded offset put two dependent HBM round trips into every group.
        struct Hdr {
            uint32_t masks, depths;  // the lane's next four nodes: a mask byte and a depth nibble each
        };
        auto load_hdr = [&](int g) -> Hdr {
            if (g >= GROUPS) return Hdr{0u, 0u};
            return Hdr{a.img.st_mask[(sid * GROUPS + g) * 64 + lane], (uint32_t)a.img.st_depth[(sid * GROUPS + g) * 64 + lane]};
        };
        struct Bytes {
            uint4 b0, b1;
        };
        auto load_bytes = [&](int g, const Hdr& h) {
            Bytes r;
            const uint32_t mine = (uint32_t)__popc(h.masks);  // <= 32
            uint32_t off = 0;
#pragma unroll
            for (int bit = 5; bit >= 0; --bit) off = mbcnt64(__ballot((mine >> bit) & 1u), off << 1);
            const unsigned char* src = a.img.st_delta + (size_t)pbase[g] * 16 + off;  // byte address: unaligned 16-byte loads
            __builtin_memcpy(&r.b0, src, 16);
            __builtin_memcpy(&r.b1, src + 16, 16);
            return r;
        };
        Hdr hdr = load_hdr(0), hdr_next = load_hdr(1);
        Bytes cur = load_bytes(0, hdr);
        for (int g = 0; g < GROUPS; ++g) {
            // stage the group's bytes (the previous group's reads of the buffer were issued before: LDS keeps a
            // wavefront's operations in order)
            __builtin_amdgcn_wave_barrier();
            {
                const uint32_t w[8] = {cur.b0.x, cur.b0.y, cur.b0.z, cur.b0.w, cur.b1.x, cur.b1.y, cur.b1.z, cur.b1.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) drow[k * 64 + lane] = w[k];
            }
            __builtin_amdgcn_wave_barrier();
            const Hdr hdr_after = load_hdr(g + 2);
            if (g + 1 < GROUPS) cur = load_bytes(g + 1, hdr_next);
            uint32_t ptr = 0;  // bytes of this lane's group consumed so far
            // The group's four steps in two sweeps: decode + exact-table sums first, straight-line (a step's gathers run
            // under the next step's decode: a branch after every step kept them apart), then what may pass.
            uint32_t codes[kPhaseLen][2];
            float d32[kPhaseLen][Q];
            constexpr int H = Q == 1 ? DPQ_ADC_SPLIT : M;  // several queries: a lane is rarely out of every race -- all at once
#pragma unroll
            for (int st = 0; st < kPhaseLen; ++st) {
                const uint32_t mask = (hdr.masks >> (8 * st)) & 0xffu, depth = (hdr.depths >> (4 * st)) & 0xfu;
                const uint2 parent = stk[(depth > 0 ? depth - 1 : 0) * 64 + lane];
                const uint32_t* at = drow + (ptr >> 2) * 64 + lane;
                const uint32_t w0 = at[0], w1 = at[64], w2 = at[128];
                uint2 raw;
                raw.x = __builtin_amdgcn_alignbyte(w1, w0, ptr & 3u);
                raw.y = __builtin_amdgcn_alignbyte(w2, w1, ptr & 3u);
                const uint2 sel = dtab[mask];
                ptr += __popc(mask);
                const uint32_t pv0 = __builtin_amdgcn_perm(raw.y, raw.x, sel.x), pv1 = __builtin_amdgcn_perm(raw.y, raw.x, sel.y);
                uint32_t* code = codes[st];
                code[0] = __builtin_amdgcn_perm(parent.x, pv0, own_sel(mask & 15u));
                code[1] = __builtin_amdgcn_perm(parent.y, pv1, own_sel(mask >> 4));
                stk[depth * 64 + lane] = make_uint2(code[0], code[1]);
                // ADC against the queries' exact tables, in two halves: after sub-spaces 0..3 a lane whose partial sum plus
                // the minima of the other four tables already exceeds every query's bound is done (fp32 addition is
                // monotone: the same chain with the real entries cannot come out lower) -- the second half's gathers run
                // under the execution mask of the lanes still in the race, and a gather of a few lanes has no bank
                // conflicts to replay (on random codes the 32-lane gathers of all lanes cost 3.5 x their two cycles: half
                // of this kernel's LDS time).
                float p[Q];
#pragma unroll
                for (int j = 0; j < Q; ++j) p[j] = 0.0f;
                auto gather = [&](int m) {
                    const float* row = T + (size_t)(m * 256 + ((code[m >> 2] >> (8 * (m & 3))) & 0xffu)) * Q;
                    if constexpr (Q == 1) {
                        p[0] += row[0];
                    } else if constexpr (Q == 2) {
                        const float2 v = *reinterpret_cast<const float2*>(row);
                        p[0] += v.x, p[1] += v.y;
                    } else {
                        const float4 v = *reinterpret_cast<const float4*>(row);
                        p[0] += v.x, p[1] += v.y, p[2] += v.z, p[3] += v.w;
                    }
                };
#pragma unroll
                for (int m = 0; m < H; ++m) gather(m);
                bool may = H == M;
                if constexpr (H < M) {
                    float b = p[0];
#pragma unroll
                    for (int m = H; m < M; ++m) b += rest_min[0][m];
                    may = b <= quick[0];
                }
#pragma unroll
                for (int j = 0; j < Q; ++j) d32[st][j] = H == M ? p[j] : INFINITY;
                if constexpr (H < M) {
                    // (one look per step: a single look per group of four, summing the rest for all four nodes of a lane
                    // that has any in the race, was measured slower: 258 against 239 us on the big level)
                    if (may) {
#pragma unroll
                        for (int m = H; m < M; ++m) gather(m);
                        d32[st][0] = p[0];
                    }
                }
            }
            asm volatile("" ::: "memory");  // the rare exact sums below re-read their entries (else all 32 stay in registers)
#pragma unroll
            for (int st = 0; st < kPhaseLen; ++st) {
                const uint32_t* code = codes[st];
                const int64_t node = sid * kStripNodes + lane * kRunLen + g * kPhaseLen + st;
                const bool valid = node < a.img.n_local;
#pragma unroll
                for (int j = 0; j < Q; ++j) {
                    bool pass = valid && d32[st][j] <= quick[j];
                    if (__ballot(pass) == 0) continue;  // wave-uniform
                    uint64_t key = 0;
                    if (pass) {  // the reference's distance (fp64 sum rounded once) and the whole (distance, id) key
                        double dsum = 0.0;
#pragma unroll
                        for (int m = 0; m < M; ++m)
                            dsum = __dadd_rn(dsum, (double)T[(size_t)(m * 256 + ((code[m >> 2] >> (8 * (m & 3))) & 0xffu)) * Q + j]);
                        key = make_key((float)dsum, a.img.id_base + (uint32_t)node);
                        pass = key <= thr[j];
                    }
                    const uint64_t found = __ballot(pass);
                    if (found) {
                        uint32_t* count = a.cand_count + (size_t)(slot0 + j) * kRegionStride + 1;
                        uint64_t* region = a.cand_key + (size_t)(slot0 + j) * a.cand_stride + a.region_off;
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(count, (uint32_t)__popcll(found));
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        const uint32_t li = base + mbcnt64(found, 0);
                        if (pass && li < (uint32_t)a.region_cap) region[li] = key;
                    }
                }
            }
            hdr = hdr_next;
            hdr_next = hdr_after;
        }
    }
}

// ---------------------------------------------------------------------------
// strand1: ONE query per pass over the strand image -- the reference's own call shape (main:328-339: one query per
// call) on a big shard, the regime in which the path has to be bound by HBM.  strand_kernel<1> is bound by the LDS
// array instead: the eight exact-table gathers of a node (ds_read_b32 at random rows of an 8 KB table) replay 3.5-fold
// bank conflicts on random codes (38 % of its LDS cycles), and a node's changed bytes make a round trip through
// dword rows in LDS.  This kernel removes both:
//  * ADC (h:2896-2905) in fixed point first: the scan's conservative 8-bit lower bound,
//    entry[m][c] = min(floor((T[m][c] - min_m) s), 255), s = QT / (tau' - sum of minima) -- floor and min only lower an
//    entry, so a node within the threshold has sum <= QT; what the bound lets through (a few per million nodes) is
//    summed exactly (fp64 sum of the fp32 entries, the reference's distance) and compared as a whole (distance, id) key.
//    The table holds the eight entries of a code value c as TWO words (sub-spaces 0..3, 4..7), each stored once per LDS
//    bank: word (c, half, lane mod 32) at c << 8 | half << 7 | (lane mod 32) << 2, 64 KB.  A lane's ds_read_u8 at
//    (c << 8 | (lane mod 32) << 2) + offset:(half << 7 | m & 3) hits its own bank whatever c is (a 4-byte-class DS access
//    serves 32 lanes per pass: SQ_LDS_BANK_CONFLICT 14 cycles per launch); the address is one v_perm_b32 (the code byte
//    lands on bits 8..15), the eight bytes are added with v_add3_u32.  A lane owns its accumulator (no saturation tricks,
//    QT = 250 units instead of the batched scan's 64).  (DPQ_S1_U8 = 0 keeps the round's first shape: 8-byte rows per
//    bank pair read whole with ds_read_b64, the wanted byte added with v_dot4 against a one-hot constant: 2 VALU
//    instructions per node more, 8 VGPRs more, 2 % slower.)
//  * the changed bytes never touch LDS: a lane loads the <= 8 bytes of each of its next four nodes with FOUR unaligned
//    8-byte global loads at exactly the node's byte offset (the wave prefix sum of the masks' popcounts gives a lane's
//    offset in the phase, the popcounts of its own earlier masks the rest), a phase ahead.  No row parking, no
//    funnel shifts, no 6.5 KB of rows per wavefront.
//  * the byte-permute selectors (the decoder[256] of main:312-325) come from a 16-entry NIBBLE table, also stored once
//    per bank (2 KB): low half = perm(parent.x, raw.x, sel[mask & 15]), high half = perm(parent.y, raw >> 8 popc(low), ...).
// What stays in LDS is the reference's stack machine itself (h:2888-2905): the ancestor stacks [level][lane] x 8 B, one
// conflict-free read (the parent) and one write per node.  Per 64 nodes: 13 LDS instructions, ~28 LDS issue cycles
// (2 + 4 + 6 + 16) against ~84.
// One workgroup of 16 wavefronts per CU (138 KB of LDS: the bound rows, the selectors, 4 KB of stack per wavefront, the
// exact table for the rare exact sums).
// In-kernel threshold tightening: one pass over the whole shard at the bootstrap's threshold would admit every node
// below it (rank ~30 K on 125 M codes).  Candidates are counted in a global histogram by the bound-table unit their
// distance falls under; a wavefront looks at it when it starts a strip (and a few times inside its first): once top_k
// candidates lie under a cut of e units the final k-th key cannot lie above it -- the wavefront lowers ITS cut (an SGPR)
// and its threshold key.  Thresholds only tighten and every one is an upper bound of the final k-th key, so the result
// does not depend on when a wavefront looks or how fresh the counts are.
// grid = (workgroups <= 256, slots), block = 1024; a.seg_list / a.n_seg_pass name strips.
// ---------------------------------------------------------------------------
#ifndef DPQ_S1_QT
#define DPQ_S1_QT 250  // units of the bound table that span (tau' - sum of minima); entries saturate at 255
#endif
#ifndef DPQ_S1_SKIP
#define DPQ_S1_SKIP 0  // (timing experiments only, wrong results: 1 no ADC, 2 no stack / selector traffic, 4 no changed-byte loads, 8 one of the four, 16 the four from a cache-resident 64 KB)
#endif
#ifndef DPQ_S1_U8
#define DPQ_S1_U8 1  // bound rows as two words per code value and bank, entries read with ds_read_u8 and added; 0: round 4's first shape, 8-byte rows + v_dot4
#endif
#ifndef DPQ_S1_PAIR
#define DPQ_S1_PAIR 2  // two changed-byte loads per phase serve the four nodes where a pair's bytes fit one window: 1 = 8-byte windows, 2 = 12, 3 = 16 (always); 0: four loads
#endif
#ifndef DPQ_S1_DEPTH
#define DPQ_S1_DEPTH 1  // phases the changed bytes are fetched ahead of their decode, 1 .. 3 (headers: four ahead; 2: 215 against 217 us)
#endif
#ifndef DPQ_S1_THREADS
#define DPQ_S1_THREADS 1024
#endif
constexpr int kS1Threads = DPQ_S1_THREADS, kS1Waves = kS1Threads / 64;
constexpr int kS1Buckets = 64;  // histogram words: bucket b = candidates under a cut of 4 b + 3 units
// The global histogram (a.tight_hist, zero at launch) is kept in kS1Replicas copies 2 KB apart, a workgroup adds to copy
// blockIdx % 8 and a look sums the copies: the few thousand adds of a launch's first microseconds otherwise queue up in
// ONE memory channel, and every wavefront's in-order loads wait for the one that goes there.
constexpr int kS1Replicas = 8, kS1ReplicaWords = 512;
static_assert((DPQ_S1_QT >> 2) < kS1Buckets && DPQ_S1_QT <= 253, "a lane per bucket; 8 entries of 255 must reject");

struct S1Lds {
    static constexpr size_t kTq = 0;                                          // [256][2][32] x 4 B bound words ([256][32] x 8 B rows without DPQ_S1_U8)
    static constexpr size_t kSel = kTq + 256 * 32 * 8;                        // [16][32] x 4 B nibble selectors
    static constexpr size_t kStack = kSel + 16 * 32 * 4;                      // [waves][8][64] x 8 B ancestor stacks
    static constexpr size_t kT32 = kStack + (size_t)kS1Waves * 8 * 64 * 8;    // [8][256] f32 exact table
    // the workgroup's shared words: u32 cut, (pad), u64 threshold key, u32 candidates appended, u32 wavefronts done, u32 next strip, (pad),
    // then [64] u32 candidate counts not yet moved to the global histogram
    static constexpr size_t kShare = kT32 + 8 * 256 * 4;
    static constexpr size_t kHist = kShare + 48;
    static constexpr size_t kBytes = kHist + 64 * 4;
};
static_assert(S1Lds::kBytes <= 160 * 1024 && S1Lds::kTq == 0, "one workgroup per CU; the bound rows sit at LDS offset 0");

// LDS accesses by BYTE OFFSET (this kernel has no static LDS: its dynamic LDS starts at offset 0, checked in the
// prologue): the bound rows' addresses are built with v_perm_b32 and must reach ds_read_b64 as they are
#define DPQ_LDS __attribute__((address_space(3)))
__device__ __forceinline__ uint2 lds_ld64(uint32_t off) {
    const uint64_t v = *(const DPQ_LDS uint64_t*)(uintptr_t)off;
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ uint32_t lds_ld32(uint32_t off) { return *(const DPQ_LDS uint32_t*)(uintptr_t)off; }
__device__ __forceinline__ void lds_st64(uint32_t off, uint2 v) { *(DPQ_LDS uint64_t*)(uintptr_t)off = ((uint64_t)v.y << 32) | v.x; }
// a whole 8-byte row although one half is used: a ds_read_b64 of 32 lanes covers all 64 banks with (lane mod 32) * 8;
// narrowed to ds_read_b32 the same addresses would meet two to a bank
__device__ __forceinline__ uint32_t lds_ld8(uint32_t off) { return *(const DPQ_LDS uint8_t*)(uintptr_t)off; }
__device__ __forceinline__ uint2 lds_ld64_whole(uint32_t off) {
    const uint64_t v = *(const volatile DPQ_LDS uint64_t*)(uintptr_t)off;
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}

// v_perm_b32 selector of a mask nibble: byte i = rank of bit i among the set bits (take the i-th changed byte) where the
// bit is set, 4 + i (keep the parent's byte) elsewhere.  perm(parent, raw, sel).
__device__ __forceinline__ uint32_t s1_nibble_sel(uint32_t nib) {
    uint32_t sel = 0, rank = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool set = (nib >> i) & 1u;
        sel |= (set ? rank : 4u + i) << (8 * i);
        rank += set ? 1u : 0u;
    }
    return sel;
}

// inclusive prefix sum over the 64 lanes with DPP row shifts and row broadcasts (gfx9: the sequence LLVM's own wave
// scan uses): 12 VALU operations, no LDS, no ballots
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}

__global__ __launch_bounds__(kS1Threads) void strand1_kernel(const ScanArgs a) {
    constexpr int M = 8, TE = M * 256, GROUPS = kRunLen / kPhaseLen, D = DPQ_S1_DEPTH, QT = DPQ_S1_QT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = blockIdx.y;
    const int qq = a.slot_query ? (slot < a.n_queries ? a.slot_query[slot] : -1) : (slot < a.n_queries ? slot : -1);
    if (qq < 0) return;  // block-uniform: a slot nobody asks for
    if ((uint32_t)(uintptr_t)(DPQ_LDS unsigned char*)smem != 0u) __builtin_trap();  // (see lds_ld64: offsets are addresses)
    uint64_t thr = a.thr_key[slot];  // wave-uniform; only ever lowered
    // developer diagnostics (a.stamps NULL in every query call): per wavefront 16 marks on the 100 MHz clock -- start,
    // end of the prologue, end of each of its first strips
    unsigned long long* const stp = a.stamps ? a.stamps + ((size_t)blockIdx.x * kS1Waves + wave) * 16 : nullptr;
    int n_stamp = 0;
    unsigned long long slow_ticks = 0, slow_entries = 0;  // (diagnostics: time in and entries into the exact-check path)
    auto stamp = [&]() {
        if (stp && n_stamp < 14) {
            // low 40 bits: the 100 MHz clock; high 24 bits: the shader clock / 1024 (clock rate between two marks)
            const unsigned long long t = (__builtin_amdgcn_s_memrealtime() & 0xffffffffffull) | ((__builtin_amdgcn_s_memtime() >> 10) << 40);
            if (lane == 0) stp[n_stamp] = t;
            ++n_stamp;
        }
    };
    stamp();
    // scale of the bound table (filter_scale with this kernel's QT): s32 = QT / (tau' - B) (1 - 2^-20), tau' = tau (1 + 2^-20)
    FilterScale fs{0.0f, 0u, 0.0};
    float min_m[M];
    {
        double B = 0.0;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            min_m[m] = lut_min_of(a.lut_min, (size_t)qq, M, m);
            B += (double)min_m[m];
        }
        const double taup = (double)__uint_as_float((uint32_t)(thr >> 32)) * (1.0 + 0x1p-20);
        const double R = taup - B;
        fs.B = B;
        if (thr != ~0ull && R > 0.0 && R < 1e300) fs.s32 = (float)((double)QT / R * (1.0 - 0x1p-20));  // else: everything passes the bound
    }
    const float sc = fs.s32, bdn = __double2float_rd(fs.B);
    // developer experiments (a.debug_pass >= 16, never in a query call): bit 0 nothing is checked exactly, bit 1 no looks
    // inside the first strip, bit 2 no tightening
    const int dbg = a.debug_pass >= 16 ? a.debug_pass - 16 : 0;
    uint32_t* hist = a.tight_hist && sc != 0.0f && !(dbg & 4) ? a.tight_hist + (size_t)slot * (kS1Replicas * kS1ReplicaWords) : nullptr;
    int cut = QT;  // wave-uniform: a node passes the bound iff its sum <= cut + 1

    // ---- prologue: exact table, selectors, bound rows ----
    float* T32 = reinterpret_cast<float*>(smem + S1Lds::kT32);
    {
        const float* src = a.lut32 + (size_t)qq * TE;
        for (int i = tid; i < TE; i += kS1Threads) T32[i] = src[i];
        if (tid < 16 * 32) reinterpret_cast<uint32_t*>(smem + S1Lds::kSel)[tid] = s1_nibble_sel((uint32_t)tid >> 5);
        __syncthreads();
        uint2* qtmp = reinterpret_cast<uint2*>(smem + S1Lds::kStack);  // the stacks' space, until the loop starts
        if (tid < 256) {
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                // as the scan's filter_field: half a unit off first, so whichever way v_cvt_pk_u8_f32 rounds the byte is
                // <= floor((T - min) s): the entry stays a lower bound; it clamps to [0, 255] (inf of centroids beyond K: 255)
                const float of = filter_offset<M>(fs, min_m[m], 1);
                const float fv = fminf(__fmaf_rn(T32[m * 256 + tid], sc, of), 255.0f);
                if (m < 4)
                    lo = __builtin_amdgcn_cvt_pk_u8_f32(fv, (uint32_t)m, lo);
                else
                    hi = __builtin_amdgcn_cvt_pk_u8_f32(fv, (uint32_t)(m - 4), hi);
            }
            qtmp[tid] = make_uint2(lo, hi);
        }
        __syncthreads();
        uint2* Tq = reinterpret_cast<uint2*>(smem + S1Lds::kTq);
        if constexpr (DPQ_S1_U8) {
            // word (c, half, bank) at c << 8 | half << 7 | bank << 2: sub-spaces 4 half .. 4 half + 3 of code value c
            uint32_t* Tw = reinterpret_cast<uint32_t*>(smem + S1Lds::kTq);
            for (int w = tid; w < 256 * 64; w += kS1Threads) {
                const uint2 row = qtmp[w >> 6];
                Tw[w] = ((w >> 5) & 1) ? row.y : row.x;
            }
        } else {
            for (int w = tid; w < 256 * 32; w += kS1Threads) Tq[w] = qtmp[w >> 5];
        }
        if (tid == 0) {
            *reinterpret_cast<uint32_t*>(smem + S1Lds::kShare) = (uint32_t)QT;
            *reinterpret_cast<unsigned long long*>(smem + S1Lds::kShare + 8) = thr;
            *reinterpret_cast<uint4*>(smem + S1Lds::kShare + 16) = make_uint4(0u, 0u, 0u, 0u);  // candidates, wavefronts done, next strip
            *reinterpret_cast<uint4*>(smem + S1Lds::kShare + 32) = make_uint4(0u, 0u, 0u, 0u);  // statistics
        }
        if (tid < 64) reinterpret_cast<uint32_t*>(smem + S1Lds::kHist)[tid] = 0u;
        __syncthreads();
    }

    stamp();
    // LDS byte offsets a lane uses all along
    const uint32_t lane8 = (uint32_t)(lane & 31) * (DPQ_S1_U8 ? 4u : 8u);                          // its copy of a bound row
    const uint32_t lane_sel = (uint32_t)S1Lds::kSel + (uint32_t)(lane & 31) * 4u;                  // ... of a selector
    // row -1 of its stack column: depth 0 (the root, mask 0xFF: every byte replaced) reads a parent that does not matter
    // from the 512 bytes in front of the stack -- the previous wavefront's row 7 or the end of the selectors: valid LDS
    const uint32_t lane_stk = (uint32_t)S1Lds::kStack + (uint32_t)wave * (8 * 64 * 8) + (uint32_t)lane * 8u - 512u;

    // ---- in-kernel tightening: looks at the candidate histogram, the workgroup's cut in LDS ----
    // bucket b counts candidates whose distance lies under a cut of 4 b + 3 units (see the exact check below); the first b
    // whose running count reaches top_k gives a valid cut: the k-th best key seen so far is <= the key of that cut.
    // A wavefront looks when it starts a strip (its pipeline is empty there) and once inside its first strip, the
    // workgroup's wavefronts one phase after the other; what it finds goes to the workgroup's shared (cut, threshold key)
    // in LDS with atomic minima, and every wavefront reads those once per phase.
    DPQ_LDS uint32_t* const sh_cut = (DPQ_LDS uint32_t*)(uintptr_t)S1Lds::kShare;
    DPQ_LDS unsigned long long* const sh_thr = (DPQ_LDS unsigned long long*)(uintptr_t)(S1Lds::kShare + 8);
    DPQ_LDS uint32_t* const sh_count = (DPQ_LDS uint32_t*)(uintptr_t)(S1Lds::kShare + 16);
    DPQ_LDS uint32_t* const sh_done = (DPQ_LDS uint32_t*)(uintptr_t)(S1Lds::kShare + 20);
    DPQ_LDS uint32_t* const sh_next = (DPQ_LDS uint32_t*)(uintptr_t)(S1Lds::kShare + 24);
    DPQ_LDS uint32_t* const sh_stat = (DPQ_LDS uint32_t*)(uintptr_t)(S1Lds::kShare + 32);  // [2] statistics
    DPQ_LDS uint32_t* const sh_hist = (DPQ_LDS uint32_t*)(uintptr_t)S1Lds::kHist;
    auto look = [&]() {
        // the workgroup's counts since its last look go to the global histogram first: ONE add per bucket that has any.
        // (Candidates never touch global counters themselves: a few thousand atomics on a handful of words were measured
        // to cost every wavefront of the chip 70 us -- the words' memory channel queues up, and each wavefront's in-order
        // loads wait for the one that goes there.)
        const uint32_t mine = __hip_atomic_exchange(sh_hist + lane, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (mine) atomicAdd(hist + (blockIdx.x % kS1Replicas) * kS1ReplicaWords + lane, mine);
        uint32_t v = 0;
#pragma unroll
        for (int r = 0; r < kS1Replicas; ++r) v += __hip_atomic_load(hist + r * kS1ReplicaWords + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t incl = v;  // inclusive prefix over the lanes (= buckets)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
            if (lane >= o) incl += up;
        }
        const uint64_t ge = __ballot(incl >= (uint32_t)a.tight_k);
        if (ge == 0) return;
        const int e = 4 * (__ffsll((unsigned long long)ge) - 1) + 3;
        if (e < cut) {
            // the cut in distance terms: a candidate counted under <= e units has d <= t2 (exact check below)
            const double t2 = ((double)bdn + (double)e / (double)sc) / (1.0 + 0x1p-20);
            const uint64_t key = ((uint64_t)__float_as_uint(__double2float_ru(t2)) << 32) | 0xffffffffull;
            if (lane == 0) {
                __hip_atomic_fetch_min(sh_cut, (uint32_t)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_min(sh_thr, (unsigned long long)key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };
    auto refresh = [&]() {  // the workgroup's cut and threshold key: both only ever fall, each valid on its own
        // (LDS loads by address space: through a generic pointer these were flat loads behind a full s_waitcnt vmcnt(0))
        const uint32_t c = __hip_atomic_load(sh_cut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint64_t t = __hip_atomic_load(sh_thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        cut = min(cut, (int)__builtin_amdgcn_readfirstlane((int)c));
        const uint64_t tu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 32)) << 32) |
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)t);
        thr = tu < thr ? tu : thr;
    };

    uint32_t n_checked = 0, n_found = 0;  // statistics (a.counters)
    int iter = 0;                         // strips this wavefront has done
    bool first_strip = !(dbg & 2);
    // A workgroup takes a CONTIGUOUS share of the launch's list and its sixteen wavefronts sweep it side by side: with
    // the list in storage order (the one-level plan) a CU's sixty-four streams stay within a few hundred KB of each
    // array at any time -- dealt out across the whole image instead (4096 x 4 streams over 1.5 GB), the first strips of
    // a launch took twice the time of the last ones (address translation).  256 shares are still a spread sample of
    // the shard at every moment, which is what the tightening wants.
    // Its wavefronts DRAW their strips from an LDS counter: the four wavefronts of a SIMD do not advance at the same
    // pace (instruction issue goes to the oldest first: the youngest took 1.6 x the time per strip), and with a fixed deal
    // the fast ones sat idle for the last fifth of the launch.
    const int per_wg = (a.n_seg_pass + (int)gridDim.x - 1) / (int)gridDim.x;
    const int wg_begin = (int)blockIdx.x * per_wg, wg_end = min(a.n_seg_pass, wg_begin + per_wg);
    for (;;) {
        uint32_t drawn = 0;
        if (lane == 0) drawn = __hip_atomic_fetch_add(sh_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int entry = wg_begin + __builtin_amdgcn_readfirstlane((int)drawn);
        if (entry >= wg_end) break;
        const int64_t sid = __builtin_amdgcn_readfirstlane(a.seg_list ? (int)a.seg_list[entry] : entry);
        // looks at strip starts (the pipeline is empty there: a look costs its own latency only), a quarter of the
        // workgroup's wavefronts each round: the others take what those find from LDS (refresh)
        if (hist && !first_strip && ((iter + wave) & 3) == 0) look();
        ++iter;
        // the runs' ancestor stacks (h:2858-2862), and where the strip's phases start (16-byte units)
        const uint32_t pb = a.img.st_pbase[sid * GROUPS + min(lane, GROUPS)];
#pragma unroll
        for (int lv = 0; lv < 8; ++lv) {
            const uint64_t c = a.img.st_ckpt[(sid * 8 + lv) * 64 + lane];
            lds_st64(lane_stk + 512u * (lv + 1), make_uint2((uint32_t)c, (uint32_t)(c >> 32)));
        }
        struct Hdr {
            uint32_t masks, depths;  // the lane's next four nodes: a mask byte and a depth nibble each
        };
        struct Bytes {
            uint2 w[kPhaseLen];  // the (up to) eight changed bytes of each of them, from the node's first byte on
            uint32_t x[4];       // (DPQ_S1_PAIR >= 2: the third (and fourth) dwords of the 12- (16-) byte windows of nodes 0 and 2)
        };
        auto load_hdr = [&](int g) -> Hdr {
            const int gc = min(g, GROUPS - 1);  // (past the strip: the last phase again, never used)
            return Hdr{a.img.st_mask[(sid * GROUPS + gc) * 64 + lane], (uint32_t)a.img.st_depth[(sid * GROUPS + gc) * 64 + lane]};
        };
        auto load_bytes = [&](int g, const Hdr& h) -> Bytes {
            const int gc = min(g, GROUPS - 1);
            Bytes r;
            // where the lane's bytes start in the phase (lane after lane): wave prefix sum of the lanes' byte counts
            const uint32_t mine = (uint32_t)__popc(h.masks);  // <= 32
            const uint32_t off = wave_inclusive_sum(mine) - mine;
            const unsigned char* base = a.img.st_delta + (size_t)(uint32_t)__builtin_amdgcn_readlane((int)pb, gc) * 16;
            const uint32_t o1 = off + (uint32_t)__popc(h.masks & 0xffu), o2 = off + (uint32_t)__popc(h.masks & 0xffffu),
                           o3 = off + (uint32_t)__popc(h.masks & 0xffffffu);
            if constexpr ((DPQ_S1_SKIP & 4) != 0) {
                r.w[0] = make_uint2(off, o1), r.w[1] = make_uint2(o1, o2), r.w[2] = make_uint2(o2, o3), r.w[3] = make_uint2(o3, (uint32_t)(uintptr_t)base);
                return r;
            }
            if constexpr ((DPQ_S1_SKIP & 8) != 0) {  // one of the four loads (the others' registers: copies)
                __builtin_memcpy(&r.w[0], base + off, 8);
                r.w[1] = make_uint2(r.w[0].x + o1, r.w[0].y), r.w[2] = make_uint2(r.w[0].x + o2, r.w[0].y), r.w[3] = make_uint2(r.w[0].x + o3, r.w[0].y);
                return r;
            }
            if constexpr ((DPQ_S1_SKIP & 16) != 0) {  // the four loads, from the image's first 64 KB (cache-resident)
                const unsigned char* b0 = a.img.st_delta + (((size_t)(uint32_t)__builtin_amdgcn_readlane((int)pb, gc) * 16) & 0xf000u);
                __builtin_memcpy(&r.w[0], b0 + off, 8);
                __builtin_memcpy(&r.w[1], b0 + o1, 8);
                __builtin_memcpy(&r.w[2], b0 + o2, 8);
                __builtin_memcpy(&r.w[3], b0 + o3, 8);
                return r;
            }
            if constexpr (DPQ_S1_PAIR == 3) {  // 16-byte windows: a pair always fits
                uint32_t a4[4], c4[4];
                __builtin_memcpy(a4, base + off, 16);
                __builtin_memcpy(c4, base + o2, 16);
                r.w[0] = make_uint2(a4[0], a4[1]), r.x[0] = a4[2], r.x[2] = a4[3], r.w[2] = make_uint2(c4[0], c4[1]), r.x[1] = c4[2], r.x[3] = c4[3];
                return r;
            }
            if constexpr (DPQ_S1_PAIR == 2) {
                // As below with 12-byte windows: a pair runs past one in 0.3 % of the lanes, so the second node's own load is
                // skipped by most wavefronts altogether (the branch around it is wave-uniform) instead of issued for a few lanes.
                uint32_t a3[3], c3[3];
                __builtin_memcpy(a3, base + off, 12);
                __builtin_memcpy(c3, base + o2, 12);
                r.w[0] = make_uint2(a3[0], a3[1]), r.x[0] = a3[2], r.w[2] = make_uint2(c3[0], c3[1]), r.x[1] = c3[2];
                asm volatile("" : "=v"(r.w[1].x), "=v"(r.w[1].y), "=v"(r.w[3].x), "=v"(r.w[3].y));
                if (o2 - off > 12u) __builtin_memcpy(&r.w[1], base + o1, 8);
                if (off + mine - o2 > 12u) __builtin_memcpy(&r.w[3], base + o3, 8);
                return r;
            }
            if constexpr (DPQ_S1_PAIR) {
                // Two loads serve the four nodes wherever a pair's bytes fit one 8-byte window (the second node's are the first
                // load's, shifted: below); only the lanes whose pair runs past it load the second node's bytes themselves.
                __builtin_memcpy(&r.w[0], base + off, 8);
                __builtin_memcpy(&r.w[2], base + o2, 8);
                asm volatile("" : "=v"(r.w[1].x), "=v"(r.w[1].y), "=v"(r.w[3].x), "=v"(r.w[3].y));  // (whatever they hold: read only where loaded)
                if (o2 - off > 8u) __builtin_memcpy(&r.w[1], base + o1, 8);
                if (off + mine - o2 > 8u) __builtin_memcpy(&r.w[3], base + o3, 8);
                return r;
            }
            __builtin_memcpy(&r.w[0], base + off, 8);  // byte addresses: unaligned 8-byte loads
            __builtin_memcpy(&r.w[1], base + o1, 8);
            __builtin_memcpy(&r.w[2], base + o2, 8);
            __builtin_memcpy(&r.w[3], base + o3, 8);
            return r;
        };
        // Headers and bytes wait in RINGS of R slots with compile-time indices -- the phase loop is unrolled R times -- so that
        // no register is ever copied from one queue position to the next: a copy of a register a load is still writing is a
        // wait for that load (the shifted queues of the first version drained every prefetch at the bottom of each phase
        // as soon as the bytes were fetched more than one phase ahead).
        constexpr int R = 4;
        static_assert(D >= 1 && D < R && GROUPS % R == 0, "the header of phase g + D must be in its slot when phase g starts");
        Hdr hq[R];
        Bytes bq[R];
#pragma unroll
        for (int i = 0; i < R; ++i) hq[i] = load_hdr(i);
#pragma unroll
        for (int i = 0; i < D; ++i) bq[i] = load_bytes(i, hq[i]);
#pragma unroll 1
        for (int g0 = 0; g0 < GROUPS; g0 += R)
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int g = g0 + u;
            if (hist) {
                // inside the first strip: wavefronts 0..3 after 1, 2, 4, 8 phases (drains that wavefront's prefetches, once)
                if (first_strip && wave < 4 && g == (1 << wave)) look();
#ifndef DPQ_S1_REFRESH
#define DPQ_S1_REFRESH 4  // phases between two reads of the workgroup's cut (every phase inside the first strip)
#endif
                if (first_strip || (g % DPQ_S1_REFRESH) == 0) refresh();
            }
            const Hdr hdr = hq[u];
            hq[u] = load_hdr(g + R);  // (past the strip: the last phase again, never used)
            bq[(u + D) % R] = load_bytes(g + D, hq[(u + D) % R]);
            const Bytes& cur = bq[u];
            uint32_t codes[kPhaseLen][2], sum[kPhaseLen];
#if defined(DPQ_S1_PAD_VALU) || defined(DPQ_S1_PAD_SALU) || defined(DPQ_S1_PAD_LDS)
            uint32_t pad_v[4] = {1u, 2u, 3u, 4u}, pad_s = 5u, pad_l[4] = {0u, 0u, 0u, 0u};
#endif
            // The four steps as a software pipeline over the LDS round trips: step s + 1's parent and selectors are read
            // behind step s's eight bound-row gathers and before their sums -- one round trip per step instead of two.
            struct Dec {
                uint32_t paddr, lo;
                uint2 parent;
                uint32_t sel_lo, sel_hi;
            };
            auto dec_reads = [&](int st) -> Dec {
                // the reference's stack machine (h:2888-2905): code = stack[depth - 1] with the masked positions replaced
                Dec d;
                const uint32_t depth = (hdr.depths >> (4 * st)) & 0xfu;
                const uint32_t hi = (hdr.masks >> (8 * st + 4)) & 0xfu;
                d.lo = (hdr.masks >> (8 * st)) & 0xfu;
                d.paddr = (depth << 9) + lane_stk;
                if constexpr ((DPQ_S1_SKIP & 2) != 0) {
                    d.parent = make_uint2(d.paddr, hi), d.sel_lo = d.lo * 0x01010101u, d.sel_hi = hi * 0x01010101u;
                    return d;
                }
                d.parent = lds_ld64(d.paddr);
                d.sel_lo = lds_ld32((d.lo << 7) + lane_sel);
                d.sel_hi = lds_ld32((hi << 7) + lane_sel);
                return d;
            };
            Dec dc = dec_reads(0);
#pragma unroll
            for (int st = 0; st < kPhaseLen; ++st) {
                uint2 raw = cur.w[st];
                if constexpr (DPQ_S1_PAIR == 3) {
                    if (st & 1) {  // bytes [pa, pa + 8) of the first node's 16-byte window
                        const uint32_t pa = (uint32_t)__popc((hdr.masks >> (8 * (st - 1))) & 0xffu);
                        const uint32_t w0 = cur.w[st - 1].x, w1 = cur.w[st - 1].y, w2 = cur.x[st >> 1], w3 = cur.x[2 + (st >> 1)];
                        const bool s1 = pa >= 4u, s2 = pa >= 8u;
                        const uint32_t d0 = s2 ? w2 : s1 ? w1 : w0, d1 = s2 ? w3 : s1 ? w2 : w1, d2 = s2 ? 0u : s1 ? w3 : w2;
                        raw = make_uint2(__builtin_amdgcn_alignbyte(d1, d0, pa & 3u), __builtin_amdgcn_alignbyte(d2, d1, pa & 3u));
                    }
                } else if constexpr (DPQ_S1_PAIR == 2) {
                    if (st & 1) {  // bytes [pa, pa + 8) of the first node's 12-byte window, unless the pair ran past it
                        const uint32_t pa = (uint32_t)__popc((hdr.masks >> (8 * (st - 1))) & 0xffu), pn = (uint32_t)__popc((hdr.masks >> (8 * st)) & 0xffu);
                        const uint32_t w0 = cur.w[st - 1].x, w1 = cur.w[st - 1].y, w2 = cur.x[st >> 1];
                        const bool s1 = pa >= 4u, s2 = pa >= 8u;
                        const uint32_t d0 = s2 ? w2 : s1 ? w1 : w0, d1 = s2 ? 0u : s1 ? w2 : w1, d2 = s1 ? 0u : w2;
                        if (pa + pn <= 12u) raw = make_uint2(__builtin_amdgcn_alignbyte(d1, d0, pa & 3u), __builtin_amdgcn_alignbyte(d2, d1, pa & 3u));
                    }
                } else if constexpr (DPQ_S1_PAIR) {
                    if (st & 1) {  // a pair's second node: its own load where the pair ran past 8 bytes, else the first node's window shifted
                        const uint32_t pa = (uint32_t)__popc((hdr.masks >> (8 * (st - 1))) & 0xffu), pn = (uint32_t)__popc((hdr.masks >> (8 * st)) & 0xffu);
                        const uint2 fst = cur.w[st - 1];
                        const uint64_t sh = ((((uint64_t)fst.y << 32) | fst.x) >> ((8u * pa) & 63u));
                        if (pa + pn <= 8u) raw = make_uint2((uint32_t)sh, (uint32_t)(sh >> 32));
                    }
                }
                const uint32_t raw_hi = (uint32_t)((((uint64_t)raw.y << 32) | raw.x) >> (8 * __popc(dc.lo)));
                uint32_t* code = codes[st];
                code[0] = __builtin_amdgcn_perm(dc.parent.x, raw.x, dc.sel_lo);
                code[1] = __builtin_amdgcn_perm(dc.parent.y, raw_hi, dc.sel_hi);
                if constexpr ((DPQ_S1_SKIP & 2) == 0) lds_st64(dc.paddr + 512u, make_uint2(code[0], code[1]));  // stack[depth] = code
                // ADC, lower bound: sum of the eight entries; row of code byte c at c << 8, the lane's copy at lane8
                if constexpr ((DPQ_S1_SKIP & 1) != 0) {
                    if (st + 1 < kPhaseLen) dc = dec_reads(st + 1);
                    sum[st] = 300u + (code[0] ^ code[1]);
                    continue;
                }
                uint32_t row[M];
#pragma unroll
                for (int m = 0; m < M; ++m) row[m] = __builtin_amdgcn_perm(code[m >> 2], lane8, 0x0c0c0400u + ((uint32_t)(m & 3) << 8));
                if constexpr (DPQ_S1_U8) {
                    uint32_t e[M];
#pragma unroll
                    for (int m = 0; m < M; ++m) e[m] = lds_ld8(row[m] + (uint32_t)(((m >> 2) << 7) | (m & 3)));
                    if (st + 1 < kPhaseLen) dc = dec_reads(st + 1);
                    sum[st] = ((e[0] + e[1] + e[2]) + (e[3] + e[4])) + (e[5] + e[6] + e[7]);
#if defined(DPQ_S1_PAD_VALU) || defined(DPQ_S1_PAD_SALU) || defined(DPQ_S1_PAD_LDS)
                    {  // (timing experiments: instructions that do nothing, per step)
#ifdef DPQ_S1_PAD_VALU
#pragma unroll
                        for (int i = 0; i < DPQ_S1_PAD_VALU; ++i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(pad_v[i & 3]) : "v"(lane8), "v"(code[0]));
#endif
#ifdef DPQ_S1_PAD_SALU
#pragma unroll
                        for (int i = 0; i < DPQ_S1_PAD_SALU; ++i) asm volatile("s_mov_b32 %0, %0" : "+s"(pad_s));
#endif
#ifdef DPQ_S1_PAD_LDS
#pragma unroll
                        for (int i = 0; i < DPQ_S1_PAD_LDS; ++i) {
                            uint32_t t;
                            asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(t) : "v"(row[i & 7]), "n"(4 * (i & 7)));
                            pad_l[i & 3] = t;
                        }
#endif
                    }
#endif
                    continue;
                }
                uint2 e[M];
#pragma unroll
                for (int m = 0; m < M; ++m) e[m] = lds_ld64_whole(row[m]);
                if (st + 1 < kPhaseLen) dc = dec_reads(st + 1);
                uint32_t acc0 = 0, acc1 = 0;  // two chains
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    acc0 = __builtin_amdgcn_udot4(e[m].x, 1u << (8 * m), acc0, false);
                    acc1 = __builtin_amdgcn_udot4(e[m + 4].y, 1u << (8 * m), acc1, false);
                }
                sum[st] = acc0 + acc1;
            }
#if defined(DPQ_S1_PAD_VALU) || defined(DPQ_S1_PAD_SALU) || defined(DPQ_S1_PAD_LDS)
            asm volatile("" ::"v"(pad_v[0]), "v"(pad_v[1]), "v"(pad_v[2]), "v"(pad_v[3]), "s"(pad_s), "v"(pad_l[0]), "v"(pad_l[1]), "v"(pad_l[2]), "v"(pad_l[3]));
#endif
            // what the bound lets through: the reference's distance and the whole (distance, id) key
            const uint32_t cutp = (uint32_t)cut + 1u;
            if (__ballot(min(min(sum[0], sum[1]), min(sum[2], sum[3])) <= cutp) && !(dbg & 1)) {
                const unsigned long long t_in = stp ? __builtin_amdgcn_s_memrealtime() : 0ull;
#pragma unroll
                for (int st = 0; st < kPhaseLen; ++st) {
                    const uint32_t c0 = codes[st][0], c1 = codes[st][1], sm = sum[st];
                    const int64_t node = sid * kStripNodes + lane * kRunLen + g * kPhaseLen + st;
                    bool pass = sm <= cutp && node < a.img.n_local;
                    const uint64_t checked = __ballot(pass);
                    if (checked == 0) continue;  // wave-uniform
                    uint64_t key = 0;
                    float d = 0.0f;
                    if (pass) {
                        const uint32_t c[2] = {c0, c1};
                        double dsum = 0.0;
#pragma unroll
                        for (int m = 0; m < M; ++m) dsum = __dadd_rn(dsum, (double)T32[m * 256 + ((c[m >> 2] >> (8 * (m & 3))) & 0xffu)]);
                        d = (float)dsum;
                        key = make_key(d, a.img.id_base + (uint32_t)node);
                        pass = key <= thr;
                    }
                    const uint64_t found = __ballot(pass);
                    n_checked += (uint32_t)__popcll(checked);
                    n_found += (uint32_t)__popcll(found);
                    if (found == 0) continue;
                    // the workgroup's own region of the slot's buffer, filled through an LDS counter (no global atomics)
                    uint64_t* region = a.cand_key + (size_t)slot * a.cand_stride + a.region_off + (size_t)blockIdx.x * a.region_cap;
                    uint32_t base = 0;
                    if (lane == 0) base = __hip_atomic_fetch_add(sh_count, (uint32_t)__popcll(found), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    const uint32_t li = base + mbcnt64(found, 0);
                    if (pass && li < (uint32_t)a.region_cap) region[li] = key;
                    if (hist && pass) {
                        // the candidate's distance in units of the bound table, rounded against it (as scan_kernel's
                        // tightening: the roundings are below 1e-4 units, the margin 2e-3): it lies under a cut of e units
                        const float u = __fmaf_rn(d * sc, 0x1.000002p-20f, (d - bdn) * sc);
                        const int e = max((int)(u + 2e-3f) + 1, 0);
                        // bucket b: under a cut of 4 b + 3 units; counted in LDS, moved to the global histogram by the next look
                        if (e < cut) __hip_atomic_fetch_add(sh_hist + (e >> 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                if (stp) {
                    slow_ticks += __builtin_amdgcn_s_memrealtime() - t_in;
                    ++slow_entries;
                }
            }
        }
        first_strip = false;
        stamp();
        if (stp && n_stamp == 3 && lane == 0) stp[14] = slow_ticks, stp[15] = slow_entries;  // ... of the first strip
    }
    // ---- epilogue: the last wavefront of the workgroup publishes its region's count (a count above region_cap tells
    // the select kernel that keys were dropped) ----
    uint32_t done = 0;
    if (lane == 0) {
        if (a.counters) {  // (statistics: summed over the workgroup in LDS, two global adds per workgroup)
            __hip_atomic_fetch_add(sh_stat, n_checked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(sh_stat + 1, n_found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        done = __hip_atomic_fetch_add(sh_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (done == kS1Waves - 1) {
            a.cand_count[(size_t)slot * kRegionStride + 1 + blockIdx.x] = __hip_atomic_load(sh_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (a.counters) {
                const uint32_t c0 = __hip_atomic_load(sh_stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t c1 = __hip_atomic_load(sh_stat + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (c0) atomicAdd(a.counters, (unsigned long long)c0);
                if (c1) atomicAdd(a.counters + 1, (unsigned long long)c1);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// a6: select.  grid = slots, block = 512 (level 0) or 256 threads, dynamic LDS
// ---------------------------------------------------------------------------

__device__ __forceinline__ void block_bitonic_sort(uint64_t* v, int n_pow2, int tid, int nthreads) {
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < n_pow2 / 2; t += nthreads) {
                // t-th compare-exchange of this stage: i has bit j clear
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int ixj = i | j;
                const uint64_t x = v[i], y = v[ixj];
                const bool up = (i & k) == 0;
                if ((x > y) == up) {
                    v[i] = y;
                    v[ixj] = x;
                }
            }
            __syncthreads();
        }
    }
}

// k-th smallest (1-based rank `rank`) of keys[0..n) -- LDS or HBM -- by MSB-first
// radix select with digits of up to 8 bits.  Distances of one query share their
// sign/exponent bits, so raw keys would put every element into one bin for the
// first passes (thousands of serialised LDS atomics): the passes run on
// (key - min) and the first digit starts at the top set bit of (max - min).
// Histogram with LDS atomics, bin scan by one wavefront (no barriers inside),
// 3 barriers per pass; the walk stops as soon as the chosen bin holds one key.
// `ignore` (= ~0) marks padding entries; they sort last and are never selected
// because rank <= number of valid keys.
// K = uint64_t (distance bits << 32 | id) or uint32_t (distance bits alone; widened on the fly, 0xffffffff = padding).
template <class K>
__device__ uint64_t block_radix_select(const K* keys_in, int n, int rank, uint32_t* hist, uint32_t* bcast,
                                       int tid, int nthreads) {
    struct Loader {
        const K* p;
        __device__ __forceinline__ uint64_t operator[](int i) const {
            if constexpr (sizeof(K) == 8) return p[i];
            const uint32_t v = p[i];
            return v == 0xffffffffu ? ~0ull : (uint64_t)v;
        }
    };
    const Loader keys{keys_in};
    // block min / max of the valid keys (padding = ~0 is excluded from max)
    uint64_t lo = ~0ull, hi = 0ull;
    for (int i = tid; i < n; i += nthreads) {
        const uint64_t k = keys[i];
        if (k != ~0ull) {
            lo = k < lo ? k : lo;
            hi = k > hi ? k : hi;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t olo = ((uint64_t)(uint32_t)__shfl_xor((int)(lo >> 32), off, 64) << 32) |
                             (uint32_t)__shfl_xor((int)(uint32_t)lo, off, 64);
        const uint64_t ohi = ((uint64_t)(uint32_t)__shfl_xor((int)(hi >> 32), off, 64) << 32) |
                             (uint32_t)__shfl_xor((int)(uint32_t)hi, off, 64);
        lo = olo < lo ? olo : lo;
        hi = ohi > hi ? ohi : hi;
    }
    uint64_t* mm = reinterpret_cast<uint64_t*>(hist);  // [2 * waves] scratch, hist is free until the first pass
    if ((tid & 63) == 0) {
        mm[2 * (tid >> 6)] = lo;
        mm[2 * (tid >> 6) + 1] = hi;
    }
    __syncthreads();
    for (int w = 0; w < nthreads / 64; ++w) {
        lo = mm[2 * w] < lo ? mm[2 * w] : lo;
        hi = mm[2 * w + 1] > hi ? mm[2 * w + 1] : hi;
    }
    __syncthreads();
    const uint64_t span = hi - lo;                                   // valid keys live in [0, span] after the shift
    // digits are cut from the most significant set bit of span downwards, so the first pass already
    // spreads the keys over >= 128 bins (a byte-aligned first digit may hold only a few values)
    const int hi_bit = span ? 63 - __clzll((long long)span) : 0;
    int shift = hi_bit > 7 ? hi_bit - 7 : 0;
    int width = hi_bit - shift + 1;  // <= 8
    bool first = true;
    uint64_t prefix = 0;
    uint32_t rem = (uint32_t)rank;
    for (;;) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const uint32_t dmask = (1u << width) - 1u;
        for (int i = tid; i < n; i += nthreads) {
            const uint64_t raw = keys[i];
            if (raw == ~0ull) continue;
            const uint64_t key = raw - lo;
            const bool match = first ? true : ((key >> (shift + width)) == (prefix >> (shift + width)));
            if (match) atomicAdd(&hist[(uint32_t)(key >> shift) & dmask], 1u);
        }
        __syncthreads();
        if (tid < 64) {  // wave 0: lane l owns bins 4l..4l+3
            const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const uint32_t mine = h0 + h1 + h2 + h3;
            uint32_t incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
                if (tid >= off) incl += up;
            }
            const uint32_t before = incl - mine;
            if (before < rem && rem <= incl) {  // exactly one lane
                uint32_t r = rem - before;
                int bin = 4 * tid;
                if (r > h0) { r -= h0; ++bin; if (r > h1) { r -= h1; ++bin; if (r > h2) { r -= h2; ++bin; } } }
                const uint32_t hb = bin == 4 * tid ? h0 : bin == 4 * tid + 1 ? h1 : bin == 4 * tid + 2 ? h2 : h3;
                bcast[0] = (uint32_t)bin | (hb == 1u ? 0x100u : 0u);  // bit 8: the bin holds exactly one key
                bcast[1] = r;
            }
        }
        __syncthreads();
        prefix |= (uint64_t)(bcast[0] & 0xffu) << shift;
        rem = bcast[1];
        if (shift == 0) break;
        if (bcast[0] & 0x100u) {
            // one key left under this prefix: fetch it instead of walking the remaining digits
            uint64_t* found = reinterpret_cast<uint64_t*>(hist);  // the histogram is no longer needed
            for (int i = tid; i < n; i += nthreads) {
                const uint64_t raw = keys[i];
                if (raw != ~0ull && ((raw - lo) >> shift) == (prefix >> shift)) *found = raw;
            }
            __syncthreads();
            const uint64_t r = *found;
            __syncthreads();  // callers reuse hist
            return r;
        }
        width = shift < 8 ? shift : 8;
        shift -= width;
        first = false;
    }
    return prefix + lo;
}

// An UPPER BOUND of the rank-th smallest of the 32-bit keys[0..n) (LDS) from ONE histogram pass: 2^BITS bins over
// (key - min) cut from the top set bit of (max - min) downwards; the bound is the upper edge of the bin the rank falls
// into (never above max), i.e. less than (max - min) / 2^(BITS - 1) above the exact answer -- on fp32 distances of one
// query 2^-9 of a binade at most, a tenth of a filter unit.  Three barriers and 32-bit arithmetic (block_radix_select:
// about fourteen, on widened keys).  lo / hi: the calling thread's own minimum / maximum over the keys IT wrote (every
// key counted by some thread); hist: [2^BITS] words, xch: [2 * waves + 1] words; rank <= n, no padding keys.
template <int BITS>
__device__ uint32_t block_kth_bound_u32(const uint32_t* keys, int n, int rank, uint32_t lo, uint32_t hi, uint32_t* hist,
                                        uint32_t* xch, int tid, int nthreads) {
    constexpr int NB = 1 << BITS;
    static_assert(NB % 256 == 0, "wave 0 scans NB / 64 bins per lane, four at a time");
    for (int i = tid; i < NB; i += nthreads) hist[i] = 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, off, 64));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, off, 64));
    }
    const int nw = nthreads >> 6;
    if ((tid & 63) == 0) {
        xch[2 * (tid >> 6)] = lo;
        xch[2 * (tid >> 6) + 1] = hi;
    }
    __syncthreads();
    for (int w = 0; w < nw; ++w) {
        lo = min(lo, xch[2 * w]);
        hi = max(hi, xch[2 * w + 1]);
    }
    const uint32_t span = hi - lo;
    const int hi_bit = span ? 31 - __clz((int)span) : 0;
    const int shift = hi_bit >= BITS ? hi_bit - (BITS - 1) : 0;
    for (int i = tid; i < n; i += nthreads) atomicAdd(&hist[(keys[i] - lo) >> shift], 1u);
    __syncthreads();
    if (tid < 64) {  // wave 0: lane l owns bins [PER l, PER (l + 1))
        constexpr int PER = NB / 64;
        uint32_t h[PER];
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < PER; j += 4) {
            const uint4 v = *reinterpret_cast<const uint4*>(hist + PER * tid + j);
            h[j] = v.x; h[j + 1] = v.y; h[j + 2] = v.z; h[j + 3] = v.w;
            mine += v.x + v.y + v.z + v.w;
        }
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
            if (tid >= off) incl += up;
        }
        const uint32_t before = incl - mine;
        if (before < (uint32_t)rank && (uint32_t)rank <= incl) {  // exactly one lane
            uint32_t r = (uint32_t)rank - before;
            int bin = PER * tid + PER - 1;
            bool found = false;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                if (!found && r <= h[j]) {
                    bin = PER * tid + j;
                    found = true;
                }
                if (!found) r -= h[j];
            }
            xch[2 * nw] = (uint32_t)bin;
        }
    }
    __syncthreads();
    const uint64_t edge = (uint64_t)lo + ((((uint64_t)xch[2 * nw]) + 1ull) << shift) - 1ull;
    return edge < (uint64_t)hi ? (uint32_t)edge : hi;
}

__device__ __forceinline__ int32_t report_id(uint32_t pos, int64_t n_total) {
    // h:2949, 2970: for even N the trailing node is pushed as i+1 == N, not N-1.
    if ((n_total & 1) == 0 && (int64_t)pos == n_total - 1) return (int32_t)n_total;
    return (int32_t)pos;
}

#ifndef DPQ_RANK_BY_COUNT_MAX
// winners up to which the final order comes from counting smaller winners instead of a sort (kk broadcast reads per winner:
// it was the rule up to 2 x THREADS winners; top-300 3.91 -> 4.33 M q/s, top-512 3.20 -> 3.40 M with the sort from 257 on)
#define DPQ_RANK_BY_COUNT_MAX 256
#endif
// Candidate keys a block of the later levels holds in LDS: what leaves the block within 39 KB together with the winner area
// behind the list (four blocks per CU, so the 1024 blocks of a batch are resident at once): 3968 keys at top-100, 3520 at
// top-512, 3072 at top-1000, 1984 at top-2048 (longer lists are selected from their HBM copy, as lists beyond kSortMax always
// were).
// The winner area behind the list (8-byte units): the exact way's winner array (top_k rounded up to a power of two, for
// its bitonic network), or the last level's histogram (1024 words) + bucket-sorted winner list (top_k + kBucketSlack) --
// whichever is larger, so that the last level's bucket sort always has its room whatever the list holds.
constexpr int kBucketSlack = 256;  // keys the k-th key's bin may bring beyond top_k, and the fullest bin the bucket sort ranks
__host__ __device__ inline int select_area_keys(int top_k) {
    int kp = 1;
    while (kp < top_k) kp <<= 1;
    return kp > 512 + kBucketSlack + top_k ? kp : 512 + kBucketSlack + top_k;
}
// (39 KB, not 40: with 40 824 B per block the per-block stamps showed three blocks per CU -- a quarter of a batch's blocks
// started 19 us late -- while 39 256 B run four)
__host__ __device__ inline int select_list_keys(int top_k) {
    const int room = (39936 - 16 - 266 * 4 - select_area_keys(top_k) * 8) / 8;
    return room >= kSortMax ? kSortMax : (room > 1024 ? room : 1024) & ~63;
}

// THREADS: 512 for level 0 (3840 nodes to evaluate), 256 for the later levels (about a thousand keys: fewer
// wavefronts per barrier, 16-19 us instead of 19-21)
// Second launch bound (eight wavefronts per SIMD): without it the compiler took 94 + 6 SGPRs = 112 allocated, i.e. seven
// wavefronts per SIMD (800 SGPRs each), and a CU held three 512-thread blocks instead of four -- every 512-thread launch
// (top_k > 512) ran its 1000 blocks in two rounds (per-block stamps: a quarter of the blocks started 20 - 45 us late,
// profiles/r04b_select_stamps_large_k.txt).
template <int M, int THREADS>
__global__ __launch_bounds__(THREADS, 8) void select_kernel(const SelectArgs a) {
    constexpr int W = Cfg<M>::W;
    constexpr int TE = M * 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the winner array is sized by top_k (rounded up to a power of two for the final sort), not by the
    // 2048 maximum: at top_k = 100 a block needs 35 KB and four blocks share a CU
    int KP = 1;
    while (KP < a.top_k) KP <<= 1;
    // LDS: [candidate keys][winner keys, histogram | level 0: the query's exact tables][counters].
    // The tables are dead once the level-0 list is evaluated, so they share their space with what the
    // selection needs afterwards: a level-0 block stays under 40 KB and four of them fit a CU.
    const int n_lds_keys = a.shared_id ? min(a.shared_n, kSortMax) : select_list_keys(a.top_k);
    uint64_t* skeys = reinterpret_cast<uint64_t*>(smem);                              // [n_lds_keys] candidate keys
    unsigned char* shared_area = smem + (size_t)n_lds_keys * 8;
    float* T = reinterpret_cast<float*>(shared_area);                                 // [M][256], level 0 only
    uint64_t* wkeys = reinterpret_cast<uint64_t*>(shared_area);                       // [KP] winner keys
    const int AREA = a.shared_id ? KP : select_area_keys(a.top_k);                    // 8-byte units behind the list
    uint32_t* hist = reinterpret_cast<uint32_t*>(wkeys + AREA);                       // [kRegionStride + 1 <= 264]
    uint32_t* bcast = hist + 264;                                                     // [2]
    const size_t select_bytes = (size_t)AREA * 8 + 266 * 4, table_bytes = a.shared_id ? (size_t)TE * 4 : 0;
    uint32_t* counters = reinterpret_cast<uint32_t*>(shared_area + (select_bytes > table_bytes ? select_bytes : table_bytes));  // [2]: winners, padding nodes

    const int slot = blockIdx.x;
    const int tid = threadIdx.x;
    const int q = a.slot_query ? a.slot_query[slot] : slot;
    if (q < 0) return;  // unused slot of a rerun group
    auto mark = [&](int i) {  // developer diagnostics: phase boundaries
        if (a.stamps && tid == 0) a.stamps[(size_t)slot * 8 + i] = __builtin_amdgcn_s_memtime();
    };
    mark(0);
    if (a.stamps && tid == 0) a.stamps[(size_t)slot * 8 + 6] = __builtin_amdgcn_s_memrealtime();
    const bool shared = a.shared_id != nullptr;
    uint64_t* cand = a.cand_key + (size_t)slot * a.cand_stride;
    uint32_t* region_n = a.cand_count + (size_t)slot * kRegionStride;
    uint64_t* keys = skeys;
    int n = 0;
    uint32_t dlo = 0xffffffffu, dhi = 0u;  // extremes of the distance bits of the keys this thread gathered
    if (tid == 0) {
        counters[0] = 0;
        counters[1] = 0;
    }
    if (shared) {
        // level 0: the query-independent list of decoded nodes, evaluated exactly here.  The query's
        // exact tables are staged in LDS first: every node gathers M entries, and scattered gathers
        // from global memory are bound by the vector-memory address rate.
        n = a.shared_n;
        if (n > kSortMax) keys = a.scratch + (size_t)slot * a.cand_stride;
        const float4* src = reinterpret_cast<const float4*>(a.lut32 + (size_t)q * TE);
        for (int i = tid; i < TE / 4; i += THREADS) reinterpret_cast<float4*>(T)[i] = src[i];
        __syncthreads();
        // four nodes per thread and pass: their ids and codes are fetched before any is evaluated
        for (int i0 = tid; i0 < n; i0 += 4 * THREADS) {
            uint32_t id[4], c[4][W];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * THREADS, n - 1);
                id[u] = a.shared_id[i];
#pragma unroll
                for (int w = 0; w < W; ++w) c[u][w] = a.shared_code[(size_t)W * i + w];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * THREADS;
                if (i >= n) break;
                uint64_t key = ~0ull;
                if (id[u] != 0xffffffffu)
                    key = make_key(exact_dist<M>(T, c[u], a.fp32_accum != 0), id[u]);
                else
                    atomicAdd(&counters[1], 1u);  // padding node
                keys[i] = key;
            }
        }
    } else {
        // later levels: the scan already evaluated its survivors exactly.  Gather the regions of the
        // slot's buffer (carried winners + one per scan workgroup) into one list.
        uint32_t* rstart = hist;  // [n_regions + 1] exclusive prefix of the region sizes (hist is free until the select)
        const int R = a.n_regions;
        if (tid < 64) {  // wave 0: R <= 257
            uint32_t carry = 0;
            bool dropped = false;
            for (int r0 = 0; r0 < R; r0 += 64) {
                const int r = r0 + tid;
                const uint32_t capr = r == 0 ? (uint32_t)a.region_off : (uint32_t)a.region_cap;
                const uint32_t raw = r < R ? region_n[r] : 0u;
                dropped |= raw > capr;
                const uint32_t mine = min(raw, capr);
                uint32_t incl = mine;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
                    if (tid >= off) incl += up;
                }
                if (r < R) rstart[r] = carry + incl - mine;
                carry += (uint32_t)__shfl((int)incl, 63, 64);
            }
            if (tid == 0) rstart[R] = carry;
            // candidates were dropped at this level: the final list may miss entries -> host reruns this query
            if (__any(dropped) && tid == 0) {
                a.overflow[slot] = 1u;
                if (a.any_overflow) *reinterpret_cast<volatile uint32_t*>(a.any_overflow) = 1u;  // host-visible summary
            }
        }
        __syncthreads();
        n = (int)rstart[R];
        // keys live in LDS for the usual list sizes, in HBM scratch for huge ones (overflow reruns)
        if (n > n_lds_keys) keys = a.scratch + (size_t)slot * a.cand_stride;
        // flat gather: key i of the list sits in the region r with rstart[r] <= i < rstart[r + 1]; every thread
        // has all its loads in flight at once (a region-by-region copy is a chain of global round trips)
        for (int i0 = 0; i0 < n; i0 += 4 * THREADS) {
            uint64_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + tid + u * THREADS, n - 1);
                int lo = 0, hi = R;  // rstart[lo] <= i < rstart[hi]
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (rstart[mid] <= (uint32_t)i) lo = mid; else hi = mid;
                }
                const uint64_t* src = cand + (lo == 0 ? (size_t)0 : (size_t)a.region_off + (size_t)(lo - 1) * a.region_cap);
                v[u] = src[(uint32_t)i - rstart[lo]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + tid + u * THREADS;
                if (i < n) {
                    keys[i] = v[u];
                    dlo = min(dlo, (uint32_t)(v[u] >> 32));
                    dhi = max(dhi, (uint32_t)(v[u] >> 32));
                }
            }
        }
    }
    __syncthreads();
    mark(1);
    const int n_valid = n - (int)counters[1];
    const int kk = min(a.top_k, n_valid);
    // The LAST level (round 4): selection and order from ONE histogram pass -- a bucket sort.  1024 bins over (key - least
    // distance bits << 32) from the top set bit of the span (the gather kept the distance extremes in registers); wave 0
    // scans the bins, finds the bin B the k-th key lies in and turns the counts into bin starts; every key of a bin <= B is
    // scattered to its bin's range of the winner list (the bin's word is its cursor: one LDS atomic per winner), and a
    // winner's output rank = its bin's start + the smaller keys INSIDE its bin (one to three keys per bin as a rule).  Three
    // barriers and two passes over the keys, against the radix select's two or three passes + single-key fetch + winner
    // compaction (nine barriers, six passes) followed by kk broadcast reads per winner (top_k <= 256) or a bitonic network of
    // 55 barrier stages (top-1000).  The histogram and the winner list live behind the keys (select_area_keys keeps their
    // room whatever the list holds; the keys may be in the HBM scratch).  A crowded bin (more than kBucketSlack keys: ties by the
    // hundred) or a k-th bin that brings more than kBucketSlack keys too many takes the exact way below.  The slot's threshold for a rerun after an overflow is
    // bin B's upper edge.
    const int n_even = (n + 1) & ~1;
    const int fast_base = keys == skeys ? n_even : 0;  // u64 units of the key + winner area in use by the keys
    if (a.fast_final && !shared && a.final_pass && kk == a.top_k) {  // (fast_base <= n_lds_keys, 512 + kk + kBucketSlack <= AREA)
        uint32_t* h1 = reinterpret_cast<uint32_t*>(skeys + fast_base);  // [1024]
        uint64_t* wp = skeys + fast_base + 512;                          // [kk + kBucketSlack]
        for (int i = tid; i < 1024; i += THREADS) h1[i] = 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            dlo = min(dlo, (uint32_t)__shfl_xor((int)dlo, off, 64));
            dhi = max(dhi, (uint32_t)__shfl_xor((int)dhi, off, 64));
        }
        if ((tid & 63) == 0) {
            hist[2 * (tid >> 6)] = dlo;
            hist[2 * (tid >> 6) + 1] = dhi;
        }
        __syncthreads();
        for (int w = 0; w < THREADS / 64; ++w) {
            dlo = min(dlo, hist[2 * w]);
            dhi = max(dhi, hist[2 * w + 1]);
        }
        const uint64_t lo64 = (uint64_t)dlo << 32, hi64 = ((uint64_t)dhi << 32) | 0xffffffffull;
        const int hi_bit = 63 - __clzll((long long)(hi64 - lo64));  // >= 31; <= 62: distance bits are those of a float >= 0
        const int shift = hi_bit - 9;
        for (int i = tid; i < n; i += THREADS) atomicAdd(&h1[(uint32_t)((keys[i] - lo64) >> shift)], 1u);
        __syncthreads();
        if (tid < 64) {  // wave 0: lane l owns bins [16 l, 16 l + 16)
            uint32_t h[16], mine = 0;
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                const uint4 v = *reinterpret_cast<const uint4*>(h1 + 16 * tid + j);
                h[j] = v.x; h[j + 1] = v.y; h[j + 2] = v.z; h[j + 3] = v.w;
                mine += v.x + v.y + v.z + v.w;
            }
            uint32_t incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
                if (tid >= off) incl += up;
            }
            const uint32_t before = incl - mine;
            // the bin of the k-th key (found by exactly one lane), the keys up to and including it
            const bool owner = before < (uint32_t)kk && (uint32_t)kk <= incl;
            uint32_t run = before, bin = 0, upto = 0;
            bool found = false;
            // counts -> bin starts (the scatter's cursors), four bins at a time
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                uint32_t st[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    st[i] = run;
                    run += h[j + i];
                    if (owner && !found && run >= (uint32_t)kk) {
                        bin = (uint32_t)(16 * tid + j + i);
                        upto = run;
                        found = true;
                    }
                }
                *reinterpret_cast<uint4*>(h1 + 16 * tid + j) = make_uint4(st[0], st[1], st[2], st[3]);
            }
            const int src = __ffsll((unsigned long long)__ballot(owner)) - 1;  // kk <= n: some lane owns it
            bin = (uint32_t)__shfl((int)bin, src, 64);
            upto = (uint32_t)__shfl((int)upto, src, 64);
            uint32_t crowd = 0;  // the fullest bin up to B
#pragma unroll
            for (int j = 0; j < 16; ++j) crowd = max(crowd, (uint32_t)(16 * tid + j) <= bin ? h[j] : 0u);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) crowd = max(crowd, (uint32_t)__shfl_xor((int)crowd, off, 64));
            if (tid == 0) {
                bcast[0] = bin;
                bcast[1] = upto;
                hist[64] = crowd;
            }
        }
        __syncthreads();
        const uint32_t bin = bcast[0];
        const int upto = (int)bcast[1];
        if (upto <= kk + kBucketSlack && hist[64] <= (uint32_t)kBucketSlack) {  // block-uniform
            if (tid == 0) {
                const uint64_t edge = lo64 + (((uint64_t)bin + 1ull) << shift) - 1ull;
                uint64_t t = edge < hi64 ? edge : hi64;
                if (a.keep_thr) t = min(t, a.thr_key[slot]);
                a.thr_key[slot] = t;
            }
            for (int i = tid; i < n; i += THREADS) {
                const uint64_t key = keys[i];
                const uint32_t b = (uint32_t)((key - lo64) >> shift);
                if (b <= bin) wp[atomicAdd(&h1[b], 1u)] = key;
            }
            __syncthreads();
            mark(2);
            mark(3);
            // bin b now spans [h1[b - 1], h1[b]) of the winner list (its cursor ran to the next bin's start)
            for (int i = tid; i < upto; i += THREADS) {
                const uint64_t mine = wp[i];
                const uint32_t b = (uint32_t)((mine - lo64) >> shift);
                const int s0 = b ? (int)h1[b - 1] : 0, e0 = (int)h1[b];
                int rank = s0;
                for (int j = s0; j < e0; ++j) rank += wp[j] < mine ? 1 : 0;
                if (rank < kk) {
                    const size_t o = (size_t)q * a.top_k + rank;
                    a.out_ids[o] = report_id((uint32_t)(mine & 0xffffffffu), a.n_codes_total);
                    a.out_dists[o] = __uint_as_float((uint32_t)(mine >> 32));
                }
            }
            mark(4);
            if (a.stamps && tid == 0) a.stamps[(size_t)slot * 8 + 7] = __builtin_amdgcn_s_memrealtime();
            return;
        }
        __syncthreads();  // hist / bcast are reused by the exact way
    }
    uint64_t kth = ~0ull;
    if (kk > 0) kth = block_radix_select(keys, n, kk, hist, bcast, tid, THREADS);
    mark(2);

    // The k-th smallest key seen so far bounds the final k-th key from above
    // (candidates are real nodes), so it is the next level's threshold.
    if (tid == 0) {
        uint64_t t = n_valid >= a.top_k ? kth : ~0ull;
        // after a bootstrap the previous threshold's nodes are not among this level's keys: keep the tighter one
        if (a.keep_thr) t = min(t, a.thr_key[slot]);
        a.thr_key[slot] = t;
    }

    // winners = the kk keys <= kth (keys are unique: the id is part of the key)
    int p2 = 1;
    while (p2 < kk) p2 <<= 1;
    if (a.final_pass)
        for (int i = tid; i < p2; i += THREADS) wkeys[i] = ~0ull;
    __syncthreads();
    for (int i = tid; i < n; i += THREADS) {
        const uint64_t key = keys[i];
        if (kk > 0 && key <= kth) {
            const uint32_t pos = atomicAdd(&counters[0], 1u);
            if (pos < (uint32_t)KP) wkeys[pos] = key;
        }
    }
    __syncthreads();

    mark(3);
    if (!a.final_pass) {
        // carry the winners: compact them to the front; the next level appends behind
        for (int i = tid; i < kk; i += THREADS) cand[i] = wkeys[i];
        if (tid == 0) region_n[0] = (uint32_t)kk;
        return;
    }

    if (kk <= DPQ_RANK_BY_COUNT_MAX) {
        // few winners: a winner's output rank = the number of smaller winners (keys are unique); kk LDS
        // broadcast reads per thread and no barrier, against the 28+ barriers of a bitonic network
        for (int i = tid; i < kk; i += THREADS) {
            const uint64_t mine = wkeys[i];
            int rank = 0;
            for (int j = 0; j < kk; ++j) rank += wkeys[j] < mine ? 1 : 0;
            const size_t o = (size_t)q * a.top_k + rank;
            a.out_ids[o] = report_id((uint32_t)(mine & 0xffffffffu), a.n_codes_total);
            a.out_dists[o] = __uint_as_float((uint32_t)(mine >> 32));
        }
        for (int r = kk + tid; r < a.top_k; r += THREADS) {
            const size_t o = (size_t)q * a.top_k + r;
            a.out_ids[o] = -1;
            a.out_dists[o] = INFINITY;
        }
        mark(4);
        if (a.stamps && tid == 0) a.stamps[(size_t)slot * 8 + 7] = __builtin_amdgcn_s_memrealtime();
        return;
    }
    block_bitonic_sort(wkeys, p2, tid, THREADS);
    for (int r = tid; r < a.top_k; r += THREADS) {
        const size_t o = (size_t)q * a.top_k + r;
        if (r < kk) {
            a.out_ids[o] = report_id((uint32_t)(wkeys[r] & 0xffffffffu), a.n_codes_total);
            a.out_dists[o] = __uint_as_float((uint32_t)(wkeys[r] >> 32));
        } else {
            a.out_ids[o] = -1;
            a.out_dists[o] = INFINITY;
        }
    }
}

// ---------------------------------------------------------------------------
// Threshold bootstrap (replaces cascade level 0 and the early filter levels on
// indexes of >= 256 K nodes).  The index carries an inverted multi-index over
// its nodes: cell = (code[0], code[1]), 65536 cells, entries = (global position,
// decoded code).  A query ranks the 256 centroids of sub-spaces 0 and 1 by its
// table rows, walks the cells (i-th best of sub-space 0, j-th best of
// sub-space 1) in shells of growing max(i, j), evaluates the nodes it meets
// EXACTLY (same distance rule and keys as everywhere else) until it has
// `target` of them, and takes the k-th smallest key as its first threshold.
// These are real nodes, so the threshold is a valid upper bound of the final
// k-th key; being the best of the nodes nearest to the query in two of the
// sub-spaces it is as tight as the k-th of a spread sample of a quarter of the
// index (scripts/sim_bootstrap.py), at the cost of ~3 K exact evaluations.
// Only the threshold leaves the kernel: the nodes are met again by the scan.
// grid = slots, block = 512 threads.
// ---------------------------------------------------------------------------
constexpr int kBootThreads = 512;
constexpr int kBootCells = 736;  // cells per round (their node prefix, clamped to 16 bits, and first entries live in LDS);
                                 // 736: the block stays at 40 912 B of LDS at cap = 3072, four blocks per CU
constexpr int kBootCellsPerThread = 2;
constexpr int kBootBatch = 6;    // nodes a thread has in flight: cap / threads at the default cap (every step of the chain is a global or LDS round trip)
// kBootPairs (multi-index classes = sub-space pairs): dpq_format.h

// V = 1 (round 4): the threshold from ONE histogram pass (block_kth_bound_u32) instead of the exact k-th key by a radix
// select -- the block is latency-bound (wavefronts wait 59 % of their cycles, LDS array 48 % busy, VALU a third:
// profiles/r04b_sq_bootstrap_counters.json), and the select was eleven of its barriers: 31.7 -> 25.6 us per launch.
// V = 0: the kernel of rounds 2 - 3, kept for A/B (BootArgs.variant).  Measured and not kept (scripts/experiments/
// r04_bootstrap_*.patch): six CONSECUTIVE nodes per thread over a list of the non-empty cells (one search + a window of
// cell starts instead of six searches: +3 us -- the entry loads of a wave-instruction then touch six times the cache lines),
// table fields four entries per thread (+2.6 us: same reason, the byte stores), the 64 nearest centroids ranked by counting
// with v_readlane instead of 21 shuffle stages (rank phase 6.4 K -> 9.1 K cycles), a cell's two bounds as one 8-byte load
// and the fields' minima fetched ahead of the threshold (nothing); the histogram filled WHILE the nodes are evaluated, over a
// range fixed by the block's first batch of keys, so that no key is stored and the block's LDS stops growing with the nodes
// it evaluates (top-1000: 65 KB = two blocks per CU -> 24.5 KB = four): top-100 22.5 -> 23.0 us and a looser bound,
// top-1000 + 1 %, top-512 - 2 % (profiles/r04b_boot_online_histogram_ab.txt): the blocks per CU were not what bounds it.
template <int M, int V>
// M = 8: four blocks per CU (40 KB of LDS each) = 8 wavefronts per SIMD: the register budget (SGPRs included:
// 800 per SIMD) must allow it; M = 16: three blocks (48 KB)
__global__ __launch_bounds__(kBootThreads, M <= 8 ? 8 : 6) void bootstrap_kernel(const BootArgs a) {
    constexpr int W = Cfg<M>::W;
    constexpr int TE = M * 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* keys = reinterpret_cast<uint32_t*>(smem);                                   // [cap] fp32 distance bits (see below)
    float* T = reinterpret_cast<float*>(smem + (size_t)a.cap * 4);                        // [M][256] exact tables
    uint32_t* cstart = reinterpret_cast<uint32_t*>(T + TE);                               // [kBootCells] first multi-index entry of the cell
    uint16_t* pre = reinterpret_cast<uint16_t*>(cstart + kBootCells);                     // [1024 + 4] node prefix of the round's cells, clamped to 65535
    uint8_t* ord = reinterpret_cast<uint8_t*>(pre + 1024 + 4);                            // [2 * kBootPairs][256] centroids by rank
    uint32_t* hist = reinterpret_cast<uint32_t*>(ord + 2 * kBootPairs * 256);             // [264] radix-select scratch
    uint32_t* bcast = hist + 264;                                                         // [2]
    uint32_t* wave_sum = bcast + 2;                                                       // [8]

    const int slot = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = a.slot_query ? a.slot_query[slot] : (slot < a.n_queries ? slot : -1);
    if (q < 0) {  // padding slot of the last query group: its filter-table fields reject everything
        if (a.qtab) write_filter_fields<M>(a.qtab, slot, nullptr, 0.0f, nullptr, 0u, nullptr, tid, kBootThreads);
        return;
    }
    if (a.stamps && tid == 0) a.stamps[(size_t)slot * 8 + 6] = __builtin_amdgcn_s_memrealtime();  // 100 MHz, chip-wide
    if (tid == 0) a.cand_count[(size_t)slot * kRegionStride] = 0;  // no carried winners: the scan meets every node again
    {
        const float4* src = reinterpret_cast<const float4*>(a.lut32 + (size_t)q * TE);
        for (int i = tid; i < TE / 4; i += kBootThreads) reinterpret_cast<float4*>(T)[i] = src[i];
    }
    __syncthreads();
    auto mark = [&](int i) {  // developer diagnostics: phase boundaries of block 0's first wavefront
        if (a.stamps && tid == 0) a.stamps[(size_t)slot * 8 + i] = __builtin_amdgcn_s_memtime();
    };
    mark(0);
    if (a.nbr) {
        // Visiting order of the centroids of the 2 * kBootPairs sub-spaces the classes are indexed by: wavefront s
        // serves sort slot s (sub-space `which` = s & 1 of pair s >> 1).  Sorting all 256 table entries of the 8
        // sub-spaces cost 9 us per block; the first rounds only ask for the ~14 best, which lie among the
        // neighbours of the nearest centroid: find it (wave arg-min), take ITS 64 nearest centroids (a table built at
        // dpq_set_codebook), sort those 64 exactly by the query's table row (one key per lane, 21 shuffle stages),
        // and let the other 192 follow in neighbour order.  Any order gives a valid threshold; this one gives
        // 11 % more candidates than the exact order (scripts/sim_bootstrap3.py) for a sixth of the work.
        const int sl = wave;  // kBootThreads / 64 == 2 * kBootPairs
        const int sub = 2 * (sl >> 1) * (M / 8) + (sl & 1);
        const float* row = T + sub * 256;
        uint32_t best = 0xffffffffu;  // (entry bits with the low 8 mantissa bits cleared) | centroid: unique keys
#pragma unroll
        for (int j = 0; j < 4; ++j)
            best = min(best, (__float_as_uint(row[lane + 64 * j]) & 0xffffff00u) | (uint32_t)(lane + 64 * j));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, off, 64));
        const uint8_t* nb = a.nbr + ((size_t)sl * 256 + (best & 0xffu)) * 256;  // its neighbours, nearest first
        uint32_t cand[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cand[j] = nb[lane + 64 * j];
        uint32_t key = (__float_as_uint(row[cand[0]]) & 0xffffff00u) | cand[0];
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const uint32_t other = (uint32_t)__shfl_xor((int)key, j, 64);
                const bool take_min = ((lane & k) == 0) == ((lane & j) == 0);
                key = take_min ? min(key, other) : max(key, other);
            }
        }
        ord[sl * 256 + lane] = (uint8_t)(key & 0xffu);
#pragma unroll
        for (int j = 1; j < 4; ++j) ord[sl * 256 + lane + 64 * j] = (uint8_t)cand[j];
    } else {
        // Exact order (no neighbour table): bitonic sorts of 256
        // keys, one position per thread and four sorts per thread (threads 0..255: sort slots 0..3, 256..511:
        // slots 4..7; slot s = sub-space `which` = s & 1 of pair s >> 1).  The order only steers which cells are
        // visited first (any order gives a valid threshold), so the low 8 mantissa bits make room for the
        // centroid index: keys are unique.  Partners closer than 64 are reached by wave shuffles, the three
        // stages with distance 64 / 128 through LDS.  (Counting ranks took 256 comparisons per thread: 13 us.)
        const int h = tid >> 8, i = tid & 255;
        uint32_t key[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int sl = 4 * h + r, sub = 2 * (sl >> 1) * (M / 8) + (sl & 1);
            key[r] = (__float_as_uint(T[sub * 256 + i]) & 0xffffff00u) | (uint32_t)i;  // entries >= 0 (or +inf): uint order
        }
        uint32_t* xch = keys;  // [4][512] exchange buffer (the key list is empty yet; cap >= 2048)
#pragma unroll
        for (int k = 2; k <= 256; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                uint32_t other[4];
                if (j >= 64) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) xch[r * kBootThreads + tid] = key[r];
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < 4; ++r) other[r] = xch[r * kBootThreads + (tid ^ j)];
                    __syncthreads();
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) other[r] = (uint32_t)__shfl_xor((int)key[r], j, 64);
                }
                const bool take_min = ((i & k) == 0) == ((i & j) == 0);  // ascending run and lower position, or neither
#pragma unroll
                for (int r = 0; r < 4; ++r) key[r] = take_min ? min(key[r], other[r]) : max(key[r], other[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ord[(4 * h + r) * 256 + i] = (uint8_t)(key[r] & 0xffu);
    }
    __syncthreads();
    mark(1);

    int have = 0;  // keys so far (block-uniform)
    int rounds = 0;
    uint32_t key_lo = 0xffffffffu, key_hi = 0u;  // of the keys this thread wrote (V & 1)
    // Every class walks its 65536 cells in the order w = 0, 1, ...: shell t = floor(sqrt(w)) (= max of the two
    // centroid ranks), position s = w - t^2 inside it; s <= t: ranks (i, j) = (t, s), else (s - t - 1, t).  A round
    // takes the next kBootCells / P cells of every class, class-interleaved (u = P (w - w0) + p), so that
    // a cut-off list keeps the best cells of all classes.
    const int P = a.n_classes;  // 4 or 1
    const int p_shift = P == 4 ? 2 : 0;
    int w0 = 0;
    while (w0 < 65536 && have < a.target) {
        const int n_w = min(kBootCells >> p_shift, 65536 - w0);
        const int n_cells = n_w << p_shift;
        uint32_t cnt[kBootCellsPerThread], first[kBootCellsPerThread], mine = 0;
#pragma unroll
        for (int r = 0; r < kBootCellsPerThread; ++r) {  // thread -> cells 2 tid, 2 tid + 1
            const int u = kBootCellsPerThread * tid + r;
            cnt[r] = 0;
            first[r] = 0;
            if (u < n_cells) {
                const int p = u & (P - 1), v = (u >> p_shift) + w0;
                int t = (int)sqrtf((float)v);
                t -= t * t > v ? 1 : 0;
                t += (t + 1) * (t + 1) <= v ? 1 : 0;
                const int sft = v - t * t;
                const int i = sft <= t ? t : sft - t - 1, j = sft <= t ? sft : t;
                const uint32_t c = (uint32_t)ord[(2 * p) * 256 + i] | ((uint32_t)ord[(2 * p + 1) * 256 + j] << 8);
                const uint32_t* cs = a.cell_start + (size_t)p * 65537 + c;
                first[r] = cs[0];
                cnt[r] = cs[1] - first[r];
            }
            mine += cnt[r];
        }
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        uint32_t base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kBootThreads / 64; ++w) {
            const uint32_t ws = wave_sum[w];
            base += w < wave ? ws : 0u;
            total += ws;
        }
        uint32_t run = base + incl - mine;
#pragma unroll
        for (int r = 0; r < kBootCellsPerThread; ++r) {
            const int u = kBootCellsPerThread * tid + r;
            // cells beyond n_cells hold `total` (the search never lands on them); what lies beyond 65535 nodes is
            // never taken (cap <= 16384), so the prefix saturates
            pre[u] = (uint16_t)min(run, 65535u);
            if (u < kBootCells) cstart[u] = first[r];
            run += cnt[r];
        }
        __syncthreads();
        mark(2);
        // Evaluation of the round's nodes, as many as the key list still takes; kBootBatch per thread go through
        // search -> entry loads -> distance together.  Only an UPPER BOUND of the k-th key has to leave this kernel,
        // so the distances are plain fp32 sums (relative error < 2^-21 against the exact fp64 sum) and the keys are
        // their 32 bits: no fp64 adds, no id loads, half the LDS per key; the bound is inflated below.
        const int take = min((int)total, a.cap - have);
        for (int v0 = 0; v0 < take; v0 += kBootThreads * kBootBatch) {
            uint32_t e[kBootBatch], code[kBootBatch][W];
#pragma unroll
            for (int r = 0; r < kBootBatch; ++r) {
                const int v = min(v0 + tid + r * kBootThreads, take - 1);
                int pos = 0;  // largest u with pre[u] <= v (pre[0] = 0)
#pragma unroll
                for (int step = 512; step > 0; step >>= 1)  // pre[u] = total > v for n_cells <= u < 1024: no guard needed
                    pos += (uint32_t)pre[pos + step] <= (uint32_t)v ? step : 0;
                e[r] = cstart[pos] + ((uint32_t)v - pre[pos]);
            }
#pragma unroll
            for (int r = 0; r < kBootBatch; ++r) {
#pragma unroll
                for (int w = 0; w < W; ++w) code[r][w] = a.mi_code[(size_t)e[r] * W + w];
            }
#pragma unroll
            for (int r = 0; r < kBootBatch; ++r) {
                const int v = v0 + tid + r * kBootThreads;
                float d = 0.0f;
#pragma unroll
                for (int m = 0; m < M; ++m) d += T[m * 256 + ((code[r][m >> 2] >> (8 * (m & 3))) & 0xffu)];
                if (v < take) {
                    keys[have + v] = __float_as_uint(d);
                    if constexpr (V & 1) {
                        key_lo = min(key_lo, __float_as_uint(d));
                        key_hi = max(key_hi, __float_as_uint(d));
                    }
                }
            }
        }
        have += take;
        ++rounds;
        w0 += n_w;
        __syncthreads();  // pre / cstart / wave_sum are rewritten by the next round
        mark(3);
        if (have >= a.cap) break;
    }
    if (a.stamps && tid == 0) a.stamps[(size_t)slot * 8 + 5] = (unsigned long long)rounds;
    uint64_t kth = ~0ull;  // fewer than k nodes in the whole multi-index: no threshold
    if (have >= a.top_k) {
        // k nodes have an fp32-summed distance <= d; their exact distances (fp64 sum rounded once; the plain scan:
        // fp32 sum in position order) are within 2^-20 of it, so d * (1 + 2^-19), rounded up, with the largest id
        // is >= the keys of k real nodes: a valid upper bound of the final k-th key.
        // (V & 1: the histogram lives where the rounds kept their cell lists -- cstart and pre, 5000 bytes)
        uint32_t dbits;
        if constexpr (V & 1)
            dbits = block_kth_bound_u32<10>(keys, have, a.top_k, key_lo, key_hi, cstart, hist, tid, kBootThreads);
        else
            dbits = (uint32_t)block_radix_select(keys, have, a.top_k, hist, bcast, tid, kBootThreads);
        const float bound = __double2float_ru((double)__uint_as_float(dbits) * (1.0 + 0x1p-19));
        kth = ((uint64_t)__float_as_uint(bound) << 32) | 0xffffffffull;
    }
    if (tid == 0) a.thr_key[slot] = kth;
    mark(4);
    if (a.qtab) {
        // The first filter level's table fields of this slot, straight from the tables in LDS (saves the
        // quantise launch between bootstrap and scan: it sat on the critical path of a pipelined batch).
        __shared__ float s_of[M];
        __shared__ float s_scale;
        __shared__ uint32_t s_bias;
        if (tid < M) {
            const FilterScale fs = filter_scale<M>(kth, a.lut_min, (size_t)q);
            s_of[tid] = filter_offset<M>(fs, lut_min_of(a.lut_min, (size_t)q, M, tid), tid);
            if (tid == 0) {
                s_scale = fs.s32;
                s_bias = fs.bias;
            }
        }
        __syncthreads();
        write_filter_fields<M>(a.qtab, slot, T, s_scale, s_of, s_bias, a.relabel, tid, kBootThreads);
    }
    if (a.stamps && tid == 0) a.stamps[(size_t)slot * 8 + 7] = __builtin_amdgcn_s_memrealtime();
}

size_t bootstrap_lds_bytes(int M, int cap) {
    return (size_t)cap * 4 + (size_t)M * 256 * 4 + (size_t)kBootCells * 4 + (1024 + 4) * 2 + 2 * kBootPairs * 256 + (264 + 2 + 8) * 4;
}

template <int V>
static hipError_t launch_bootstrap_variant(const BootArgs& a, int M, int n_slots, size_t lds, hipStream_t stream) {
    if (M == 8) {
        static std::atomic<bool> done[64] = {};
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&bootstrap_kernel<8, V>), 128 * 1024, done);  // + a few static words
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((bootstrap_kernel<8, V>), dim3((unsigned)n_slots), dim3(kBootThreads), lds, stream, a);
    } else if (M == 16) {
        static std::atomic<bool> done[64] = {};
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&bootstrap_kernel<16, V>), 128 * 1024, done);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((bootstrap_kernel<16, V>), dim3((unsigned)n_slots), dim3(kBootThreads), lds, stream, a);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_bootstrap(const BootArgs& a, int M, int n_slots, hipStream_t stream) {
    if (n_slots <= 0) return hipSuccess;
    if (a.cap < a.top_k || a.cap < 2048 || a.cap > 16384 || !a.cell_start || (a.n_classes != 1 && a.n_classes != kBootPairs))
        return hipErrorInvalidValue;  // the rank sort borrows 8 KB of the key list
    const size_t lds = bootstrap_lds_bytes(M, a.cap);
    return (a.variant & 1) ? launch_bootstrap_variant<1>(a, M, n_slots, lds, stream)
                           : launch_bootstrap_variant<0>(a, M, n_slots, lds, stream);
}

// ---------------------------------------------------------------------------
// 8e: merge of n_lists partial top-k lists per query.  grid = nq.
// Every partial list arrives sorted (ascending (distance bits, id) keys, padding rows last) and the lists are
// disjoint (shards), so a key's place in the merged list is the number of smaller keys: its position in its own
// list plus, per other list, a binary search.  One barrier; a bitonic sort of n_lists * k keys took 55 of them.
// List l, query q, rank r: ids[((l * nq + q) * row_stride) + r], dists likewise -- row_stride = top_k for two
// separate arrays, 2 * top_k for the packed [n_lists][nq][2k] tensor of the all-gather (dists = ids + top_k).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kSelectThreads) void merge_kernel(const int32_t* __restrict__ ids,
                                                                const float* __restrict__ dists, int n_lists, int nq,
                                                                int top_k, int row_stride, int32_t* __restrict__ out_ids,
                                                                float* __restrict__ out_dists) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* v = reinterpret_cast<uint64_t*>(smem);  // [n_lists][top_k]
    const int q = blockIdx.x, tid = threadIdx.x;
    const int n = n_lists * top_k;
    for (int i = tid; i < n; i += kSelectThreads) {
        const int l = i / top_k, r = i % top_k;
        const size_t o = ((size_t)l * nq + q) * row_stride + r;
        const int32_t id = ids[o];
        v[i] = id >= 0 ? make_key(dists[o], (uint32_t)id) : ~0ull;
    }
    for (int r = tid; r < top_k; r += kSelectThreads) {  // rows the lists cannot fill stay padding
        out_ids[(size_t)q * top_k + r] = -1;
        out_dists[(size_t)q * top_k + r] = INFINITY;
    }
    __syncthreads();
    for (int i = tid; i < n; i += kSelectThreads) {
        const uint64_t key = v[i];
        if (key == ~0ull) continue;
        const int own = i / top_k;
        int rank = i % top_k;
        for (int l = 0; l < n_lists && rank < top_k; ++l) {
            if (l == own) continue;
            const uint64_t* lst = v + (size_t)l * top_k;
            int lo = 0, hi = top_k;  // first position whose key is not below `key` (padding = ~0 sorts last)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (lst[mid] < key) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < top_k) {
            out_ids[(size_t)q * top_k + rank] = (int32_t)(key & 0xffffffffu);
            out_dists[(size_t)q * top_k + rank] = __uint_as_float((uint32_t)(key >> 32));
        }
    }
}

// ---------------------------------------------------------------------------
// 8f row 2: PQ encoding, PQTree::EncodePlain (pq_tree.cpp:215-237): per
// sub-space the nearest codeword in fp32 -- `diff = v - c; dist += diff * diff`
// (separately rounded multiply and add), strict `<` so the first minimum wins.
// grid = (ceil(n / 256), M), block = 256 threads = 256 vectors; the sub-space's
// codewords sit in LDS and are read as broadcasts.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void encode_pq_kernel(const float* __restrict__ vectors, int64_t n, int D,
                                                         const float* __restrict__ codebook, int M, int K, int Ds,
                                                         uint8_t* __restrict__ codes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* cw = reinterpret_cast<float*>(smem);       // [K][Ds]
    float* sv = cw + (size_t)K * Ds;                  // [Ds][256]: this thread's sub-vector, conflict-free
    const int m = blockIdx.y, tid = threadIdx.x;
    const int64_t v = (int64_t)blockIdx.x * 256 + tid;
    for (int i = tid; i < K * Ds; i += 256) cw[i] = codebook[(size_t)m * K * Ds + i];
    for (int d = 0; d < Ds; ++d) {
        const int col = m * Ds + d;
        sv[d * 256 + tid] = (v < n && col < D) ? vectors[(size_t)v * D + col] : 0.0f;  // short vectors are zero padded
    }
    __syncthreads();
    if (v >= n) return;
    float best = FLT_MAX;
    int best_k = 0;
    for (int k = 0; k < K; ++k) {
        float dist = 0.0f;
        for (int d = 0; d < Ds; ++d) {
            const float diff = __fsub_rn(sv[d * 256 + tid], cw[k * Ds + d]);
            dist = __fadd_rn(dist, __fmul_rn(diff, diff));
        }
        if (dist < best) {
            best = dist;
            best_k = k;
        }
    }
    codes[(size_t)v * M + m] = (uint8_t)best_k;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------

// hipFuncSetAttribute is per device; handles may live on several GPUs
static hipError_t ensure_dynamic_lds(const void* fn, size_t bytes, std::atomic<bool>* done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !done[dev].load(std::memory_order_acquire)) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) done[dev].store(true, std::memory_order_release);
    }
    return hipSuccess;
}

hipError_t launch_encode_pq(const float* d_vectors, int64_t n, int D, const float* d_codebook, int M, int K, int Ds,
                            uint8_t* d_codes, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const size_t lds = ((size_t)K * Ds + (size_t)Ds * 256) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static std::atomic<bool> done[64] = {};
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&encode_pq_kernel), 160 * 1024, done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(encode_pq_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)M), dim3(256), lds, stream,
                       d_vectors, n, D, d_codebook, M, K, Ds, d_codes);
    return hipGetLastError();
}

size_t scan_lds_bytes(int M) { return M <= 8 ? ScanLds<8>::kBytes : ScanLds<16>::kBytes; }

// level 0 (n_shared > 0): the block holds n_shared keys and stages the query's exact tables (M KB)
size_t select_lds_bytes(int M, int top_k, int n_shared) {
    size_t kp = 1;
    while (kp < (size_t)top_k) kp <<= 1;
    const size_t n_keys = n_shared > 0 ? (size_t)std::min(n_shared, kSortMax) : (size_t)select_list_keys(top_k);
    const size_t area = n_shared > 0 ? kp : (size_t)select_area_keys(top_k);
    const size_t select_bytes = area * 8 + 266 * 4, table_bytes = n_shared > 0 ? (size_t)M * 256 * 4 : 0;
    return n_keys * 8 + std::max(select_bytes, table_bytes) + 16;
}

hipError_t launch_lut_build(const float* d_codebook, const float* d_queries, int nq, int n_slots, int M, int K, int Ds,
                            float* d_lut32, float* d_lut_min, uint32_t* d_cand_count, uint32_t* d_overflow,
                            const uint8_t* d_relabel, float* d_lut_labels, uint32_t* d_tight_hist, hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    n_slots = std::max(n_slots, nq);
    hipLaunchKernelGGL(lut_build_kernel, dim3((unsigned)((n_slots + kLutQueries - 1) / kLutQueries), (unsigned)M),
                       dim3(256), 0, stream, d_codebook, d_queries, nq, n_slots, M, K, Ds, d_lut32, d_lut_min,
                       d_cand_count, d_overflow, d_relabel, d_relabel ? d_lut_labels : nullptr, d_tight_hist);
    return hipGetLastError();
}

hipError_t launch_decode_segments(const DeviceImage& img, const uint32_t* seg_list, int n_seg, uint32_t* out_id,
                                  uint32_t* out_code, hipStream_t stream) {
    if (n_seg <= 0) return hipSuccess;
    if (img.M == 8)
        hipLaunchKernelGGL(decode_segments_kernel<8>, dim3((unsigned)n_seg), dim3(64), 0, stream, img, seg_list,
                           out_id, out_code);
    else if (img.M == 16)
        hipLaunchKernelGGL(decode_segments_kernel<16>, dim3((unsigned)n_seg), dim3(64), 0, stream, img, seg_list,
                           out_id, out_code);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_decode_list(const DeviceImage& img, const uint32_t* seg_list, int n_seg, const uint8_t* relabel,
                              uint32_t* out_code, hipStream_t stream) {
    if (n_seg <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((n_seg + 3) / 4);
    if (img.M == 8)
        hipLaunchKernelGGL(decode_list_kernel<8>, dim3(grid), dim3(256), 0, stream, img, seg_list, n_seg, relabel, out_code);
    else if (img.M == 16)
        hipLaunchKernelGGL(decode_list_kernel<16>, dim3(grid), dim3(256), 0, stream, img, seg_list, n_seg, relabel, out_code);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

template <int M, bool PLAIN, bool STAMPS, bool TIGHT = false>
static hipError_t launch_scan_m(const ScanArgs& a, int n_slot_groups, int splits, hipStream_t stream) {
    static std::atomic<bool> done[64] = {};
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&scan_kernel<M, PLAIN, STAMPS, TIGHT>), scan_lds_bytes(M), done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((scan_kernel<M, PLAIN, STAMPS, TIGHT>), dim3((unsigned)splits, (unsigned)n_slot_groups),
                       dim3(kScanThreads), scan_lds_bytes(M), stream, a);
    return hipGetLastError();
}

// Q queries per pass (stream_kernel): ceil(n_slots / Q) passes over the launch's segment list; Q = the smallest of
// 1, 2, 4 that covers the batch (eight per pass were measured too: 64 KB of tables leave 8-16 wavefronts per CU, and
// the batched filter path answers eight queries in half the time).  The slots' region 1 counts must
// be zero before the launch.
template <int M, int Q>
static hipError_t launch_stream_mq(const ScanArgs& a, int n_slots, hipStream_t stream) {
    const size_t lds = (size_t)Q * M * 256 * sizeof(float);
    static std::atomic<bool> done[64] = {};
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&stream_kernel<M, Q>), lds, done);
    if (e != hipSuccess) return e;
    const int passes = (n_slots + Q - 1) / Q;
    constexpr int threads = stream_threads<Q>(), waves = threads / 64;
    // enough workgroups to fill 256 CUs at the occupancy the tables' LDS allows, never more wavefronts than segments
    const int per_cu = std::max(1, std::min(32 / waves, (int)(160 * 1024 / lds)));
    const int wgs = std::max(1, std::min(256 * per_cu / std::max(1, std::min(passes, 8)), (a.n_seg_pass + waves - 1) / waves));
    hipLaunchKernelGGL((stream_kernel<M, Q>), dim3((unsigned)wgs, (unsigned)passes), dim3(threads), lds, stream, a);
    return hipGetLastError();
}

int stream_queries_per_pass(int M, int n_slots) {
    (void)M;
    return n_slots <= 1 ? 1 : n_slots <= 2 ? 2 : 4;
}

hipError_t launch_stream(const ScanArgs& a, int n_slots, hipStream_t stream) {
    if (a.n_seg_pass <= 0 || n_slots <= 0) return hipSuccess;
    const int q = stream_queries_per_pass(a.img.M, n_slots);
    if (a.img.M == 8) {
        switch (q) {
            case 1: return launch_stream_mq<8, 1>(a, n_slots, stream);
            case 2: return launch_stream_mq<8, 2>(a, n_slots, stream);
            default: return launch_stream_mq<8, 4>(a, n_slots, stream);
        }
    }
    if (a.img.M == 16) {
        switch (q) {
            case 1: return launch_stream_mq<16, 1>(a, n_slots, stream);
            case 2: return launch_stream_mq<16, 2>(a, n_slots, stream);
            default: return launch_stream_mq<16, 4>(a, n_slots, stream);
        }
    }
    return hipErrorInvalidValue;
}

template <int Q>
static hipError_t launch_strand_q(const ScanArgs& a, int n_slots, hipStream_t stream) {
    const size_t lds = StrandLds<Q>::kBytes;
    static std::atomic<bool> done[64] = {};
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&strand_kernel<Q>), lds, done);
    if (e != hipSuccess) return e;
    const int passes = (n_slots + Q - 1) / Q;
    const int per_cu = std::max(1, std::min(8, (int)(160 * 1024 / lds)));
    const int wgs = std::max(1, std::min(256 * per_cu / std::max(1, std::min(passes, 8)), (a.n_seg_pass + 3) / 4));
    hipLaunchKernelGGL((strand_kernel<Q>), dim3((unsigned)wgs, (unsigned)passes), dim3(kStrandThreads), lds, stream, a);
    return hipGetLastError();
}

#ifndef DPQ_S1_MAX_WGS
#define DPQ_S1_MAX_WGS 256  // (timing experiments: fewer CUs)
#endif
int strand1_workgroups(int n_strips) { return std::max(1, std::min(DPQ_S1_MAX_WGS, n_strips)); }

// One query per pass: the bound-table kernel.  One workgroup of 16 wavefronts per CU; strips go to the workgroups
// first, so a short list still reaches every CU.
static hipError_t launch_strand1(const ScanArgs& a, int n_slots, hipStream_t stream) {
    static std::atomic<bool> done[64] = {};
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&strand1_kernel), S1Lds::kBytes, done);
    if (e != hipSuccess) return e;
    const int wgs = strand1_workgroups(a.n_seg_pass);
    hipLaunchKernelGGL(strand1_kernel, dim3((unsigned)wgs, (unsigned)n_slots), dim3(kS1Threads), S1Lds::kBytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_strand(const ScanArgs& a, int n_slots, hipStream_t stream) {
    if (a.n_seg_pass <= 0 || n_slots <= 0) return hipSuccess;
    if (a.img.M != 8 || !a.img.st_ckpt) return hipErrorInvalidValue;
    switch (stream_queries_per_pass(8, n_slots)) {
        case 1: return a.debug_pass == 3 ? launch_strand_q<1>(a, n_slots, stream) : launch_strand1(a, n_slots, stream);  // (>= 16: strand1 experiments)
        case 2: return launch_strand_q<2>(a, n_slots, stream);
        default: return launch_strand_q<4>(a, n_slots, stream);
    }
}

hipError_t launch_quantise(const ScanArgs& a, int n_slot_groups, hipStream_t stream) {
    if (n_slot_groups <= 0) return hipSuccess;
    if (!a.qtab) return hipErrorInvalidValue;
    if (a.img.M == 8)
        hipLaunchKernelGGL(quantise_kernel<8>, dim3((unsigned)(Cfg<8>::NG * 8), (unsigned)n_slot_groups), dim3(256), 0,
                           stream, a);
    else if (a.img.M == 16)
        hipLaunchKernelGGL(quantise_kernel<16>, dim3((unsigned)(Cfg<16>::NG * 16), (unsigned)n_slot_groups), dim3(256),
                           0, stream, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_scan(const ScanArgs& a, int n_slot_groups, int splits, hipStream_t stream) {
    if (a.n_seg_pass <= 0 || n_slot_groups <= 0) return hipSuccess;
    if (!a.qtab) return hipErrorInvalidValue;
    const bool plain = a.img.raw != nullptr;
    if (a.stamps) {  // diagnostic instantiation (M = 8 only)
        if (a.img.M != 8) return hipErrorInvalidValue;
        return plain ? launch_scan_m<8, true, true>(a, n_slot_groups, splits, stream)
                     : launch_scan_m<8, false, true>(a, n_slot_groups, splits, stream);
    }
    // in-scan threshold tightening: plain-code instantiations only (the decode's registers leave no room for it)
    const bool tight = plain && a.tight_hist != nullptr && a.tight_k > 0 && a.debug_pass == 0 && splits <= kTightSplits;
    if (a.img.M == 8) return !plain ? launch_scan_m<8, false, false>(a, n_slot_groups, splits, stream)
                             : tight ? launch_scan_m<8, true, false, true>(a, n_slot_groups, splits, stream)
                                     : launch_scan_m<8, true, false>(a, n_slot_groups, splits, stream);
    if (a.img.M == 16) return !plain ? launch_scan_m<16, false, false>(a, n_slot_groups, splits, stream)
                              : tight ? launch_scan_m<16, true, false, true>(a, n_slot_groups, splits, stream)
                                      : launch_scan_m<16, true, false>(a, n_slot_groups, splits, stream);
    return hipErrorInvalidValue;
}

size_t qtab_bytes_per_group(int M) { return M <= 8 ? ScanLds<8>::kTables : ScanLds<16>::kTables; }
int scan_stamp_count() { return kStCount; }

template <int M, int THREADS>
static hipError_t launch_select_m(const SelectArgs& a, int n_slots, hipStream_t stream) {
    static std::atomic<bool> done[64] = {};
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&select_kernel<M, THREADS>),
                                      64 * 1024, done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((select_kernel<M, THREADS>), dim3((unsigned)n_slots), dim3(THREADS),
                       select_lds_bytes(M, a.top_k, a.shared_id ? a.shared_n : 0), stream, a);
    return hipGetLastError();
}

hipError_t launch_select(const SelectArgs& a, int M, int n_slots, hipStream_t stream) {
    if (n_slots <= 0) return hipSuccess;
    const bool level0 = a.shared_id != nullptr;
    // block size: 256 threads for the thousand keys of a top-100 level; beyond top-512 a slot brings thousands of keys
    // and a 1024- or 2048-key final sort: 512 threads (the blocks per CU are set by the keys' LDS either way).  Measured
    // (scripts/gpu_select_threads.sh; M q/s at 256 / 512 / 1024 threads): top-300 4.33 / 4.07 / 3.64, top-512 3.40 / 3.24 /
    // 3.07, top-1000 2.28 / 2.32 / 2.10, M = 16 top-1000 1.27 / 1.32 / 1.23, top-2048 1.20 / 1.30 / 1.20.
    // A handful of slots (the one-query and stream calls): a block per slot leaves the chip empty and the block's own chain is
    // the call's -- 1024 threads: 4 K candidates of a one-query pass over 64 M codes 33.0 / 23.8 / 19.3 us at 256 / 512 / 1024.
    const int threads = a.threads > 0 ? a.threads : n_slots <= 16 ? 1024 : level0 ? 512 : a.top_k > 512 ? 512 : 256;
    if (M == 8) return threads >= 1024 ? launch_select_m<8, 1024>(a, n_slots, stream)
                     : threads >= 512 ? launch_select_m<8, 512>(a, n_slots, stream) : launch_select_m<8, 256>(a, n_slots, stream);
    if (M == 16) return threads >= 1024 ? launch_select_m<16, 1024>(a, n_slots, stream)
                      : threads >= 512 ? launch_select_m<16, 512>(a, n_slots, stream) : launch_select_m<16, 256>(a, n_slots, stream);
    return hipErrorInvalidValue;
}

hipError_t launch_merge(const int32_t* d_ids, const float* d_dists, int n_lists, int nq, int top_k, int row_stride,
                        int32_t* d_out_ids, float* d_out_dists, hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    const size_t lds = (size_t)n_lists * top_k * sizeof(uint64_t);
    if (lds > 128 * 1024 || row_stride < top_k) return hipErrorInvalidValue;
    static std::atomic<bool> done[64] = {};
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&merge_kernel), 128 * 1024, done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(merge_kernel, dim3((unsigned)nq), dim3(kSelectThreads), lds, stream, d_ids, d_dists, n_lists,
                       nq, top_k, row_stride, d_out_ids, d_out_dists);
    return hipGetLastError();
}

}  // namespace dpq
