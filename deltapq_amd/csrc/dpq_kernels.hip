// dpq_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the DeltaPQ
// query path.  Reference being replaced: the per-query loop of
// query_processing_scan_compressed_codes_opt_o_direct
// (/root/reference/deltapq_create_approx_tree.h:2805-2984; "h:" below).
//
//   lut_build_kernel   a3: T[m][k] = sum_d (c[m][k][d] - q[m*Ds+d])^2 with the
//                      reference's mixed fp32/fp64 arithmetic (h:2841-2849).
//   scan_m8_kernel     a5: delta decode + ADC + threshold filter.  One
//                      wavefront = one 64-node chunk per step; child codes are
//                      rebuilt from parent + packed deltas by pointer jumping
//                      over ds_bpermute; distances come from LDS table gathers
//                      (no MFMA: this is a lookup workload).  Each workgroup
//                      keeps the tables of QG queries in LDS and decodes every
//                      chunk once for all of them.
//   select_kernel      a6: exact fp64 re-evaluation of the surviving candidates,
//                      radix-select of the k-th smallest (distance, id) key,
//                      bitonic sort of the top-k.
//   merge_kernel       8e: merge of per-shard partial top-k lists.
//
// Top-k strategy (replaces the reference's sequential size-k max-heap,
// h:2851-2853, 2909-2914): a threshold cascade.  Level 0 scans a small sample
// of segments and keeps everything; every later level scans a ~rho x larger
// sample (the last one: everything) and keeps only nodes whose key
// (fp32 distance bits, id) is <= the k-th smallest key of the previous level,
// which is a true upper bound of the final k-th key because sample nodes are
// real nodes.  The filter inside the scan is an fp32 sum with a conservative
// relative slack; candidates inside the slack band are decided by the exact
// fp64 rule, so the candidate set is exactly {key <= threshold key}.
#include "dpq_kernels.h"

#include <cfloat>
#include <cmath>

namespace dpq {

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------

__device__ __forceinline__ uint32_t bperm(int src_lane, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}

// number of set bits of a wave-uniform 64-bit mask strictly below this lane
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m, uint32_t acc) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, acc));
}

// LUT image addressing: [group][qi/4][m][k(256)][qi%4]
__device__ __host__ __forceinline__ size_t lut_image_index(int qi, int m, int k, int M) {
    return ((size_t)((qi >> 2) * M + m) * 256 + (size_t)k) * 4 + (size_t)(qi & 3);
}

__device__ __forceinline__ uint64_t make_key(float d, uint32_t id) {
    return ((uint64_t)__float_as_uint(d) << 32) | (uint64_t)id;
}

// ---------------------------------------------------------------------------
// a3: LUT build.  grid = (nq_padded, M), block = 256 (one thread per centroid)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lut_build_kernel(const float* __restrict__ codebook,
                                                         const float* __restrict__ queries, int nq, int M, int K,
                                                         int Ds, int QG, float* __restrict__ lut) {
    const int q = blockIdx.x, m = blockIdx.y, k = threadIdx.x;
    const int group = q / QG, qi = q % QG;
    float acc = 0.0f;
    if (q < nq && k < K) {
        const float* c = codebook + ((size_t)m * K + k) * Ds;
        const float* qv = queries + (size_t)q * M * Ds + (size_t)m * Ds;
        for (int d = 0; d < Ds; ++d) {
            // h:2845-2846: `float += pow(float - float, 2)`
            const float diff = __fsub_rn(c[d], qv[d]);                 // fp32 subtract
            const double sq = __dmul_rn((double)diff, (double)diff);   // pow(.,2): exact in fp64
            acc = (float)__dadd_rn((double)acc, sq);                   // float += double
        }
    }
    lut[(size_t)group * QG * M * 256 + lut_image_index(qi, m, k, M)] = acc;
}

// ---------------------------------------------------------------------------
// a5: scan (M = 8, one code = 2 dwords)
// ---------------------------------------------------------------------------

// a7: the reference's decoder[256] (main:312-325) as byte-permute selectors.
// entry.x/.y: v_perm_b32 selectors that move the node's packed changed bytes to
// their positions 0..3 / 4..7 (0x0c = constant zero); entry.z/.w: byte masks of
// the changed positions.
__device__ __forceinline__ uint4 make_decode_entry(uint32_t b) {
    uint32_t sel[2] = {0, 0}, pm[2] = {0, 0};
    uint32_t rank = 0;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const bool set = (b >> m) & 1u;
        const uint32_t s = set ? rank : 0x0cu;
        sel[m >> 2] |= s << (8 * (m & 3));
        pm[m >> 2] |= (set ? 0xffu : 0u) << (8 * (m & 3));
        rank += set;
    }
    return make_uint4(sel[0], sel[1], pm[0], pm[1]);
}

template <int QG>
__global__ __launch_bounds__(kScanThreads) void scan_m8_kernel(const ScanArgs a) {
    constexpr int M = 8;
    constexpr int NG = QG / 4;  // float4 sub-groups
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4* lut4 = reinterpret_cast<float4*>(smem);                         // [NG][M][256]
    const float* lutf = reinterpret_cast<const float*>(smem);
    uint4* dtab = reinterpret_cast<uint4*>(smem + (size_t)QG * M * 256 * 4);  // [256]
    // candidate staging: survivors are appended with LDS atomics and flushed to
    // HBM once per workgroup (one global atomic per query), see the epilogue
    uint32_t* stg_count = reinterpret_cast<uint32_t*>(smem + (size_t)QG * M * 256 * 4 + 4096);  // [QG] (+pad)
    uint32_t* stg_id = stg_count + 16;                                                          // [QG][kStage]
    uint32_t* stg_code = stg_id + QG * kStage;                                                  // [QG][kStage][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int slot_group = blockIdx.y;
    const int group = a.group_list ? a.group_list[slot_group] : slot_group;

    {   // stage the QG tables (one contiguous 128 KB image) and the decode table
        const float4* src = reinterpret_cast<const float4*>(a.lut) + (size_t)group * (QG * M * 256 / 4);
        for (int i = tid; i < QG * M * 256 / 4; i += kScanThreads) lut4[i] = src[i];
        if (tid < 256) dtab[tid] = make_decode_entry((uint32_t)tid);
        if (tid < 16) stg_count[tid] = 0;
    }
    __syncthreads();

    const int slot0 = slot_group * QG;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const int cps = a.img.chunks_per_segment;
    const uint64_t* ckpt64 = reinterpret_cast<const uint64_t*>(a.img.seg_ckpt);

    for (int s = blockIdx.x * kScanWaves + wave; s < a.n_seg_pass; s += gridDim.x * kScanWaves) {
        const uint32_t seg = (uint32_t)__builtin_amdgcn_readfirstlane(a.seg_list ? (int)a.seg_list[s] : s);
        uint64_t doff = a.img.seg_delta_off[seg];
        // ancestor stack (vecs_stack, h:2858-2862) lives in lanes 0..7
        uint32_t stk_lo = 0, stk_hi = 0;
        if (lane < 8) {
            const uint64_t v = ckpt64[(size_t)seg * 8 + lane];
            stk_lo = (uint32_t)v;
            stk_hi = (uint32_t)(v >> 32);
        }
        for (int c = 0; c < cps; ++c) {
            const int64_t node = ((int64_t)seg * cps + c) * 64 + lane;  // local position
            // ---- load this node's depth nibble and mask (coalesced) ----
            const uint32_t nb = a.img.nib[node >> 1];
            const uint32_t d = (node & 1) ? (nb >> 4) : (nb & 15u);
            const uint32_t mk = a.img.mask[node];
            const uint32_t pc = __popc(mk);
            // ---- wave exclusive scan of pc (<= 8) by bit planes: no LDS traffic ----
            const uint64_t b0 = __ballot(pc & 1u), b1 = __ballot(pc & 2u), b2 = __ballot(pc & 4u),
                           b3 = __ballot(pc & 8u);
            uint32_t excl = mbcnt64(b3, 0);
            excl = mbcnt64(b2, excl << 1);
            excl = mbcnt64(b1, excl << 1);
            excl = mbcnt64(b0, excl << 1);
            const uint32_t total = (uint32_t)(__popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2) + 8 * __popcll(b3));
            // ---- fetch up to 8 changed bytes at byte granularity: 3 aligned dwords + funnel shift ----
            const uint8_t* dp = a.img.delta + doff + excl;
            const uintptr_t ua = reinterpret_cast<uintptr_t>(dp);
            const uint32_t* wp = reinterpret_cast<const uint32_t*>(ua & ~(uintptr_t)3);
            const uint32_t sh = (uint32_t)(ua & 3);
            const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2];
            const uint32_t raw_lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
            const uint32_t raw_hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
            doff += total;
            // ---- scatter the packed bytes to their positions (a7) ----
            const uint4 t = dtab[mk];
            uint32_t pv_lo = __builtin_amdgcn_perm(raw_hi, raw_lo, t.x);
            uint32_t pv_hi = __builtin_amdgcn_perm(raw_hi, raw_lo, t.y);
            uint32_t pm_lo = t.z, pm_hi = t.w;
            // ---- parent = nearest preceding node with depth-1 (h:2888: stack[depth-1]) ----
            uint64_t B[8];
#pragma unroll
            for (int D = 0; D < 8; ++D) B[D] = __ballot(d == (uint32_t)D);
            uint64_t selB = 0;
#pragma unroll
            for (int D = 1; D < 8; ++D) selB = (d == (uint32_t)D) ? B[D - 1] : selB;
            const uint64_t prev = selB & lt_mask;
            int P = prev ? 63 - __clzll((long long)prev) : -1;  // -1: parent precedes the chunk
            uint32_t td = d;                                     // depth of the top of my resolved chain
            // ---- pointer jumping: compose patches along the in-chunk ancestor chain (<= 7 links) ----
#pragma unroll
            for (int step = 0; step < 3; ++step) {
                const int src = P < 0 ? lane : P;
                const uint32_t q_pv_lo = bperm(src, pv_lo), q_pv_hi = bperm(src, pv_hi);
                const uint32_t q_pm_lo = bperm(src, pm_lo), q_pm_hi = bperm(src, pm_hi);
                const uint32_t q_ptd = bperm(src, ((uint32_t)(P & 0xff)) | (td << 8));
                if (P >= 0) {
                    pv_lo = (q_pv_lo & ~pm_lo) | pv_lo;
                    pv_hi = (q_pv_hi & ~pm_hi) | pv_hi;
                    pm_lo |= q_pm_lo;
                    pm_hi |= q_pm_hi;
                    const uint32_t pp = q_ptd & 0xffu;
                    P = pp == 0xffu ? -1 : (int)pp;
                    td = q_ptd >> 8;
                }
            }
            // ---- apply to the ancestor that precedes the chunk ----
            const int e = td > 0 ? (int)td - 1 : 0;
            const uint32_t anc_lo = bperm(e, stk_lo), anc_hi = bperm(e, stk_hi);
            const uint32_t code_lo = (anc_lo & ~pm_lo) | pv_lo;
            const uint32_t code_hi = (anc_hi & ~pm_hi) | pv_hi;
            // ---- carry the stack: stack[D] = code of the last node with depth D ----
            if (c + 1 < cps) {
                int srcl = -1;
#pragma unroll
                for (int D = 0; D < 8; ++D)
                    if (B[D]) srcl = lane == D ? 63 - __clzll((long long)B[D]) : srcl;
                const uint32_t n_lo = bperm(srcl < 0 ? lane : srcl, code_lo);
                const uint32_t n_hi = bperm(srcl < 0 ? lane : srcl, code_hi);
                if (srcl >= 0) {
                    stk_lo = n_lo;
                    stk_hi = n_hi;
                }
            }
            // ---- ADC: 8 LDS gathers per 4 queries, fp32 filter sums ----
            float acc[QG];
#pragma unroll
            for (int q = 0; q < QG; ++q) acc[q] = 0.0f;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const uint32_t byte = ((m < 4 ? code_lo : code_hi) >> (8 * (m & 3))) & 0xffu;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const float4 v = lut4[(g * M + m) * 256 + byte];
                    acc[4 * g + 0] += v.x;
                    acc[4 * g + 1] += v.y;
                    acc[4 * g + 2] += v.z;
                    acc[4 * g + 3] += v.w;
                }
            }
            // ---- threshold filter (replaces the heap test h:2909-2914) ----
            const bool valid = node < a.img.n_local;
            bool any = false;
#pragma unroll
            for (int q = 0; q < QG; ++q) any |= acc[q] <= a.thr_hi[slot0 + q];
            any &= valid;
            if (__any(any)) {
                const uint32_t id = a.img.id_base + (uint32_t)node;
#pragma unroll
                for (int q = 0; q < QG; ++q) {
                    if (valid && acc[q] <= a.thr_hi[slot0 + q]) {
                        bool take = true;
                        if (acc[q] >= a.thr_lo[slot0 + q]) {
                            // inside the slack band: decide with the exact rule
                            // (fp64 sum of the fp32 entries, rounded to fp32, then (dist, id) order)
                            double dsum = 0.0;
#pragma unroll
                            for (int m = 0; m < M; ++m) {
                                const uint32_t byte = ((m < 4 ? code_lo : code_hi) >> (8 * (m & 3))) & 0xffu;
                                dsum = __dadd_rn(dsum, (double)lutf[lut_image_index(q, m, (int)byte, M)]);
                            }
                            take = make_key((float)dsum, id) <= a.thr_key[slot0 + q];
                        }
                        if (take) {
                            const uint32_t li = atomicAdd(&stg_count[q], 1u);
                            if (li < (uint32_t)kStage) {
                                stg_id[q * kStage + li] = id;
                                stg_code[2 * (q * kStage + li)] = code_lo;
                                stg_code[2 * (q * kStage + li) + 1] = code_hi;
                            } else {  // staging full (level 0 keeps everything): straight to HBM
                                const uint32_t idx = atomicAdd(&a.cand_count[slot0 + q], 1u);
                                if (idx < (uint32_t)a.cap) {
                                    const size_t o = (size_t)(slot0 + q) * a.cap + idx;
                                    a.cand_id[o] = id;
                                    a.cand_code[2 * o] = code_lo;
                                    a.cand_code[2 * o + 1] = code_hi;
                                }
                            }
                        }
                    }
                }
            }
        }
    }

    // ---- epilogue: flush the staged candidates, wave w serves query w ----
    __syncthreads();
    for (int q = wave; q < QG; q += kScanWaves) {
        const uint32_t n = min(stg_count[q], (uint32_t)kStage);
        if (n == 0) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&a.cand_count[slot0 + q], n);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t idx = base + i;
            if (idx < (uint32_t)a.cap) {
                const size_t o = (size_t)(slot0 + q) * a.cap + idx;
                a.cand_id[o] = stg_id[q * kStage + i];
                a.cand_code[2 * o] = stg_code[2 * (q * kStage + i)];
                a.cand_code[2 * o + 1] = stg_code[2 * (q * kStage + i) + 1];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// a6: select.  grid = slots, block = kSelectThreads
// ---------------------------------------------------------------------------

__device__ __forceinline__ void block_bitonic_sort(uint64_t* v, int n_pow2, int tid, int nthreads) {
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n_pow2; i += nthreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t x = v[i], y = v[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) {
                        v[i] = y;
                        v[ixj] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// k-th smallest (1-based rank `rank`) of keys[0..n) by MSB-first 8-bit radix select.
__device__ uint64_t block_radix_select(const uint64_t* keys, int n, int rank, uint32_t* hist, uint32_t* bcast,
                                       int tid, int nthreads) {
    uint64_t prefix = 0;
    uint32_t rem = (uint32_t)rank;
    for (int pass = 7; pass >= 0; --pass) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const int shift = 8 * pass;
        for (int i = tid; i < n; i += nthreads) {
            const uint64_t key = keys[i];
            const bool match = pass == 7 ? true : ((key >> (shift + 8)) == (prefix >> (shift + 8)));
            if (match) atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        // inclusive scan of the 256 bins (Hillis-Steele in LDS)
        for (int off = 1; off < 256; off <<= 1) {
            uint32_t add = 0;
            if (tid < 256 && tid >= off) add = hist[tid - off];
            __syncthreads();
            if (tid < 256) hist[tid] += add;
            __syncthreads();
        }
        if (tid < 256) {
            const uint32_t cum = hist[tid];
            const uint32_t below = tid ? hist[tid - 1] : 0u;
            if (cum >= rem && below < rem) {
                bcast[0] = (uint32_t)tid;
                bcast[1] = rem - below;
            }
        }
        __syncthreads();
        prefix |= (uint64_t)bcast[0] << shift;
        rem = bcast[1];
        __syncthreads();
    }
    return prefix;
}

__device__ __forceinline__ int32_t report_id(uint32_t pos, int64_t n_total) {
    // h:2949, 2970: for even N the trailing node is pushed as i+1 == N, not N-1.
    if ((n_total & 1) == 0 && (int64_t)pos == n_total - 1) return (int32_t)n_total;
    return (int32_t)pos;
}

__global__ __launch_bounds__(kSelectThreads) void select_kernel(const SelectArgs a) {
    __shared__ uint64_t sel[kMaxTopK];        // winner keys
    __shared__ uint32_t win_id[kMaxTopK];     // winner entries, staged so they can be
    __shared__ uint32_t win_code[kMaxTopK * 2];  // compacted to the front in place
    __shared__ uint32_t hist[256];
    __shared__ uint32_t bcast[2];
    __shared__ uint32_t sel_count;
    const int slot = blockIdx.x;
    const int tid = threadIdx.x;
    const int q = a.slot_query ? a.slot_query[slot] : slot;
    if (q < 0) return;  // padding slot
    const uint32_t cnt = a.cand_count[slot];
    const int n = (int)min(cnt, (uint32_t)a.cap);
    uint64_t* keys = a.keys + (size_t)slot * a.cap;
    const int QG = a.M <= 8 ? 16 : 8;
    const float* lut = a.lut + (size_t)(q / QG) * QG * a.M * 256;
    const int qi = q % QG;
    const int W = a.M / 4;

    // exact distances of the candidates: fp64 sum of the M fp32 entries, rounded
    // to fp32 == the reference's incremental fp64 stack (h:2889-2907), see DESIGN.md
    for (int i = tid; i < n; i += kSelectThreads) {
        const size_t o = (size_t)slot * a.cap + i;
        double dsum = 0.0;
        for (int w = 0; w < W; ++w) {
            const uint32_t cw = a.cand_code[o * W + w];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int m = 4 * w + b;
                dsum = __dadd_rn(dsum, (double)lut[lut_image_index(qi, m, (int)((cw >> (8 * b)) & 0xffu), a.M)]);
            }
        }
        keys[i] = make_key((float)dsum, a.cand_id[o]);
    }
    if (tid == 0) {
        sel_count = 0;
        // candidates were dropped at some level: the final list may miss entries -> host reruns this query
        if (cnt > (uint32_t)a.cap) a.overflow[slot] = 1u;
    }
    __syncthreads();

    const int kk = min(a.top_k, n);
    uint64_t kth = ~0ull;
    if (kk > 0) kth = block_radix_select(keys, n, kk, hist, bcast, tid, kSelectThreads);

    // The k-th smallest key seen so far bounds the final k-th key from above
    // (the candidates are real nodes), so it is the next level's threshold.
    if (tid == 0) {
        if (n < a.top_k) {  // fewer than k nodes seen so far: keep everything
            a.thr_key[slot] = ~0ull;
            a.thr_hi[slot] = INFINITY;
            a.thr_lo[slot] = INFINITY;
        } else {
            const float t = __uint_as_float((uint32_t)(kth >> 32));
            a.thr_key[slot] = kth;
            // |fp32 filter sum - exact| <= 7 * 2^-24 * exact, far inside 2^-20
            a.thr_hi[slot] = t * (1.0f + 0x1p-20f);
            a.thr_lo[slot] = t * (1.0f - 0x1p-20f);
        }
    }

    // winners = the kk keys <= kth (keys are unique: the id is part of the key)
    int p2 = 1;
    while (p2 < kk) p2 <<= 1;
    if (a.final_pass)
        for (int i = tid; i < p2; i += kSelectThreads) sel[i] = ~0ull;
    __syncthreads();
    for (int i = tid; i < n; i += kSelectThreads) {
        const uint64_t key = keys[i];
        if (key <= kth && kk > 0) {
            const uint32_t pos = atomicAdd(&sel_count, 1u);
            if (pos < (uint32_t)kMaxTopK) {
                sel[pos] = key;
                if (!a.final_pass) {
                    const size_t o = (size_t)slot * a.cap + i;
                    win_id[pos] = a.cand_id[o];
                    for (int w = 0; w < W && w < 2; ++w) win_code[pos * 2 + w] = a.cand_code[o * W + w];
                }
            }
        }
    }
    __syncthreads();

    if (!a.final_pass) {
        // carry the winners: compact them to the front; the next level appends behind
        for (int i = tid; i < kk; i += kSelectThreads) {
            const size_t o = (size_t)slot * a.cap + i;
            a.cand_id[o] = win_id[i];
            for (int w = 0; w < W && w < 2; ++w) a.cand_code[o * W + w] = win_code[i * 2 + w];
        }
        if (tid == 0) a.cand_count[slot] = (uint32_t)kk;
        return;
    }

    block_bitonic_sort(sel, p2, tid, kSelectThreads);
    for (int r = tid; r < a.top_k; r += kSelectThreads) {
        const size_t o = (size_t)q * a.top_k + r;
        if (r < kk) {
            a.out_ids[o] = report_id((uint32_t)(sel[r] & 0xffffffffu), a.n_codes_total);
            a.out_dists[o] = __uint_as_float((uint32_t)(sel[r] >> 32));
        } else {
            a.out_ids[o] = -1;
            a.out_dists[o] = INFINITY;
        }
    }
}

// Level 0 keeps everything (+inf); padding slots of the last LUT group keep nothing.
__global__ void init_thresholds_kernel(uint64_t* thr_key, float* thr_hi, float* thr_lo, int n, int n_real) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const bool real = i < n_real;
        thr_key[i] = real ? ~0ull : 0ull;
        thr_hi[i] = real ? INFINITY : -1.0f;
        thr_lo[i] = real ? INFINITY : -1.0f;
    }
}

// ---------------------------------------------------------------------------
// 8e: merge of n_lists partial top-k lists per query.  grid = nq
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kSelectThreads) void merge_kernel(const int32_t* __restrict__ ids,
                                                                const float* __restrict__ dists, int n_lists, int nq,
                                                                int top_k, int32_t* __restrict__ out_ids,
                                                                float* __restrict__ out_dists) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* v = reinterpret_cast<uint64_t*>(smem);
    const int q = blockIdx.x, tid = threadIdx.x;
    const int n = n_lists * top_k;
    int p2 = 1;
    while (p2 < n) p2 <<= 1;
    for (int i = tid; i < p2; i += kSelectThreads) {
        uint64_t key = ~0ull;
        if (i < n) {
            const int l = i / top_k, r = i % top_k;
            const size_t o = ((size_t)l * nq + q) * top_k + r;
            const int32_t id = ids[o];
            if (id >= 0) key = make_key(dists[o], (uint32_t)id);
        }
        v[i] = key;
    }
    __syncthreads();
    block_bitonic_sort(v, p2, tid, kSelectThreads);
    for (int r = tid; r < top_k; r += kSelectThreads) {
        const uint64_t key = v[r];
        const size_t o = (size_t)q * top_k + r;
        if (key != ~0ull) {
            out_ids[o] = (int32_t)(key & 0xffffffffu);
            out_dists[o] = __uint_as_float((uint32_t)(key >> 32));
        } else {
            out_ids[o] = -1;
            out_dists[o] = INFINITY;
        }
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------

size_t scan_lds_bytes(int M) {
    const int QG = queries_per_group(M);
    return lut_group_floats(M) * sizeof(float) + 256 * sizeof(uint4) + 16 * sizeof(uint32_t) +
           (size_t)QG * kStage * (1 + M / 4) * sizeof(uint32_t);
}

hipError_t launch_lut_build(const float* d_codebook, const float* d_queries, int nq, int nq_padded, int M, int K,
                            int Ds, float* d_lut, hipStream_t stream) {
    dim3 grid((unsigned)nq_padded, (unsigned)M);
    hipLaunchKernelGGL(lut_build_kernel, grid, dim3(256), 0, stream, d_codebook, d_queries, nq, M, K, Ds,
                       queries_per_group(M), d_lut);
    return hipGetLastError();
}

hipError_t launch_scan(const ScanArgs& a, int n_slot_groups, int splits, hipStream_t stream) {
    if (a.img.M != 8) return hipErrorInvalidValue;
    const size_t lds = scan_lds_bytes(8);
    if (a.n_seg_pass <= 0 || n_slot_groups <= 0) return hipSuccess;
    {   // the attribute is per device and handles may live on several GPUs
        static bool done[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= 64 || !done[dev]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_m8_kernel<16>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) done[dev] = true;
        }
    }
    dim3 grid((unsigned)splits, (unsigned)n_slot_groups);
    hipLaunchKernelGGL(scan_m8_kernel<16>, grid, dim3(kScanThreads), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_select(const SelectArgs& a, int n_slots, hipStream_t stream) {
    if (n_slots <= 0) return hipSuccess;
    hipLaunchKernelGGL(select_kernel, dim3((unsigned)n_slots), dim3(kSelectThreads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_init_thresholds(uint64_t* thr_key, float* thr_hi, float* thr_lo, int n, int n_real,
                                  hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(init_thresholds_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, thr_key,
                       thr_hi, thr_lo, n, n_real);
    return hipGetLastError();
}

hipError_t launch_merge(const int32_t* d_ids, const float* d_dists, int n_lists, int nq, int top_k,
                        int32_t* d_out_ids, float* d_out_dists, hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    int n = n_lists * top_k, p2 = 1;
    while (p2 < n) p2 <<= 1;
    const size_t lds = (size_t)p2 * sizeof(uint64_t);
    if (lds > 128 * 1024) return hipErrorInvalidValue;
    {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&merge_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(merge_kernel, dim3((unsigned)nq), dim3(kSelectThreads), lds, stream, d_ids, d_dists, n_lists,
                       nq, top_k, d_out_ids, d_out_dists);
    return hipGetLastError();
}

}  // namespace dpq
