// dpq_format.cpp -- DTC parse / validate / transcode / serialise and the file
// loaders that feed the query path.  Host only.
#include "dpq_format.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace dpq {

static inline int popcnt(unsigned x) { return __builtin_popcount(x); }

DtcWalker::DtcWalker(const uint8_t* payload, int64_t n_bytes, int64_t n_codes, int M)
    : p_(payload), nb_(n_bytes), n_(n_codes), M_(M), mb_(mask_bytes_for(M)), dmask_(depth_field_mask(M)) {}

// Control flow of the reference scan: root (h:2866-2871), pair loop
// (h:2876-2948), trailing node with a whole depth byte (h:2949-2975).
int DtcWalker::next(NodeRec* rec) {
    char msg[160];
    if (i_ >= n_) {
        err_ = "walk past the last node";
        return DPQ_ERR_FORMAT;
    }
    if (i_ == 0) {
        if (nb_ < M_) {
            err_ = "payload shorter than the root code";
            return DPQ_ERR_FORMAT;
        }
        rec->depth = 0;
        rec->mask = (1u << M_) - 1u;
        rec->deltas = p_;
        rec->n_diff = M_;
        rec->payload_begin = 0;
        rec->payload_end = M_;
        off_ = M_;
        i_ = 1;
        return DPQ_OK;
    }
    int64_t begin = off_;
    int depth;
    if (pending_depth_ >= 0) {  // second node of a pair (h:2916)
        depth = pending_depth_;
        pending_depth_ = -1;
    } else {
        if (off_ >= nb_) {
            snprintf(msg, sizeof msg, "payload ends before the depth byte of node %lld", (long long)i_);
            err_ = msg;
            return DPQ_ERR_FORMAT;
        }
        int b = p_[off_++];
        if (i_ + 1 < n_) {  // pair byte (h:2879-2883)
            depth = b & dmask_;
            pending_depth_ = (b >> 4) & dmask_;
        } else {  // trailing node: the whole byte is the depth (h:2951)
            depth = b;
        }
    }
    if (depth < 1 || depth >= M_ || depth > max_depth_seen_ + 1) {
        snprintf(msg, sizeof msg, "node %lld: depth %d invalid (deepest so far %d, M %d)", (long long)i_, depth,
                 max_depth_seen_, M_);
        err_ = msg;
        return DPQ_ERR_FORMAT;
    }
    if (depth > max_depth_seen_) max_depth_seen_ = depth;
    if (off_ + mb_ > nb_) {
        snprintf(msg, sizeof msg, "payload ends before the mask of node %lld", (long long)i_);
        err_ = msg;
        return DPQ_ERR_FORMAT;
    }
    unsigned mask = p_[off_];
    if (mb_ == 2) mask |= (unsigned)p_[off_ + 1] << 8;
    off_ += mb_;
    if (M_ < 16 && (mask >> M_) != 0) {
        snprintf(msg, sizeof msg, "node %lld: mask 0x%x has bits beyond M=%d", (long long)i_, mask, M_);
        err_ = msg;
        return DPQ_ERR_FORMAT;
    }
    int nd = popcnt(mask);
    if (off_ + nd > nb_) {
        snprintf(msg, sizeof msg, "payload ends inside the changed bytes of node %lld", (long long)i_);
        err_ = msg;
        return DPQ_ERR_FORMAT;
    }
    rec->depth = depth;
    rec->mask = mask;
    rec->deltas = p_ + off_;
    rec->n_diff = nd;
    rec->payload_begin = begin;
    off_ += nd;
    rec->payload_end = off_;
    i_++;
    return DPQ_OK;
}

static int check_args(const uint8_t* payload, int64_t n_bytes, int64_t n_codes, int M, std::string* err) {
    if (!payload || n_bytes < 0 || n_codes < 1 || M < 1 || M > 16) {
        if (err) *err = "bad argument (payload NULL, n_codes < 1 or M outside 1..16)";
        return DPQ_ERR_ARG;
    }
    if (n_codes >= (int64_t)INT32_MAX) {  // h:1757-1761
        if (err) *err = "number of codes does not fit a 32-bit id";
        return DPQ_ERR_ARG;
    }
    // a header that promises more nodes than the payload can hold (root + per node a mask and half a depth byte)
    // is refused before anything is sized by it
    const int64_t least = (int64_t)M + (n_codes - 1) * mask_bytes_for(M) + n_codes / 2;
    if (n_bytes < least) {
        if (err) *err = "n_codes in the header needs more payload bytes than n_bytes";
        return DPQ_ERR_FORMAT;
    }
    return DPQ_OK;
}

int validate(const uint8_t* payload, int64_t n_bytes, int64_t n_codes, int M, dpq_dtc_stats* stats,
             std::string* err) {
    int rc = check_args(payload, n_bytes, n_codes, M, err);
    if (rc) return rc;
    dpq_dtc_stats st;
    memset(&st, 0, sizeof st);
    st.n_codes = n_codes;
    st.n_bytes = n_bytes;
    st.M = M;
    DtcWalker w(payload, n_bytes, n_codes, M);
    NodeRec r;
    while (!w.done()) {
        rc = w.next(&r);
        if (rc) {
            if (err) *err = w.error();
            return rc;
        }
        st.depth_hist[r.depth]++;
        if (r.depth > st.max_depth) st.max_depth = r.depth;
        if (w.pos() > 1) st.n_diffs += r.n_diff;
    }
    if (w.offset() != n_bytes) {
        if (err) {
            char msg[160];
            snprintf(msg, sizeof msg, "stream consumed %lld bytes but n_bytes is %lld", (long long)w.offset(),
                     (long long)n_bytes);
            *err = msg;
        }
        return DPQ_ERR_FORMAT;
    }
    if (stats) *stats = st;
    return DPQ_OK;
}

int transcode(const uint8_t* payload, int64_t n_bytes, int64_t n_codes, int M, int shard_rank, int shard_count,
              int chunks_per_segment, SoA* out, std::string* err, int64_t scan_codes, int multi_index_stride,
              int multi_index_classes, int strands) {
    int rc = check_args(payload, n_bytes, n_codes, M, err);
    if (rc) return rc;
    if (scan_codes < 0 || scan_codes > n_codes) {
        if (err) *err = "num_codes (prefix to scan) outside 0..n_codes";
        return DPQ_ERR_ARG;
    }
    // The prefix [0, n_scan) is what this image stands for; n_codes keeps deciding how the stream parses.
    const int64_t n_scan = scan_codes > 0 ? scan_codes : n_codes;
    const bool prefix = n_scan < n_codes;
    if (shard_count <= 0) shard_count = 1;
    if (chunks_per_segment <= 0) chunks_per_segment = kDefaultChunksPerSegment;
    if (shard_rank < 0 || shard_rank >= shard_count || chunks_per_segment > 1024) {
        if (err) *err = "bad shard_rank / shard_count / chunks_per_segment";
        return DPQ_ERR_ARG;
    }
    const int64_t S = (int64_t)kChunk * chunks_per_segment;
    const int64_t nseg_total = (n_scan + S - 1) / S;
    const int levels = levels_for(M);
    const int mb = mask_bytes_for(M);

    // Shard range: contiguous segments, balanced by payload bytes.
    int64_t seg_lo = 0, seg_hi = nseg_total;
    if (shard_count > 1) {
        std::vector<int64_t> seg_bytes((size_t)nseg_total, 0);
        DtcWalker w(payload, n_bytes, n_codes, M);
        NodeRec r;
        int64_t bytes_scanned = 0;
        while (!w.done() && w.pos() < n_scan) {
            int64_t i = w.pos();
            rc = w.next(&r);
            if (rc) {
                if (err) *err = w.error();
                return rc;
            }
            seg_bytes[(size_t)(i / S)] += r.payload_end - r.payload_begin;
            bytes_scanned += r.payload_end - r.payload_begin;
        }
        if (!prefix && w.offset() != n_bytes) {
            if (err) *err = "stream length does not match n_bytes";
            return DPQ_ERR_FORMAT;
        }
        // boundary b_r = first segment whose byte prefix reaches total * r / count
        auto boundary = [&](int r_) -> int64_t {
            if (r_ <= 0) return 0;
            if (r_ >= shard_count) return nseg_total;
            long double target = (long double)bytes_scanned * r_ / shard_count;
            int64_t acc = 0;
            for (int64_t s = 0; s < nseg_total; s++) {
                if ((long double)acc >= target) return s;
                acc += seg_bytes[(size_t)s];
            }
            return nseg_total;
        };
        seg_lo = boundary(shard_rank);
        seg_hi = boundary(shard_rank + 1);
        if (seg_hi < seg_lo) seg_hi = seg_lo;
    }

    SoA& o = *out;
    o = SoA();
    o.M = M;
    o.levels = levels;
    o.mask_bytes = mb;
    o.chunks_per_segment = chunks_per_segment;
    o.n_codes_total = n_scan;  // the even-N id rule (h:2949, 2970) applies to the codes scanned
    o.n_bytes_total = n_bytes;
    o.node_lo = std::min(seg_lo * S, n_scan);
    o.node_hi = std::min(seg_hi * S, n_scan);
    o.n_segments = seg_hi - seg_lo;
    const int64_t n_pad = o.n_segments * S;
    o.nib.assign((size_t)(n_pad / 2), 0x00);  // padding nodes: children of stack[0] with no change
    o.par.assign((size_t)n_pad, 0xFF);
    o.carry.assign((size_t)(n_pad / kChunk) * levels, 0xFF);
    o.mask.assign((size_t)(n_pad * mb), 0);
    o.seg_delta_off.assign((size_t)o.n_segments + 1, 0);
    o.seg_ckpt.assign((size_t)(o.n_segments * levels * M), 0);
    o.delta.reserve((size_t)std::min<int64_t>(n_bytes, (o.node_hi - o.node_lo) * (int64_t)M / 2 + 64));

    std::vector<uint8_t> stack((size_t)levels * M, 0);  // vecs_stack (h:2858-2862)
    int last_lane[16], top_level[64];                   // per chunk: last lane of every depth; per lane: its chain's stack level
    // co-occurrence of code values inside 16-node read groups (every group_step-th group): input of the relabelling
    const bool want_relabel = multi_index_stride > 0 && M % 4 == 0;
    const int64_t group_step = std::max<int64_t>(1, (o.node_hi - o.node_lo) / (16 * 65536));
    std::vector<uint32_t> cooc(want_relabel ? (size_t)M * 65536 : 0, 0);
    uint8_t grp[16][16];  // [sub-space][member]
    int grp_n = 0;
    std::vector<uint32_t> mi_ids;   // bootstrap sample: global position and decoded code of every stride-th node
    std::vector<uint8_t> mi_codes;
    if (multi_index_stride > 0) {
        mi_ids.reserve((size_t)((o.node_hi - o.node_lo) / multi_index_stride + 1));
        mi_codes.reserve((size_t)((o.node_hi - o.node_lo) / multi_index_stride + 1) * M);
    }
    // strand image (see dpq_format.h): filled strip by strip
    const bool want_strands = multi_index_stride > 0 && M == 8 && strands != 0;
    constexpr int kPhases = kRunLen / kPhaseLen;
    std::vector<uint8_t> strip_bytes;                      // changed bytes of the current strip, [lane][step][<= 8]
    std::vector<uint8_t> strip_cnt;                        // their counts
    if (want_strands) {
        o.n_strips = (o.node_hi - o.node_lo + kStripNodes - 1) / kStripNodes;
        o.st_ckpt.assign((size_t)o.n_strips * levels * 64, 0);
        // padding nodes: depth 1, mask 0 (a copy of stack[0]); the kernel never reports them
        o.st_mask.assign((size_t)o.n_strips * (kRunLen / 4) * 64, 0u);
        o.st_depth.assign((size_t)o.n_strips * (kRunLen / 4) * 64, 0x1111u);
        o.st_poff.assign((size_t)o.n_strips * kPhases * 64, 0);
        o.st_pbase.assign((size_t)o.n_strips * kPhases + 1, 0);
        strip_bytes.assign((size_t)kStripNodes * 8, 0);
        strip_cnt.assign((size_t)kStripNodes, 0);
    }
    auto flush_strip = [&](int64_t strip) {  // phases of the strip: for every phase the lanes' bytes, lane after lane
        for (int ph = 0; ph < kPhases; ++ph) {
            o.st_pbase[(size_t)(strip * kPhases + ph)] = (uint32_t)(o.st_delta.size() / 16);
            size_t used = 0;
            for (int lane = 0; lane < 64; ++lane) {
                o.st_poff[(size_t)((strip * kPhases + ph) * 64 + lane)] = (uint16_t)used;
                for (int st = ph * kPhaseLen; st < (ph + 1) * kPhaseLen; ++st) {
                    const size_t node = (size_t)lane * kRunLen + st;
                    o.st_delta.insert(o.st_delta.end(), &strip_bytes[node * 8], &strip_bytes[node * 8] + strip_cnt[node]);
                    used += strip_cnt[node];
                }
            }
            o.st_delta.resize((o.st_delta.size() + 15) / 16 * 16, 0);
        }
        std::fill(strip_cnt.begin(), strip_cnt.end(), 0);
    };
    DtcWalker w(payload, n_bytes, n_codes, M);
    NodeRec r;
    while (!w.done()) {
        const int64_t i = w.pos();
        if ((shard_count > 1 || prefix) && i >= o.node_hi) break;  // the rest was validated by the first pass / is not scanned
        if (want_strands && i >= o.node_lo && i < o.node_hi && (i - o.node_lo) % kRunLen == 0) {
            // a run starts: the ancestor stack as it stands, level-major and lane-interleaved
            const int64_t l = i - o.node_lo, strip = l / kStripNodes;
            const int lane = (int)((l % kStripNodes) / kRunLen);
            for (int lv = 0; lv < levels; ++lv) memcpy(&o.st_ckpt[(size_t)((strip * levels + lv) * 64 + lane)], &stack[(size_t)lv * M], 8);
        }
        if (i >= o.node_lo && i < o.node_hi && (i - o.node_lo) % S == 0) {
            const int64_t t = (i - o.node_lo) / S;
            memcpy(&o.seg_ckpt[(size_t)(t * levels * M)], stack.data(), (size_t)levels * M);
            o.seg_delta_off[(size_t)t] = o.delta.size();
        }
        rc = w.next(&r);
        if (rc) {
            if (err) *err = w.error();
            return rc;
        }
        // stack machine (h:2888, 2896-2900)
        uint8_t* cur = &stack[(size_t)r.depth * M];
        if (r.depth > 0) memcpy(cur, &stack[(size_t)(r.depth - 1) * M], (size_t)M);
        int j = 0;
        for (int m = 0; m < M; m++)
            if (r.mask & (1u << m)) cur[m] = r.deltas[j++];
        if (i >= o.node_lo && i < o.node_hi) {
            const int64_t l = i - o.node_lo;
            const int lane = (int)(l % kChunk);
            if (lane == 0)
                for (int d = 0; d < 16; ++d) last_lane[d] = -1;
            // parent = the latest node of depth - 1 (h:2888); inside the chunk it is a lane, before it a stack level
            const int pl = r.depth > 0 ? last_lane[r.depth - 1] : -1;
            const int level = pl >= 0 ? top_level[pl] : (r.depth > 0 ? r.depth - 1 : 0);
            top_level[lane] = level;
            last_lane[r.depth] = lane;
            o.par[(size_t)l] = pl >= 0 ? (uint8_t)pl : 0xFF;
            o.carry[(size_t)(l / kChunk) * levels + r.depth] = (uint8_t)lane;
            uint8_t& nb = o.nib[(size_t)(l >> 1)];
            if (l & 1)
                nb = (uint8_t)((nb & 0x0F) | (level << 4));
            else
                nb = (uint8_t)((nb & 0xF0) | level);
            o.mask[(size_t)(l * mb)] = (uint8_t)(r.mask & 0xFF);
            if (mb == 2) o.mask[(size_t)(l * mb + 1)] = (uint8_t)(r.mask >> 8);
            o.delta.insert(o.delta.end(), r.deltas, r.deltas + r.n_diff);
            if (want_strands) {
                const int64_t strip = l / kStripNodes, in_strip = l % kStripNodes;
                const int s_lane = (int)(in_strip / kRunLen), step = (int)(in_strip % kRunLen);
                // (the root, depth 0, arrives with mask 0xFF and its 8 bytes: every position "changes")
                const size_t hw = (size_t)((strip * (kRunLen / 4) + step / 4) * 64 + s_lane);
                o.st_mask[hw] |= (uint32_t)(r.mask & 0xFFu) << (8 * (step % 4));
                o.st_depth[hw] = (uint16_t)((o.st_depth[hw] & ~(0xFu << (4 * (step % 4)))) | ((unsigned)r.depth << (4 * (step % 4))));
                memcpy(&strip_bytes[(size_t)in_strip * 8], r.deltas, (size_t)r.n_diff);
                strip_cnt[(size_t)in_strip] = (uint8_t)r.n_diff;
                if (in_strip + 1 == kStripNodes || i + 1 == o.node_hi) flush_strip(strip);
            }
            o.algorithmic_bytes += r.payload_end - r.payload_begin;
            if (i > 0) o.n_diffs += r.n_diff;
            if (r.depth > o.max_depth) o.max_depth = r.depth;
            if (multi_index_stride > 0 && l % multi_index_stride == 0) {
                mi_ids.push_back((uint32_t)i);
                mi_codes.insert(mi_codes.end(), cur, cur + M);
            }
            if (want_relabel && (l / 16) % group_step == 0) {
                for (int m = 0; m < M; ++m) grp[m][grp_n] = cur[m];
                if (++grp_n == 16 || i + 1 == o.node_hi) {
                    for (int m = 0; m < M; ++m) {
                        uint8_t* v = grp[m];
                        std::sort(v, v + grp_n);
                        const int nd = (int)(std::unique(v, v + grp_n) - v);
                        uint32_t* Wm = &cooc[(size_t)m * 65536];
                        for (int x = 0; x < nd; ++x)
                            for (int y = x + 1; y < nd; ++y) {
                                Wm[(size_t)v[x] * 256 + v[y]]++;
                                Wm[(size_t)v[y] * 256 + v[x]]++;
                            }
                    }
                    grp_n = 0;
                }
            }
        }
    }
    if (want_relabel) {
        // balanced 16-colouring per sub-space: heaviest values first, each to the colour (bank-quad residue) where it
        // meets the least weight, at most 16 values per colour; label = colour + 16 * rank inside the colour
        o.relabel.assign((size_t)M * 256, 0);
        for (int m = 0; m < M; ++m) {
            const uint32_t* Wm = &cooc[(size_t)m * 65536];
            std::vector<uint64_t> weight(256, 0);
            for (int a = 0; a < 256; ++a)
                for (int b = 0; b < 256; ++b) weight[(size_t)a] += Wm[(size_t)a * 256 + b];
            std::vector<int> order(256);
            for (int a = 0; a < 256; ++a) order[(size_t)a] = a;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return weight[(size_t)a] > weight[(size_t)b]; });
            int load[16] = {0};
            std::vector<uint64_t> cost(256 * 16, 0);  // cost[value][colour]
            int colour[256];
            for (int a : order) {
                int best = -1;
                for (int c = 0; c < 16; ++c)
                    if (load[c] < 16 && (best < 0 || cost[(size_t)a * 16 + c] < cost[(size_t)a * 16 + best] ||
                                         (cost[(size_t)a * 16 + c] == cost[(size_t)a * 16 + best] && load[c] < load[best])))
                        best = c;
                colour[a] = best;
                load[best]++;
                for (int b = 0; b < 256; ++b) cost[(size_t)b * 16 + best] += Wm[(size_t)b * 256 + a];
            }
            int seen[16] = {0};
            for (int a = 0; a < 256; ++a) o.relabel[(size_t)m * 256 + a] = (uint8_t)(colour[a] + 16 * seen[colour[a]]++);
        }
    }
    if (multi_index_stride > 0)
        build_multi_index(mi_ids, mi_codes, M, multi_index_stride,
                          multi_index_classes > 0 ? multi_index_classes : bootstrap_classes_for((int64_t)mi_ids.size()), &o);
    if (w.done() && w.offset() != n_bytes) {
        if (err) {
            char msg[160];
            snprintf(msg, sizeof msg, "stream consumed %lld bytes but n_bytes is %lld", (long long)w.offset(),
                     (long long)n_bytes);
            *err = msg;
        }
        return DPQ_ERR_FORMAT;
    }
    o.seg_delta_off[(size_t)o.n_segments] = o.delta.size();
    o.delta.resize(o.delta.size() + 32, 0);  // lanes read up to 20 bytes past their first delta
    if (want_strands) {
        o.st_pbase[(size_t)o.n_strips * kPhases] = (uint32_t)(o.st_delta.size() / 16);
        o.st_delta.resize(o.st_delta.size() + 48, 0);  // a lane reads 32 bytes from its offset
        if (o.st_delta.size() / 16 > 0xffffffffull) {
            if (err) *err = "strand image beyond 64 GB of changed bytes";
            return DPQ_ERR_NOMEM;
        }
    }
    return DPQ_OK;
}

void build_multi_index(const std::vector<uint32_t>& ids, const std::vector<uint8_t>& codes, int M, int stride, int classes,
                       SoA* out) {
    // `classes` disjoint classes (sample j belongs to class j % classes); class p is indexed by the cell
    // (code[s], code[s + 1]) of ITS sub-space pair, s = bootstrap_pair_subspace(M, p).  Entries are class-major,
    // cell-major, DFS order inside a cell; cell_start holds absolute entry positions.
    const size_t n = ids.size(), W = (size_t)M / 4;
    const size_t P = (size_t)std::min(std::max(classes, 1), kBootPairs);
    out->mi_stride = stride;
    out->mi_classes = (int)P;
    out->mi_cell_start.assign(P * 65537, 0);
    out->mi_code.assign(n * W, 0);
    out->mi_id.assign(n, 0);
    auto cls = [&](size_t e) { return e % P; };
    auto cell_of = [&](size_t e) {
        const int s = bootstrap_pair_subspace(M, (int)cls(e));
        return (size_t)codes[e * M + s] | ((size_t)codes[e * M + s + 1] << 8);
    };
    for (size_t e = 0; e < n; ++e) out->mi_cell_start[cls(e) * 65537 + cell_of(e) + 1]++;
    uint32_t run = 0;
    for (size_t p = 0; p < P; ++p) {
        uint32_t* cs = &out->mi_cell_start[p * 65537];
        cs[0] = run;  // counts were stored at [cell + 1]
        for (size_t c = 0; c < 65536; ++c) cs[c + 1] += cs[c];
        run = cs[65536];
    }
    std::vector<uint32_t> fill(P * 65536);
    for (size_t p = 0; p < P; ++p)
        for (size_t c = 0; c < 65536; ++c) fill[p * 65536 + c] = out->mi_cell_start[p * 65537 + c];
    for (size_t e = 0; e < n; ++e) {  // stable: entries of a cell stay in DFS order
        const size_t pos = fill[cls(e) * 65536 + cell_of(e)]++;
        out->mi_id[pos] = ids[e];
        memcpy(&out->mi_code[pos * W], &codes[e * M], (size_t)M);
    }
}

// Writer: h:1765-1826 (M <= 8).  For M > 8 (this build's own extension, the
// reference format stops at M = 8): 2-byte little-endian masks, 4-bit depths.
int encode(const uint8_t* root_code, const uint8_t* depths, const uint16_t* masks, const uint8_t* deltas,
           int64_t n_codes, int M, uint8_t* out, int64_t* n_bytes, std::string* err) {
    if (!root_code || !depths || !masks || n_codes < 1 || M < 1 || M > 16 || !n_bytes) {
        if (err) *err = "bad argument";
        return DPQ_ERR_ARG;
    }
    const int mb = mask_bytes_for(M);
    int64_t nd = 0;
    for (int64_t i = 1; i < n_codes; i++) nd += popcnt(masks[i]);
    const int64_t total = M + nd + (n_codes - 1) * mb + (n_codes - 1 + 1) / 2;  // == h:1765 for M = 8
    *n_bytes = total;
    if (!out) return DPQ_OK;
    int64_t off = 0, doff = 0;
    for (int m = 0; m < M; m++) out[off++] = root_code[m];  // h:1771-1773
    auto put_node = [&](int64_t i) {
        out[off++] = (uint8_t)(masks[i] & 0xFF);
        if (mb == 2) out[off++] = (uint8_t)(masks[i] >> 8);
        int c = popcnt(masks[i]);
        for (int j = 0; j < c; j++) out[off++] = deltas[doff++];
    };
    int64_t i = 1;
    for (; i + 1 < n_codes; i += 2) {  // h:1776-1812
        if (depths[i] < 1 || depths[i] >= M || depths[i + 1] < 1 || depths[i + 1] >= M) {
            if (err) *err = "depth outside 1..M-1";
            return DPQ_ERR_ARG;
        }
        out[off++] = (uint8_t)(depths[i] | (depths[i + 1] << 4));
        put_node(i);
        put_node(i + 1);
    }
    if (i == n_codes - 1) {  // h:1813-1826
        if (depths[i] < 1 || depths[i] >= M) {
            if (err) *err = "depth outside 1..M-1";
            return DPQ_ERR_ARG;
        }
        out[off++] = depths[i];
        put_node(i);
    }
    if (off != total) {
        if (err) *err = "internal: encoded size mismatch";
        return DPQ_ERR_FORMAT;
    }
    return DPQ_OK;
}

// ---------------------------------------------------------------------------
// loaders
// ---------------------------------------------------------------------------

int read_file(const std::string& path, std::vector<uint8_t>* out, std::string* err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        if (err) *err = "cannot open file " + path;  // the reference prints this and goes on (h:2819-2821)
        return DPQ_ERR_IO;
    }
    long sz = -1;
    if (fseek(f, 0, SEEK_END) == 0) sz = ftell(f);
    if (sz < 0 || fseek(f, 0, SEEK_SET) != 0) {
        fclose(f);
        if (err) *err = "cannot determine the size of " + path;
        return DPQ_ERR_IO;
    }
    out->resize((size_t)sz);
    size_t got = sz ? fread(out->data(), 1, (size_t)sz, f) : 0;
    fclose(f);
    if ((long)got != sz) {
        if (err) *err = "short read on " + path;
        return DPQ_ERR_IO;
    }
    return DPQ_OK;
}

int read_dtc_header(const std::string& path, int64_t* n_codes, int64_t* n_bytes, std::string* err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        if (err) *err = "cannot open file " + path;
        return DPQ_ERR_IO;
    }
    int64_t h[2];
    size_t got = fread(h, sizeof(int64_t), 2, f);  // h:2823-2824
    fclose(f);
    if (got != 2) {
        if (err) *err = "short header in " + path;
        return DPQ_ERR_IO;
    }
    *n_codes = h[0];
    *n_bytes = h[1];
    return DPQ_OK;
}

// pq.cpp:288-312: `ifs >> M >> c >> Ks >> c >> Ds`, then per m `ifs >> v >> c`
// and Ks*Ds times `ifs >> float >> c`.
int read_codewords(const std::string& path, int* M, int* K, int* Ds, std::vector<float>* out, std::string* err) {
    std::ifstream ifs(path);
    if (!ifs.is_open()) {
        if (err) *err = "cannot open codewords file " + path;
        return DPQ_ERR_IO;
    }
    char c1, c2;
    int v;
    ifs >> *M >> c1 >> *K >> c2 >> *Ds;
    if (!ifs || *M <= 0 || *K <= 0 || *Ds <= 0 || (int64_t)*M * *K * *Ds > (1 << 28)) {
        if (err) *err = "bad codewords header in " + path;
        return DPQ_ERR_FORMAT;
    }
    if (!out) return DPQ_OK;
    out->assign((size_t)*M * *K * *Ds, 0.f);
    for (int m = 0; m < *M; ++m) {
        ifs >> v >> c1;
        if (!ifs || v != m) {
            if (err) *err = "bad sub-space label in " + path;
            return DPQ_ERR_FORMAT;
        }
        for (int ks = 0; ks < *K; ++ks)
            for (int ds = 0; ds < *Ds; ++ds) ifs >> (*out)[((size_t)m * *K + ks) * *Ds + ds] >> c1;
    }
    if (!ifs) {
        if (err) *err = "truncated codewords file " + path;
        return DPQ_ERR_FORMAT;
    }
    return DPQ_OK;
}

// utils.cpp:14-32 (fvecs: int32 D + D floats) and 46-71 (bvecs: int32 D + D bytes -> float).
int read_vecs(const std::string& path, bool is_bvecs, int64_t* n, int* D, std::vector<float>* out, int64_t cap,
              std::string* err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        if (err) *err = "cannot open vector file " + path;
        return DPQ_ERR_IO;
    }
    int64_t count = 0;
    int d = 0;
    std::vector<uint8_t> buf;
    if (out) out->clear();
    while (fread(&d, sizeof(int), 1, f) == 1) {
        if (d <= 0 || d > (1 << 20)) {
            fclose(f);
            if (err) *err = "bad dimension in " + path;
            return DPQ_ERR_FORMAT;
        }
        *D = d;
        const bool keep = out && count < cap;
        if (is_bvecs) {
            buf.resize((size_t)d);
            if (fread(buf.data(), 1, (size_t)d, f) != (size_t)d) break;
            if (keep)
                for (int k = 0; k < d; k++) out->push_back((float)buf[(size_t)k]);
        } else if (keep) {
            size_t base = out->size();
            out->resize(base + (size_t)d);
            if (fread(out->data() + base, sizeof(float), (size_t)d, f) != (size_t)d) {
                out->resize(base);
                break;
            }
        } else if (fseek(f, (long)sizeof(float) * d, SEEK_CUR) != 0) {
            break;
        }
        count++;
    }
    fclose(f);
    *n = count;
    return DPQ_OK;
}

std::string dtc_file_name(const std::string& dir, int M, int K, int64_t N) {  // h:2812-2814
    return dir + "/M" + std::to_string(M) + "K" + std::to_string(K) + "_Approx_compressed_codes_opt_N" +
           std::to_string(N);
}

}  // namespace dpq
