// dpq_format.h -- host side of the DTC format: parse, validate, transcode to the
// structure-of-arrays image the GPU scans, and serialise.  No HIP here.
//
// DTC ("delta tree compressed") is the byte stream written by the reference's
// qnodes_to_compressed_codes_opt (deltapq_create_approx_tree.h:1730-1845) and
// consumed by its scan (h:2866-2975).  See DESIGN.md "Data layout" for the SoA.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/deltapq_amd.h"

namespace dpq {

constexpr int kChunk = 64;  // nodes per wavefront step
constexpr int kBootPairs = 4;  // classes of the threshold-bootstrap multi-index (sub-space pairs)
constexpr int kRunLen = 64;                     // strand image: nodes a lane decodes one after the other
constexpr int kStripNodes = 64 * kRunLen;       // nodes per strip (64 lanes)
constexpr int kPhaseLen = 4;                    // steps whose changed bytes are staged in LDS together (at most 64 * 4 * 8 = 2 KB)
// chunks per independently decodable segment: 128-node segments balance the scan's wavefronts better than 256
// (0.431 vs 0.451 ms per 1000-query step at 1 M codes) for 0.56 B/node of checkpoints and offsets
constexpr int kDefaultChunksPerSegment = 2;

inline int mask_bytes_for(int M) { return M > 8 ? 2 : 1; }
inline int depth_field_mask(int M) { return M > 8 ? 15 : 7; }  // h:2883 uses & 7 for M <= 8
inline int levels_for(int M) { return M > 8 ? 16 : 8; }        // ancestor stack entries (h:2858-2864: M of them)

// One (depth, mask, changed bytes) record while walking the payload.
struct NodeRec {
    int depth;
    unsigned mask;
    const uint8_t* deltas;  // popcount(mask) bytes
    int n_diff;
    int64_t payload_begin;  // first payload byte attributed to this node (its depth byte if it owns one)
    int64_t payload_end;
};

// Sequential reader of a DTC payload; mirrors the control flow of h:2866-2975.
class DtcWalker {
public:
    DtcWalker(const uint8_t* payload, int64_t n_bytes, int64_t n_codes, int M);
    // Returns DPQ_OK and fills rec for node `pos()`; DPQ_ERR_FORMAT with a message on violation.
    int next(NodeRec* rec);
    int64_t pos() const { return i_; }
    bool done() const { return i_ >= n_; }
    int64_t offset() const { return off_; }
    const std::string& error() const { return err_; }

private:
    const uint8_t* p_;
    int64_t nb_, n_;
    int M_, mb_, dmask_;
    int64_t i_ = 0, off_ = 0;
    int pending_depth_ = -1;  // depth of node i+1 taken from the pair byte
    int max_depth_seen_ = 0;
    std::string err_;
};

struct SoA {
    int M = 8;
    int levels = 8;
    int mask_bytes = 1;
    int chunks_per_segment = kDefaultChunksPerSegment;
    int64_t n_codes_total = 0, n_bytes_total = 0;
    int64_t node_lo = 0, node_hi = 0;  // global DFS positions in this image
    int64_t n_segments = 0;            // local segments
    int64_t algorithmic_bytes = 0;
    int64_t n_diffs = 0;
    int max_depth = 0;
    // The tree topology of a chunk is resolved once, here, instead of by every wavefront that decodes it:
    std::vector<uint8_t> nib;             // 4 bits per local node (node 2j in the low nibble of byte j): the stack level the
                                          // node's in-chunk ancestor chain hangs from = depth(topmost in-chunk ancestor) - 1
    std::vector<uint8_t> par;             // per local node: lane (0..63) of its parent inside its 64-node chunk, 0xFF = the
                                          // parent precedes the chunk (it is stack[nib])
    std::vector<uint8_t> carry;           // [chunks][levels]: lane of the chunk's last node of depth D, 0xFF = none (the
                                          // stack entries a chunk hands to the next one of its segment)
    std::vector<uint8_t> mask;            // mask_bytes per local node (little endian)
    std::vector<uint8_t> delta;           // changed bytes, node order, ascending position; 16 bytes of tail padding
    std::vector<uint64_t> seg_delta_off;  // [n_segments + 1]
    std::vector<uint8_t> seg_ckpt;        // [n_segments][levels][M]: ancestor stack at the segment's first node
    // Threshold bootstrap (DESIGN.md "bootstrap"): inverted multi-indexes over every mi_stride-th node.  The
    // sampled nodes are dealt to kBootPairs classes; class p is indexed by cell = (code[s_p], code[s_p + 1]) of
    // its own sub-space pair.  A query evaluates the nodes of its best cells of every class exactly and takes
    // the k-th key as its first threshold -- as tight as a spread sample of a quarter of the index, and (four
    // different pairs) without the heavy tail one pair alone has.  Empty = not built.
    int mi_stride = 0;
    int mi_classes = 0;                   // kBootPairs, or 1 for shards of 64 K .. 256 K nodes (a class needs about one node per cell)
    std::vector<uint32_t> mi_cell_start;  // [mi_classes][65537] absolute position of every cell's first entry
    std::vector<uint32_t> mi_code;        // [entries][M / 4] decoded codes, cell-major
    std::vector<uint32_t> mi_id;          // [entries] global DFS position
    // Bank-aware relabelling of the centroids for the per-batch plain-code scratch: relabel[m * 256 + c]
    // = the label code value c of sub-space m carries THERE (a permutation of 0..255 per sub-space; empty = identity).
    // The scan reads a 16-byte table row per (sub-space, label): rows whose labels agree mod 16 share an LDS bank
    // quad, so values that often meet inside a 16-node read group are given different residues.
    std::vector<uint8_t> relabel;
    // ---- strand image (M = 8; built with multi_index_stride > 0): the SAME nodes laid out for the stream kernel, in
    // which a LANE decodes a run of kRunLen consecutive nodes one after the other -- the reference's own stack machine
    // (h:2876-2905), 64 of them side by side -- instead of a wavefront resolving a 64-node chunk cooperatively.
    // A strip = 64 runs = kStripNodes consecutive nodes of the shard; lane l of a strip owns nodes [l kRunLen, (l+1) kRunLen).
    // Everything a wavefront reads per step is lane-interleaved (coalesced); the changed bytes come in PHASES of kPhaseLen
    // steps: the bytes all 64 lanes need for those steps lie together (lane after lane) and are staged in LDS.
    int64_t n_strips = 0;
    std::vector<uint64_t> st_ckpt;   // [n_strips][levels][64]: the ancestor stack (one code per level) at a run's first node
    std::vector<uint32_t> st_mask;   // [n_strips][kRunLen / 4][64]: the diff masks of the lane's next four nodes, one byte each
    std::vector<uint16_t> st_depth;  // [n_strips][kRunLen / 4][64]: their depths, four bits each (1.5 B per node of headers: what the
                                     // DTC payload spends on a node's mask byte and depth nibble)
    std::vector<uint16_t> st_poff;   // [n_strips][kRunLen / kPhaseLen][64]: offset of the lane's bytes inside the phase (host only:
                                     // the kernels compute it as a wave prefix sum of the masks' popcounts)
    std::vector<uint32_t> st_pbase;  // [n_strips * phases + 1]: start of a phase in st_delta, in units of 16 bytes
    std::vector<uint8_t> st_delta;   // the phases, each padded to 16 bytes; 16 bytes of tail padding
    int64_t strand_bytes() const {
        return (int64_t)(st_ckpt.size() * 8 + st_mask.size() * 4 + st_depth.size() * 2 + st_pbase.size() * 4 + st_delta.size());
    }
    int64_t nodes_per_segment() const { return (int64_t)kChunk * chunks_per_segment; }
    int64_t device_bytes() const {
        return (int64_t)(nib.size() + par.size() + carry.size() + mask.size() + delta.size() + seg_delta_off.size() * 8 +
                         seg_ckpt.size());
    }
    int64_t bootstrap_bytes() const { return (int64_t)((mi_cell_start.size() + mi_code.size() + mi_id.size()) * 4); }
};

int validate(const uint8_t* payload, int64_t n_bytes, int64_t n_codes, int M, dpq_dtc_stats* stats, std::string* err);
// scan_codes > 0: image of the first scan_codes nodes only (the reference's `-N` below the header's n_codes,
// h:2825-2829); the stream is parsed with the header's n_codes (it decides which node owns a whole depth byte).
// multi_index_stride > 0: also build the bootstrap multi-index over every multi_index_stride-th local node, dealt to
// multi_index_classes classes (0 = chosen by bootstrap_classes_for).
// strands: build the strand image too (M = 8, with a multi-index): -1 = whenever the multi-index is built, 0 = never,
// 1 = yes.  dpq_open_* asks for it only where the stream pass will read it (shards from kStrandMinNodes, or forced).
int transcode(const uint8_t* payload, int64_t n_bytes, int64_t n_codes, int M, int shard_rank, int shard_count,
              int chunks_per_segment, SoA* out, std::string* err, int64_t scan_codes = 0, int multi_index_stride = 0,
              int multi_index_classes = 0, int strands = -1);
// The strand pass has 64 x fewer, 64 x longer work items than the chunk-per-wavefront decode: it is taken from this many
// codes per shard (dpq_capi.cpp: measured break-even), and the image is only built and uploaded for such shards.
constexpr int64_t kStrandMinNodes = (int64_t)8 << 20;
// The bootstrap multi-index of a list of (global position, code) pairs: counting sort by class and cell.
void build_multi_index(const std::vector<uint32_t>& ids, const std::vector<uint8_t>& codes, int M, int stride, int classes,
                       SoA* out);
// first sub-space of class p's pair: (0,1) (2,3) (4,5) (6,7) at M = 8; (0,1) (4,5) (8,9) (12,13) at M = 16
inline int bootstrap_pair_subspace(int M, int p) { return 2 * p * (M / 8); }
// Nodes below which the bootstrap is not worth its tables (the spread-sample cascade serves small indexes), the
// sampling stride that keeps the multi-index at <= 4 M entries, and the number of classes: four different sub-space
// pairs from 256 K sampled nodes (one node per cell and class), one class (pair 0/1) below.
constexpr int64_t kBootstrapMinNodes = 65536;
inline int bootstrap_stride_for(int64_t n_local) {
    if (n_local < kBootstrapMinNodes) return 0;
    return (int)((n_local + (1 << 22) - 1) >> 22);
}
inline int bootstrap_classes_for(int64_t n_sampled) { return n_sampled >= 4 * 65536 ? kBootPairs : 1; }
int encode(const uint8_t* root_code, const uint8_t* depths, const uint16_t* masks, const uint8_t* deltas,
           int64_t n_codes, int M, uint8_t* out, int64_t* n_bytes, std::string* err);

// loaders (a10)
int read_file(const std::string& path, std::vector<uint8_t>* out, std::string* err);
int read_dtc_header(const std::string& path, int64_t* n_codes, int64_t* n_bytes, std::string* err);
int read_codewords(const std::string& path, int* M, int* K, int* Ds, std::vector<float>* out, std::string* err);
int read_vecs(const std::string& path, bool is_bvecs, int64_t* n, int* D, std::vector<float>* out, int64_t cap,
              std::string* err);
std::string dtc_file_name(const std::string& dir, int M, int K, int64_t N);

}  // namespace dpq
