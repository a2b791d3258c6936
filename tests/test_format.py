"""CPU tests of the product's host code through the C-ABI: DTC validation,
serialisation, the SoA transcoder (checked by decoding the image the way the
GPU kernel is specified to), sharding, and the file loaders."""
import os

import numpy as np
import pytest

from conftest import make_case


def decode_soa(soa, M=8):
    """Decode a HostSoA image the way the scan kernel is specified to: per segment the ancestor stack starts from the
    checkpoint; inside a 64-node chunk a node's code is its parent's (lane `par` of the chunk, or stack[`nib`] when
    the parent precedes the chunk -- for a chain of in-chunk ancestors `nib` names the stack level the CHAIN hangs
    from) patched with the node's changed bytes; after a chunk stack[D] = code of lane carry[chunk][D]."""
    info = soa.info
    cps = info["chunks_per_segment"]
    n = info["node_hi"] - info["node_lo"]
    levels = 8 if M <= 8 else 16
    out = np.zeros((n, M), np.uint8)
    ck = soa.seg_ckpt.reshape(-1, levels, M)
    carry = soa.carry.reshape(-1, levels)
    for t in range(info["n_segments"]):
        stack = ck[t].copy()
        off = int(soa.seg_delta_off[t])
        for c in range(cps):
            chunk = t * cps + c
            codes = np.zeros((64, M), np.uint8)
            for lane in range(64):
                l = chunk * 64 + lane
                if l >= n:
                    break
                nb = soa.nib[l >> 1]
                level = (nb >> 4) if (l & 1) else (nb & 15)
                mk = int(soa.mask[l]) if M <= 8 else int(soa.mask[2 * l]) | (int(soa.mask[2 * l + 1]) << 8)
                p = int(soa.par[l])
                if p == 0xFF:
                    code = stack[level].copy()
                else:
                    assert p < lane
                    code = codes[p].copy()
                    # the chain of in-chunk ancestors ends on the same stack level for every node of the chain
                    pl = chunk * 64 + p
                    pnb = soa.nib[pl >> 1]
                    assert ((pnb >> 4) if (pl & 1) else (pnb & 15)) == level
                for m in range(M):
                    if (mk >> m) & 1:
                        code[m] = soa.delta[off]
                        off += 1
                codes[lane] = code
                out[l] = code
            for d in range(levels):
                if carry[chunk, d] != 0xFF:
                    stack[d] = codes[carry[chunk, d]]
        assert off == int(soa.seg_delta_off[t + 1])
    return out


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 256, 257, 5000])
@pytest.mark.parametrize("cps", [1, 4])
def test_transcode_is_lossless(lib, n, cps):
    from deltapq_amd import api, synth
    tree, payload, nb = make_case(n, seed=n)
    codes = synth.decode_tree_codes(tree)
    soa = api.HostSoA(payload, n, 8, chunks_per_segment=cps)
    assert soa.info["node_lo"] == 0 and soa.info["node_hi"] == n
    assert soa.info["algorithmic_bytes"] == nb == soa.info["n_bytes_total"]
    assert np.array_equal(decode_soa(soa), codes)
    # the image costs the DTC payload plus per-segment tables only
    tables = soa.seg_delta_off.nbytes + soa.seg_ckpt.nbytes
    pad = soa.info["n_segments"] * 64 * cps - n
    topology = soa.par.nbytes + soa.carry.nbytes                  # resolved tree topology: 1 B per node + 8 B per chunk
    assert soa.info["device_bytes"] - tables - topology <= nb + 1 + 1.5 * pad + 40


@pytest.mark.parametrize("n", [1, 2, 65, 1000, 1001])
def test_m16_extension_is_lossless(lib, oracle, n):
    """M = 16 has no reference format (h:1765, 1791-1795 stop at M = 8): this build's
    extension (2-byte masks, 4-bit depths) must still round-trip every code."""
    from deltapq_amd import api, synth
    tree = synth.synth_tree(n, 16, seed=n, mean_diffs=5.0)
    payload, nb = synth.encode_dtc(tree)
    codes = synth.decode_tree_codes(tree)
    assert np.array_equal(api.dtc_encode(tree["root"], tree["depths"], tree["masks"], tree["deltas"], 16), payload)
    st = api.dtc_validate(payload, n, 16)
    assert st["n_bytes"] == nb and st["max_depth"] <= 15
    soa = api.HostSoA(payload, n, 16, chunks_per_segment=2)
    assert np.array_equal(decode_soa(soa, 16), codes)
    lut = np.random.default_rng(0).random((16, 256)).astype(np.float32)
    _, _, alld, allc = oracle.scan_lut(payload, n, lut, 1, want_all=True)
    assert np.array_equal(allc, codes)
    s = sum(lut[m, codes[:, m]].astype(np.float64) for m in range(16)).astype(np.float32)
    assert np.array_equal(s.view(np.uint32), alld.view(np.uint32))


def test_c_encoder_matches_reference_layout(lib):
    from deltapq_amd import api, synth
    for n in (1, 2, 3, 10, 777, 778):
        tree, payload, nb = make_case(n, seed=3 * n)
        enc = api.dtc_encode(tree["root"], tree["depths"], tree["masks"], tree["deltas"])
        assert np.array_equal(enc, payload)
        n_diffs = int(synth.popcount16(tree["masks"][1:]).sum())
        assert nb == 8 + n_diffs + (3 * (n - 1) + 1) // 2          # h:1765
        st = api.dtc_validate(payload, n)
        assert st["n_diffs"] == n_diffs and st["n_bytes"] == nb
        assert st["depth_hist"][:8] == np.bincount(tree["depths"], minlength=8).tolist()


def test_shards_partition_the_index(lib):
    from deltapq_amd import api, dist, synth
    n = 20000
    tree, payload, nb = make_case(n, seed=77)
    codes = synth.decode_tree_codes(tree)
    for world in (2, 3, 8):
        ranges = dist.shard_ranges(payload, n, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        for a, b in zip(ranges[:-1], ranges[1:]):
            assert a[1] == b[0]                                     # contiguous, no gap, no overlap
        assert sum(r[2] for r in ranges) == nb                      # every payload byte owned once
        assert max(r[2] for r in ranges) < 1.2 * nb / world + 2000  # balanced by bytes
        for r in range(world):
            soa = api.HostSoA(payload, n, 8, shard_rank=r, shard_count=world)
            lo, hi = soa.info["node_lo"], soa.info["node_hi"]
            assert np.array_equal(decode_soa(soa), codes[lo:hi])    # shard decodes alone from its checkpoints


def test_more_shards_than_segments(lib):
    from deltapq_amd import dist
    tree, payload, nb = make_case(300, seed=4)                      # 2 segments of 256
    ranges = dist.shard_ranges(payload, 300, 8, chunks_per_segment=4)
    assert ranges[0][0] == 0 and ranges[-1][1] == 300
    assert sum(hi - lo for lo, hi, _ in ranges) == 300
    assert sum(1 for lo, hi, _ in ranges if hi > lo) <= 2           # the rest are empty shards


def test_malformed_streams_are_rejected(lib):
    from deltapq_amd import api
    tree, payload, nb = make_case(1001, seed=8)
    n = 1001
    api.dtc_validate(payload, n)
    cases = {}
    bad = payload.copy(); bad[8] = (bad[8] & 0xF0) | 0x00            # node 1 at depth 0
    cases["depth 0"] = (bad, n)
    bad = payload.copy(); bad[8] = (bad[8] & 0xF0) | 0x03            # node 1 deeper than root+1
    cases["depth jump"] = (bad, n)
    cases["truncated"] = (payload[:-3].copy(), n)
    cases["trailing garbage"] = (np.concatenate([payload, np.zeros(5, np.uint8)]), n)
    cases["wrong n_codes"] = (payload, n + 2)
    for name, (pl, nn) in cases.items():
        with pytest.raises(api.DpqError) as e:
            api.dtc_validate(pl, nn)
        assert e.value.status == -3, name                            # DPQ_ERR_FORMAT
        with pytest.raises(api.DpqError):
            api.HostSoA(pl, nn)
    with pytest.raises(api.DpqError) as e:
        api.dtc_validate(payload, 0)
    assert e.value.status == -1                                      # DPQ_ERR_ARG


def test_loaders_match_oracle_loaders(lib, oracle, tmp_path, codebook):
    """pq.cpp:288-312 and utils.cpp:14-110 through the product and through the oracle."""
    from deltapq_amd import api, synth
    d = str(tmp_path)
    tree, cb_rt, queries = synth.make_dataset_dir(d, 500, 7, seed=3, ext="fvecs")
    cw = os.path.join(d, "M8K256codewords.txt")
    a, b = api.read_codewords(cw), oracle.read_codewords(cw)
    assert a.shape == (8, 256, 16) and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(a, cb_rt)
    qa, qb = api.read_vecs(os.path.join(d, "query.fvecs"), "fvecs"), oracle.read_vecs(os.path.join(d, "query.fvecs"), "fvecs")
    assert np.array_equal(qa, qb) and np.array_equal(qa, queries)
    assert api.read_vecs(os.path.join(d, "query.fvecs"), "fvecs", top_n=3).shape == (3, 128)
    synth.write_bvecs(os.path.join(d, "query.bvecs"), queries.astype(np.uint8))
    qc = api.read_vecs(os.path.join(d, "query.bvecs"), "bvecs")
    assert np.array_equal(qc, oracle.read_vecs(os.path.join(d, "query.bvecs"), "bvecs"))
    assert np.array_equal(qc, queries)                               # integer-valued SIFT-like queries
    n_codes, payload = api.read_dtc_file(synth.dtc_file_name(d, 8, 256, 500))
    assert n_codes == 500 and np.array_equal(payload, synth.encode_dtc(tree)[0])
    with pytest.raises(api.DpqError) as e:
        api.read_codewords(os.path.join(d, "missing.txt"))
    assert e.value.status == -2                                      # DPQ_ERR_IO


def test_merge_topk_host(lib):
    from deltapq_amd import api
    rng = np.random.default_rng(0)
    nq, k, L = 5, 7, 3
    d = np.sort(rng.integers(0, 20, size=(L, nq, k)).astype(np.float32), axis=2)   # many ties across lists
    ids = rng.permutation(L * nq * k).reshape(L, nq, k).astype(np.int32)
    ids[2, :, 5:] = -1
    d[2, :, 5:] = np.inf
    mi, md = api.merge_topk_host(ids, d)
    for q in range(nq):
        cand = [(d[l, q, r], ids[l, q, r]) for l in range(L) for r in range(k) if ids[l, q, r] >= 0]
        cand.sort()
        assert [c[1] for c in cand[:k]] == mi[q].tolist()
        assert [c[0] for c in cand[:k]] == md[q].tolist()


@pytest.mark.parametrize("stride,shards,n", [(1, 1, 6000), (3, 1, 6000), (2, 3, 6000), (1, 1, 270000)])
def test_bootstrap_multi_index_layout(lib, stride, shards, n):
    """The threshold-bootstrap multi-indexes: every stride-th local node exactly once; sample j in class j % 4,
    filed under the cell (code[s], code[s + 1]) of its class's sub-space pair (s = 0, 2, 4, 6), global DFS
    position attached, DFS order inside a cell, absolute entry positions in cell_start."""
    from deltapq_amd import api, synth
    tree, payload, _ = make_case(n, seed=91)
    codes = synth.decode_tree_codes(tree)
    seen = []
    for r in range(shards):
        soa = api.HostSoA(payload, n, 8, shard_rank=r, shard_count=shards, multi_index_stride=stride)
        lo, hi = soa.info["node_lo"], soa.info["node_hi"]
        want = np.arange(lo, hi, stride)
        P = 4 if len(want) >= 4 * 65536 else 1       # four sub-space pairs need about one node per cell and class
        cs, ids, cd = soa.mi_cell_start.reshape(P, 65537).astype(np.int64), soa.mi_id, soa.mi_code.view(np.uint8).reshape(-1, 8)
        assert cs[0, 0] == 0 and cs[P - 1, -1] == len(ids) == len(want)
        assert np.array_equal(np.sort(ids), want) and np.array_equal(cd, codes[ids])
        for p in range(P):
            assert np.all(np.diff(cs[p]) >= 0) and (p == 0 or cs[p, 0] == cs[p - 1, -1])
            sl = slice(cs[p, 0], cs[p, -1])
            assert np.array_equal(np.sort(ids[sl]), want[p::P])              # class p = samples p, p + P, ...
            cell = cd[sl, 2 * p].astype(np.int64) | (cd[sl, 2 * p + 1].astype(np.int64) << 8)
            assert np.all(np.diff(cell) >= 0)                                  # cell-major
            assert np.array_equal(cs[p][cell] - cs[p, 0], np.searchsorted(cell, cell, side="left"))
            same = np.diff(cell) == 0
            assert np.all(np.diff(ids[sl].astype(np.int64))[same] > 0)        # DFS order inside a cell
        assert soa.info["bootstrap_stride"] == stride and soa.info["bootstrap_bytes"] == 4 * (P * 65537 + 3 * len(ids))
        seen.append(ids)
    assert len(np.unique(np.concatenate(seen))) == sum(len(s) for s in seen)


def decode_strands(soa, n_local):
    """The strand image as strand_kernel reads it: per strip 64 lanes, each running the reference's stack machine
    (h:2876-2905) over its 64 nodes -- four mask bytes to a dword and four depth nibbles to a halfword, the changed bytes of a
    four-step phase at st_pbase (16-byte units) + st_poff, ancestor stacks from st_ckpt [strip][level][lane]."""
    n_strips = len(soa.st_ckpt) // (8 * 64)
    out = np.zeros((n_strips * 4096, 8), dtype=np.uint8)
    masks = soa.st_mask.reshape(n_strips, 16, 64)
    depths = soa.st_depth.reshape(n_strips, 16, 64)
    ck = soa.st_ckpt.view(np.uint8).reshape(n_strips, 8, 64, 8)
    poff = soa.st_poff.reshape(n_strips, 16, 64)
    for s in range(n_strips):
        for lane in range(64):
            stack = [ck[s, lv, lane].copy() for lv in range(8)]
            for g in range(16):
                ptr = int(soa.st_pbase[s * 16 + g]) * 16 + int(poff[s, g, lane])
                for st in range(4):
                    mask, depth = int(masks[s, g, lane]) >> (8 * st) & 0xFF, int(depths[s, g, lane]) >> (4 * st) & 0xF
                    code = stack[max(depth, 1) - 1].copy()
                    for m in range(8):
                        if mask >> m & 1:
                            code[m] = soa.st_delta[ptr]
                            ptr += 1
                    stack[depth] = code
                    out[s * 4096 + lane * 64 + g * 4 + st] = code
                assert ptr <= int(soa.st_pbase[s * 16 + g + 1]) * 16                 # a lane stays inside its phase
    return out[:n_local]


@pytest.mark.parametrize("n,shards", [(1, 1), (2, 1), (4095, 1), (4096, 1), (4097, 1), (20000, 1), (30000, 3)])
def test_strand_image_is_lossless(lib, n, shards):
    """The stream pass's lane-per-run layout (built next to the multi-index, M = 8) decodes to the same codes as the
    stream itself, on whole shards and on shards cut by payload bytes; a phase never exceeds 2 KB; padding nodes are
    copies of stack level 0."""
    from deltapq_amd import api, synth
    tree, payload, nb = make_case(n, seed=n + 3)
    codes = synth.decode_tree_codes(tree)
    for r in range(shards):
        soa = api.HostSoA(payload, n, 8, shard_rank=r, shard_count=shards, multi_index_stride=1)
        lo, hi = soa.info["node_lo"], soa.info["node_hi"]
        n_strips = -(-(hi - lo) // 4096)
        assert len(soa.st_ckpt) == n_strips * 8 * 64 and len(soa.st_mask) == n_strips * 16 * 64 and len(soa.st_pbase) == n_strips * 16 + 1
        assert len(soa.st_depth) == n_strips * 16 * 64
        assert np.all(np.diff(soa.st_pbase.astype(np.int64)) * 16 <= 2048) and len(soa.st_delta) == int(soa.st_pbase[-1]) * 16 + 48
        assert np.array_equal(decode_strands(soa, hi - lo), codes[lo:hi])
    # without the multi-index (small shards, bootstrap off) there is no strand image
    assert len(api.HostSoA(payload, n, 8).st_ckpt) == 0


@pytest.mark.parametrize("n_scan", [1, 2, 63, 64, 65, 999, 1000, 2500, 2501])
def test_prefix_transcode(lib, n_scan):
    """`-N` below the header's n_codes (h:2825-2829): the image holds the first n_scan nodes of the stream."""
    from deltapq_amd import api, synth
    n = 2501
    tree, payload, _ = make_case(n, seed=17)
    codes = synth.decode_tree_codes(tree)
    for shards in (1, 2):
        got = []
        for r in range(shards):
            soa = api.HostSoA(payload, n, 8, shard_rank=r, shard_count=shards, num_codes=n_scan)
            assert soa.info["n_codes_total"] == n_scan
            got.append((soa.info["node_lo"], soa.info["node_hi"], decode_soa(soa)))
        assert got[0][0] == 0 and got[-1][1] == n_scan
        assert np.array_equal(np.concatenate([g[2] for g in got]), codes[:n_scan])
    with pytest.raises(api.DpqError):
        api.HostSoA(payload, n, 8, num_codes=n + 1)


def test_header_promising_more_nodes_than_bytes_is_refused(lib):
    """A corrupt header must not size anything: n_codes far beyond what n_bytes can hold."""
    from deltapq_amd import api
    tree, payload, _ = make_case(100, seed=3)
    with pytest.raises(api.DpqError) as e:
        api.HostSoA(payload, 2_000_000_000, 8)
    assert e.value.status in (-3, -1)


def test_codes_plain_other_record_layouts(lib, tmp_path):
    """PQTree::Read (pq_tree.cpp:1032-1081) also knows two-byte codes (K > 256) and (code, int id) records (with_id)."""
    from deltapq_amd import api
    rng = np.random.default_rng(4)
    n, M = 1000, 8
    codes = rng.integers(0, 256, size=(n, M), dtype=np.uint8)
    ids = rng.integers(0, 1 << 30, size=n, dtype=np.int32)
    p1 = str(tmp_path / "with_id")
    with open(p1, "wb") as f:
        f.write(np.int64(n).tobytes())
        rec = np.zeros((n, M + 4), dtype=np.uint8)
        rec[:, :M] = codes
        rec[:, M:] = ids.view(np.uint8).reshape(n, 4)
        f.write(rec.tobytes())
    c, i = api.read_codes_plain_ex(p1, M, 256, with_id=True)
    assert np.array_equal(c, codes) and np.array_equal(i, ids)
    wide = rng.integers(0, 1024, size=(n, M), dtype=np.uint16)
    p2 = str(tmp_path / "k1024")
    with open(p2, "wb") as f:
        f.write(np.int64(n).tobytes() + wide.tobytes())          # PQTree::Write, pq_tree.cpp:1025-1026
    c, i = api.read_codes_plain_ex(p2, M, 1024)
    assert i is None and c.dtype == np.uint16 and np.array_equal(c, wide)
    with pytest.raises(api.DpqError):
        api.read_codes_plain_ex(p2, M, 1024, with_id=True)


def _perm(src0, src1, sel):
    """v_perm_b32: byte i of the result = byte sel_i of {src0 (4..7), src1 (0..3)}; 0x0c = zero."""
    pool = [(src1 >> (8 * j)) & 0xFF for j in range(4)] + [(src0 >> (8 * j)) & 0xFF for j in range(4)]
    out = 0
    for i in range(4):
        s = (sel >> (8 * i)) & 0xFF
        out |= (0 if s == 0x0C else pool[s]) << (8 * i)
    return out


def _nibble_sel(nib):
    sel, rank = 0, 0
    for i in range(4):
        sel |= (rank if nib >> i & 1 else 4 + i) << (8 * i)
        rank += nib >> i & 1
    return sel


def decode_strands_like_strand1(soa, n_local):
    """strand1_kernel's decode, instruction for instruction: a lane's byte offset in a phase = sum of the earlier lanes'
    popcounts, each node's (up to) eight bytes read from ITS first byte on, low half = perm(parent.x, raw.x, sel[mask & 15]),
    high half = perm(parent.y, raw >> 8 popc(mask & 15), sel[mask >> 4]), stack[depth] = code."""
    n_strips = len(soa.st_ckpt) // (8 * 64)
    out = np.zeros((n_strips * 4096, 8), dtype=np.uint8)
    masks = soa.st_mask.reshape(n_strips, 16, 64)
    depths = soa.st_depth.reshape(n_strips, 16, 64)
    ck = soa.st_ckpt.reshape(n_strips, 8, 64)
    delta = np.concatenate([soa.st_delta, np.zeros(16, np.uint8)])
    for s in range(n_strips):
        stack = [[int(ck[s, lv, lane]) for lane in range(64)] for lv in range(8)]
        stack.insert(0, [0xDEADBEEFCAFEF00D] * 64)            # row -1: whatever the LDS holds in front of the stack
        for g in range(16):
            base = int(soa.st_pbase[s * 16 + g]) * 16
            mine = [bin(int(masks[s, g, lane])).count("1") for lane in range(64)]
            off = np.concatenate([[0], np.cumsum(mine)[:-1]])
            for lane in range(64):
                mk, dp = int(masks[s, g, lane]), int(depths[s, g, lane])
                for st in range(4):
                    o = base + int(off[lane]) + bin(mk & ((1 << (8 * st)) - 1)).count("1")
                    raw = int.from_bytes(delta[o:o + 8].tobytes(), "little")
                    lo, hi, depth = mk >> (8 * st) & 15, mk >> (8 * st + 4) & 15, dp >> (4 * st) & 15
                    parent = stack[depth][lane]               # row depth - 1 of the real stack
                    raw_hi = (raw >> (8 * bin(lo).count("1"))) & 0xFFFFFFFF
                    c0 = _perm(parent & 0xFFFFFFFF, raw & 0xFFFFFFFF, _nibble_sel(lo))
                    c1 = _perm(parent >> 32, raw_hi, _nibble_sel(hi))
                    stack[depth + 1][lane] = c0 | c1 << 32
                    out[s * 4096 + lane * 64 + g * 4 + st] = np.frombuffer((c0 | c1 << 32).to_bytes(8, "little"), np.uint8)
    return out[:n_local]


@pytest.mark.parametrize("n", [1, 4097, 20000])
def test_strand1_decode_rules(lib, n):
    """The one-query strand pass reads a node's bytes with one unaligned 8-byte load at the node's own offset and
    scatters them with nibble selectors: the same codes as the reference's stack machine."""
    from deltapq_amd import api, synth
    tree, payload, nb = make_case(n, seed=n + 5)
    codes = synth.decode_tree_codes(tree)
    soa = api.HostSoA(payload, n, 8, multi_index_stride=1)
    assert np.array_equal(decode_strands_like_strand1(soa, n), codes)


def test_strand1_bound_table_address_map():
    """strand1_kernel's bound table in LDS (dpq_kernels.hip, DPQ_S1_U8): the eight entries of code value c as two words
    (sub-spaces 0..3, 4..7), each stored once per bank -- word (c, half, bank) at c << 8 | half << 7 | bank << 2.  The fill
    loop and the read address (one v_perm_b32: code byte to bits 8..15, (lane mod 32) << 2 below it, sub-space in the
    ds_read_u8 offset field) restated in numpy: every lane reads entry[m][c] and the 32 lanes of a pass hit 32 banks."""
    rng = np.random.default_rng(5)
    entry = rng.integers(0, 256, size=(8, 256), dtype=np.uint8)  # [m][c]
    lds = np.zeros(256 * 64 * 4, dtype=np.uint8)
    # the fill: word w of the table = (c = w >> 6, half = (w >> 5) & 1, bank = w & 31)
    w = np.arange(256 * 64)
    c, half = w >> 6, (w >> 5) & 1
    for b in range(4):
        lds[4 * w + b] = entry[4 * half + b, c]
    assert lds.size == 64 * 1024
    lane = np.arange(64)
    lane4 = (lane & 31) << 2
    codes = rng.integers(0, 256, size=(64, 8), dtype=np.uint8)  # a code per lane
    for m in range(8):
        addr = (codes[:, m].astype(np.uint32) << 8) | lane4  # the v_perm_b32
        addr = addr + (((m >> 2) << 7) | (m & 3))             # the instruction's offset field
        assert np.array_equal(lds[addr], entry[m, codes[:, m]])
        banks = (addr >> 2) & 31                              # 4-byte banks; a 4-byte-class DS access serves 32 lanes per pass
        assert np.array_equal(banks[:32], lane[:32] & 31) and len(set(banks[:32])) == 32
        assert np.array_equal(banks[32:], lane[32:] & 31) and len(set(banks[32:])) == 32


def test_strand1_pair_window_extraction():
    """strand1_kernel (DPQ_S1_PAIR = 2): the second node of a pair takes bytes [p, p + 8) of the first node's 12-byte window
    -- a dword select on p >= 4 / p >= 8 and two v_alignbyte_b32 -- instead of a load of its own.  The rule restated in numpy
    for every p in 0..8: what it yields equals the window's bytes from p on (zeros behind the twelfth), so whenever the pair's
    bytes fit the window (p + the second node's count <= 12) the node reads exactly the bytes at its own offset."""
    rng = np.random.default_rng(11)

    def alignbyte(hi, lo, s):
        return (((hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)) >> np.uint64(8 * s)).astype(np.uint32)

    win = rng.integers(0, 256, size=(1000, 12), dtype=np.uint8)
    w = win.view("<u4")  # [1000][3]
    w0, w1, w2 = w[:, 0], w[:, 1], w[:, 2]
    zero = np.zeros_like(w0)
    padded = np.concatenate([win, np.zeros((1000, 8), np.uint8)], axis=1)
    for p in range(9):
        s1, s2 = p >= 4, p >= 8
        d0 = w2 if s2 else w1 if s1 else w0
        d1 = zero if s2 else w2 if s1 else w1
        d2 = zero if s1 else w2
        lo, hi = alignbyte(d1, d0, p & 3), alignbyte(d2, d1, p & 3)
        got = np.stack([lo, hi], axis=1).view(np.uint8)
        assert np.array_equal(got, padded[:, p:p + 8]), p
