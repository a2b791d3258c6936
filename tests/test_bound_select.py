"""The one-histogram-pass selection rules of round 4 (dpq_kernels.hip), restated in numpy and checked for what the kernels
rely on.  CPU only: the kernels themselves are checked against the oracle by the -m gpu parity tests (any valid upper bound
of the k-th key leaves the results bit-identical; these tests pin WHY the bound is valid).

* block_kth_bound_u32 (bootstrap_kernel<M, V & 1>): 1024 bins over (key - min) cut from the top set bit of (max - min);
  the threshold is the upper edge of the bin the rank falls into, never above max.
* select_kernel's last level (SelectArgs.fast_final): a bucket sort -- 1024 bins over (key - least distance bits << 32) of the
  64-bit keys; the keys up to the rank's bin are scattered to their bins' ranges (bin starts = prefix of the counts) and a
  key's output rank is its bin's start + the smaller keys inside its bin.
"""
import numpy as np
import pytest

BITS = 10


def kth_bound_u32(keys, rank):
    """block_kth_bound_u32<10>: returns (bound, shift)."""
    keys = np.asarray(keys, dtype=np.uint32)
    lo, hi = int(keys.min()), int(keys.max())
    span = hi - lo
    hi_bit = span.bit_length() - 1 if span else 0
    shift = hi_bit - (BITS - 1) if hi_bit >= BITS else 0
    bins = (keys.astype(np.uint64) - np.uint64(lo)) >> np.uint64(shift)
    assert bins.max() < (1 << BITS)                       # the histogram has 2^BITS words
    hist = np.bincount(bins.astype(np.int64), minlength=1 << BITS)
    incl = np.cumsum(hist)
    b = int(np.searchsorted(incl, rank, side="left"))     # first bin whose running count reaches the rank
    edge = lo + ((b + 1) << shift) - 1
    return min(edge, hi), shift


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("n,rank", [(3072, 100), (2048, 10), (6144, 100), (12288, 1000), (100, 100), (2048, 1)])
def test_bootstrap_bound_is_an_upper_bound_within_one_bin(seed, n, rank):
    rng = np.random.default_rng(seed * 1000 + n + rank)
    # fp32 distance bits of one query: positive floats of comparable magnitude, with duplicates
    d = (rng.gamma(4.0, 9000.0, size=n) + rng.uniform(1e3, 5e4)).astype(np.float32)
    if seed % 2:
        d[rng.integers(0, n, size=n // 8)] = d[0]
    keys = d.view(np.uint32)
    bound, shift = kth_bound_u32(keys, rank)
    exact = int(np.sort(keys)[rank - 1])
    assert bound >= exact                                  # a valid threshold: at least `rank` keys are <= it
    assert bound - exact < (1 << shift)                    # ... and less than one bin above the exact answer
    assert int((keys <= bound).sum()) >= rank
    # what it costs: positive floats compare like their bits, a bin is at most 2^-9 of the span's top binade
    assert (1 << shift) <= max(1, (int(keys.max()) - int(keys.min())) >> (BITS - 1)) + 1


def test_bootstrap_bound_degenerate_spans():
    assert kth_bound_u32([7, 7, 7, 7], 3)[0] == 7                      # span 0
    assert kth_bound_u32([5, 6, 7, 8], 2) == (6, 0)                    # span below 2^BITS: exact
    k = np.array([0, 0xfffffffe], dtype=np.uint32)                     # the widest span: no overflow of the edge
    assert kth_bound_u32(k, 1)[0] >= 0 and kth_bound_u32(k, 2)[0] == 0xfffffffe
    inf = np.array([np.float32(3.0), np.float32(np.inf), np.float32(4.0)]).view(np.uint32)
    assert kth_bound_u32(inf, 2)[0] >= int(np.float32(4.0).view(np.uint32))


def fast_final(keys, k):
    """select_kernel, fast_final (a bucket sort): returns the k winners in output order and the slot's rerun threshold, or
    (None, None) where the kernel takes the exact way."""
    keys = np.asarray(keys, dtype=np.uint64)
    dlo, dhi = int(keys.min() >> np.uint64(32)), int(keys.max() >> np.uint64(32))
    lo64, hi64 = dlo << 32, (dhi << 32) | 0xffffffff
    hi_bit = (hi64 - lo64).bit_length() - 1
    assert 31 <= hi_bit <= 62                               # distance bits are those of a float >= 0
    shift = hi_bit - 9
    bins = ((keys - np.uint64(lo64)) >> np.uint64(shift)).astype(np.int64)
    assert bins.max() < 1024
    hist = np.bincount(bins, minlength=1024)
    incl = np.cumsum(hist)
    b = int(np.searchsorted(incl, k, side="left"))           # the bin of the k-th key
    upto = int(incl[b])
    if upto > k + 64 or hist[: b + 1].max() > 64:            # crowded: the radix select
        return None, None
    edge = min(lo64 + ((b + 1) << shift) - 1, hi64)
    start = incl - hist                                       # counts -> bin starts = the scatter's cursors
    cursor = start.copy()
    wp = np.zeros(upto, dtype=np.uint64)
    for key, bn in zip(keys, bins):                           # any order of arrival
        if bn <= b:
            wp[cursor[bn]] = key
            cursor[bn] += 1
    out = np.zeros(k, dtype=np.uint64)
    for i in range(upto):
        mine = wp[i]
        bn = int((int(mine) - lo64) >> shift)
        s0, e0 = (int(cursor[bn - 1]) if bn else 0), int(cursor[bn])   # bin bn's range after the scatter
        rank = s0 + int((wp[s0:e0] < mine).sum())
        if rank < k:
            out[rank] = mine
    return out, edge


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("n,k", [(400, 100), (130, 128), (3000, 256), (101, 100), (900, 10), (4800, 1000), (9000, 2048)])
def test_fast_final_level_returns_the_k_smallest_in_order(seed, n, k):
    rng = np.random.default_rng(seed * 77 + n + k)
    d = rng.gamma(3.0, 7000.0, size=n).astype(np.float32)
    if seed % 3 == 1:
        d[: n // 2] = d[0]                                   # ties: equal distances, different ids
    ids = rng.permutation(1 << 20)[:n].astype(np.uint64)
    keys = (d.view(np.uint32).astype(np.uint64) << np.uint64(32)) | ids
    out, edge = fast_final(keys, k)
    exact = np.sort(keys)[:k]
    if out is None:                                          # a crowded bin: the kernel falls back to the radix select
        assert seed % 3 == 1
        return
    assert np.array_equal(out, exact)
    assert edge >= int(exact[-1])                            # the slot's threshold for a rerun bounds the k-th key
