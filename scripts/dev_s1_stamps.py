"""Per-wavefront timeline of strand1_kernel (GPU box): prologue, first strip, later strips.  usage: [--codes N]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deltapq_amd import _lib, api, synth
ap = argparse.ArgumentParser()
ap.add_argument("--codes", type=int, default=32_000_000)
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--burst", type=int, default=0, help="calls enqueued back to back before the stamped one (keeps the GPU busy)")
a = ap.parse_args()
n, k = a.codes, 100
cache = "/tmp/dpq_stream_%d.npy" % n
if os.path.exists(cache):
    payload = np.load(cache, mmap_mode="r")
else:
    tree = synth.synth_tree_large(n, 8, seed=102, mean_diffs=3.0)
    payload, _ = synth.encode_dtc(tree)
    np.save(cache, payload)
cb = synth.make_codebook(8, 256, 16, seed=100)
qs = synth.make_queries(64, 128, seed=101)
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.dpq_debug_strand1_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
with api.DeltaPQIndex.open_memory(np.asarray(payload), n, 8, 256, flags=a.flags) as idx:
    idx.set_codebook(cb)
    for i in range(3):
        idx.query_batch(qs[i:i + 1], k)
    buf = np.zeros(256 * 16 * 16, dtype=np.uint64)
    assert raw.dpq_debug_strand1_stamps(idx._h, buf.ctypes.data, buf.size) == 0   # arms
    if a.burst:
        qd = torch.from_numpy(qs).cuda()
        oi = torch.empty((1, k), dtype=torch.int32, device="cuda"); od = torch.empty((1, k), dtype=torch.float32, device="cuda")
        for i in range(a.burst):
            idx.query_batch_torch(qd[i % 60:i % 60 + 1].contiguous(), k, oi, od, wait=False)
        idx.finish()
        torch.cuda.synchronize()
    else:
        idx.query_batch(qs[5:6], k)
    assert raw.dpq_debug_strand1_stamps(idx._h, buf.ctypes.data, buf.size) == 0
raw_t = buf.reshape(256, 16, 16)
t = (raw_t & np.uint64(0xffffffffff)).astype(np.int64)
sclk = (raw_t >> np.uint64(40)).astype(np.int64)
t[:, :, 14:] = raw_t[:, :, 14:].astype(np.int64)
live = t[:, :, 0] > 0
t0 = t[:, :, 0][live].min()
us = (t - t0) / 100.0
print("wavefronts alive: %d; start offset: median %.1f max %.1f us" % (live.sum(), np.median(us[:, :, 0][live]), us[:, :, 0][live].max()))
pro = (us[:, :, 1] - us[:, :, 0])[live]
print("prologue: median %.1f max %.1f us" % (np.median(pro), pro.max()))
print("first strip: entries into the exact-check path per wavefront median %.0f max %.0f (of 16 phases); time in it median %.1f max %.1f us"
      % (np.median(t[:, :, 15][live]), t[:, :, 15][live].max(), np.median(t[:, :, 14][live]) / 100.0, t[:, :, 14][live].max() / 100.0))
for i in range(2, 12):
    ok = live & (t[:, :, i] > 0)
    if ok.sum() == 0:
        break
    d = (us[:, :, i] - us[:, :, i - 1])[ok]
    ghz = ((sclk[:, :, i] - sclk[:, :, i - 1]) & 0xffffff)[ok] * 1024 / ((t[:, :, i] - t[:, :, i - 1])[ok] * 10.0)
    print("strip %d: %d wavefronts, duration min %.1f median %.1f p90 %.1f max %.1f us; ends at median %.1f max %.1f us; shader clock median %.2f GHz"
          % (i - 2, ok.sum(), d.min(), np.median(d), np.percentile(d, 90), d.max(), np.median(us[:, :, i][ok]), us[:, :, i][ok].max(), np.median(ghz)))
# where are the slow wavefronts?  by XCD (blockIdx % 8 under round-robin placement), by wavefront slot, within / between workgroups
for i in (3, 5, 8):
    ok = live & (t[:, :, i] > 0)
    if ok.sum() < 1000:
        continue
    d = np.where(ok, us[:, :, i] - us[:, :, i - 1], np.nan)
    print("strip %d by XCD:" % (i - 2), " ".join("%.1f" % np.nanmedian(d[x::8]) for x in range(8)),
          "| by wavefront slot:", " ".join("%.0f" % np.nanmedian(d[:, w]) for w in range(16)))
    wg_med = np.nanmedian(d, axis=1)
    print("   workgroup medians: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f; spread inside a workgroup (max - min) median %.1f us"
          % (np.nanmin(wg_med), np.nanpercentile(wg_med, 10), np.nanmedian(wg_med), np.nanpercentile(wg_med, 90), np.nanmax(wg_med),
             np.nanmedian(np.nanmax(d, axis=1) - np.nanmin(d, axis=1))))
    slow = np.argsort(wg_med)[-12:]
    print("   slowest workgroups:", " ".join("%d(%.0f)" % (b, wg_med[b]) for b in slow))
