#!/usr/bin/env python3
"""bench.py -- queries/s of the DeltaPQ `-task query` hot path on MI355X.

Workload at N = 1 (BASELINE.json configs[1]): SIFT1M-shaped index, 1e6 codes, m = 8, k = 256, top-100,
1000 queries per step, synthetic data made by the build's own pipeline (vectors -> k-means codebook -> PQ
codes -> DeltaTree -> DTC).  A "step" answers one whole query batch: table build + threshold bootstrap +
delta-decode/ADC filter scan + select.  Queries and results stay in HBM inside the timed regions.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
      bench.py --gpus N --steps K --warmup W

N > 1 (the north-star decomposition, "scaling": "strong"): the index is cut into N DFS-position ranges, one
per rank; every rank answers the whole batch on its range; ONE all-gather of the partial top-k lists (RCCL)
and a device merge.  `--data pipeline` builds the same seeded index on every rank and opens the rank's range;
`--data stream` (indexes too large to build: BASELINE configs[3]/[4], `--codes 100000000` / `1000000000`)
synthesises only the rank's own range (no rank ever holds N codes).  Where every rank can hold the whole index
(`--data pipeline`), the query-replica decomposition (every GPU answers its own batch on a full copy, weak
scaling, no collective) is timed after the headline run and reported beside it under "query_replicas";
`--shard query` makes it the headline instead.

The timed region of K steps is repeated --reps times (each bracketed by barrier + synchronize); `value` is the
median repetition, min/max are reported.  Steps rotate over 4 distinct query batches.  Rank 0 prints ONE JSON
line with `roofline` (binding resource of the scan kernel: LDS gather bandwidth; measured HBM beside it) and
`cpu_baseline` (the oracle restatement of the reference, timed on this host).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured streaming copy)
LDS_PEAK_GBPS = 157286.4    # 256 CUs x 256 B/clk (ds_read_b128) x 2.4 GHz, MI355X_MICROARCH.md section LDS
N_BATCHES = 4               # distinct query batches the steps rotate over


def pmc_traffic(args, world):
    """HBM bytes per scan-kernel launch from the committed rocprofv3 --pmc passes (scripts/collect_pmc.sh:
    FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 x2 read correction).  Valid for the workload it was
    taken on: the summary names it."""
    for name in ("r04_pmc_summary_default.json", "r04_pmc_summary_125M_stream.json", "r04_pmc_summary_m16_top1000.json",
                 "r03_pmc_summary_default.json", "r03_pmc_summary_125M.json", "r03_pmc_summary_125M_stream.json",
                 "r02_pmc_summary_default.json", "r02_pmc_summary_125M.json", "r02_pmc_summary_125M_stream.json"):
        path = os.path.join(HERE, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            s = json.load(f)
        w = s.get("workload", {})
        if (w.get("n"), w.get("queries"), w.get("topk"), w.get("m"), w.get("data"), w.get("gpus")) == \
                (args.n, args.queries, args.topk, args.m, args.data, world):
            if "stream_kernel_hbm_bytes_per_launch" in s:   # one query per pass
                return s["stream_kernel_hbm_bytes_per_launch"], s.get("stream_kernel_avg_launch_ms_kernel_trace"), name
            return s.get("scan_kernel_hbm_bytes_per_launch"), s.get("scan_kernel_avg_launch_ms_under_pmc"), name
    return None, None, None


def profile_launch_ms(args, world):
    """The scan kernel's average launch duration in the committed rocprofv3 --kernel-trace --stats run of this workload
    (profiles/*_kernel_stats.csv, summarised by scripts/collect_pmc.sh), so that a reader can recompute `roofline.frac`
    from profiles/ alone: (ms, file) or (None, None) where no profile of this workload is committed."""
    for name in ("r04_pmc_summary_default.json", "r04_pmc_summary_m16_top1000.json", "r03_pmc_summary_default.json"):
        path = os.path.join(HERE, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            s = json.load(f)
        w = s.get("workload", {})
        if (w.get("n"), w.get("queries"), w.get("topk"), w.get("m"), w.get("data"), w.get("gpus")) == \
                (args.n, args.queries, args.topk, args.m, args.data, world) and s.get("scan_kernel_avg_launch_ms_kernel_trace"):
            return float(s["scan_kernel_avg_launch_ms_kernel_trace"]), name
    return None, None


def hbm_leg():
    """The regime the north star's HBM target is about, measured in the same run: ONE query per call (the reference's
    call shape) on an index far beyond L2 + Infinity Cache -- 125 M codes, one GPU's share of BASELINE configs[4] --
    so that every call streams the compressed index from HBM (the stream pass).  A child process
    (its own 125 M-code index; this process keeps its workload), its bench line condensed."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--codes", "125000000", "--data", "stream", "--queries", "1", "--steps", "10",
           "--warmup", "2", "--reps", "3", "--check", "4", "--min-check", "4", "--no-cpu-baseline", "--sustain-seconds", "0", "--host-steps", "0",
           "--no-hbm-leg"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=HERE)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:   # noqa: BLE001 -- the headline must not depend on this leg
        return {"error": repr(e)[:300]}
    ro = d["roofline"]
    return {"workload": d["config"]["workload"], "n_codes": d["config"]["n_codes"], "payload_bytes": d["config"]["n_bytes"],
            "queries_per_call": 1, "ms_per_call": d["ms_per_step"], "queries_per_s": d["value"],
            "kernel": ro["kernel"], "kernels_per_call": ro.get("stream_pass_launches_per_step"), "bound": ro["bound"],
            "achieved_GBps": ro["achieved"], "peak_GBps": ro["peak"], "frac": ro["frac"],
            "whole_call_GBps": d["config"]["n_bytes"] / (d["ms_per_step"] * 1e-3) / 1e9,
            "definition": "achieved = DTC payload bytes of the index / kernel time of the call's stream-pass launches (HIP events); "
                          "whole_call_GBps = the same bytes / the whole call (table build, bootstrap, three levels, selects)",
            "traffic": ro.get("traffic"), "parity_checked_queries": d["parity_checked_queries"],
            "device_bytes": d["index"]["device_bytes_rank0"], "strand_image_bytes": d["index"].get("strand_bytes_rank0")}


def make_queries(args, seed):
    from deltapq_amd import synth
    if args.data == "pipeline":
        return synth.make_clustered_vectors(args.queries, args.dim, seed=seed, n_clusters=20000, spread=12.0, centre_seed=7)
    return synth.make_queries(args.queries, args.dim, seed=seed)


def build_workload(args, device, rank, world, by_query):
    """Returns dict(codebook, payload, n_local, n_bytes_local, open_kwargs, desc).
    pipeline: the whole seeded index on every rank (cut by the library: shard_rank / shard_count).
    stream:   only this rank's part, as a self-contained stream with its global position (N > 1, index shards)."""
    from deltapq_amd import api, synth
    t0 = time.time()
    # --index-dir: the pipeline-built index (codebook + DTC payload) cached on disk, so that a profiled run (rocprofv3
    # --pmc passes collect counters per dispatch) holds the QUERY path only -- building 1 M codes at M = 16 inside the
    # profiled process was 3.9 M dispatches and timed the counter passes out
    cache = None
    if args.index_dir and args.data == "pipeline":
        os.makedirs(args.index_dir, exist_ok=True)
        cache = os.path.join(args.index_dir, "pipeline_n%d_m%d_d%d.npz" % (args.n, args.m, args.dim))
        if os.path.exists(cache):
            z = np.load(cache)
            kw = dict(shard_rank=0 if by_query else rank, shard_count=1 if by_query else world)
            return dict(codebook=z["codebook"], payload=z["payload"], n_local=args.n, n_bytes_local=len(z["payload"]), open_kwargs=kw,
                        offset=0, whole=True, gen_s=time.time() - t0, desc=str(z["desc"]) + " (index loaded from --index-dir)")
    if args.data == "pipeline":
        if args.n > 8_000_000:
            raise SystemExit("--data pipeline builds the whole index on every rank: use --data stream beyond 8 M codes")
        base = synth.make_clustered_vectors(args.n, args.dim, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
        cb = synth.kmeans_codebook(base, args.m, 256, iters=6, seed=102)
        codes = api.encode_pq(base, cb, device=device)
        del base
        tree = api.DeltaTree(codes, codebook=cb, device=device)      # edge search on the GPU, layout on the host
        payload = tree.payload()
        uniq = len(np.unique(codes.view("V%d" % args.m))) / args.n
        tree.close()
        kw = dict(shard_rank=0 if by_query else rank, shard_count=1 if by_query else world)
        desc = "vectors->kmeans->PQ encode->built DeltaTree, %.1f%% unique codes" % (100 * uniq)
        if cache and rank == 0:
            np.savez(cache + ".tmp.npz", codebook=cb, payload=payload, desc=desc)
            os.replace(cache + ".tmp.npz", cache)
        return dict(codebook=cb, payload=payload, n_local=args.n, n_bytes_local=len(payload), open_kwargs=kw, offset=0,
                    whole=True, gen_s=time.time() - t0, desc=desc)
    cb = synth.make_codebook(args.m, 256, args.dim // args.m, seed=100)
    if by_query or world == 1:
        tree = synth.synth_tree_large(args.n, args.m, seed=102, mean_diffs=args.mean_diffs)
        payload, nb = synth.encode_dtc(tree)
        return dict(codebook=cb, payload=payload, n_local=args.n, n_bytes_local=nb, open_kwargs={}, offset=0, whole=True,
                    gen_s=time.time() - t0, desc="random (depth, mask, bytes) stream")
    # index shards of a stream too large to build anywhere: this rank's DFS range only
    per = args.n // world
    n_local = per if rank + 1 < world else args.n - per * (world - 1)
    tree = synth.synth_tree_large(n_local, args.m, seed=102 + 1000 * rank, mean_diffs=args.mean_diffs)
    payload, nb = synth.encode_dtc(tree)
    del tree
    kw = dict(global_offset=rank * per, global_n_codes=args.n)
    return dict(codebook=cb, payload=payload, n_local=n_local, n_bytes_local=nb, open_kwargs=kw, offset=rank * per,
                whole=False, gen_s=time.time() - t0,
                desc="random (depth, mask, bytes) stream, every rank synthesises its own DFS range of %d codes" % per)


def canon_ids(ids, n):
    """The even-N id rule (h:2949, 2970) reports the last node as n: fold it back to n - 1 for comparisons
    between a part of an index and the oracle run on that part alone."""
    ids = np.asarray(ids).astype(np.int64)
    return np.where(ids == n, n - 1, ids) if n % 2 == 0 else ids


def parity_gate(args, wl, queries, hi, hd, budget_s=25.0):
    """Oracle check of the first rows of a result (rank 0).  Whole index: the final (merged) lists.  Own part of a
    larger index: this rank's partial lists against the oracle run on the part."""
    from oracle import dtc_oracle as O
    orc = O.Oracle()
    n, off, k = wl["n_local"], wl["offset"], args.topk
    checked, t0 = 0, time.time()
    for i in range(min(args.check, len(queries))):
        lut = orc.build_lut(wl["codebook"], queries[i])
        oi, od, alld, _ = orc.scan_lut(wl["payload"], n, lut, k, want_all=True)
        if wl["whole"]:
            ok, msg = O.tie_aware_equal(hi[i], hd[i], oi, od, alld, n)
        else:
            mine = canon_ids(hi[i], args.n) - off
            ok, msg = O.tie_aware_equal(mine, hd[i], canon_ids(oi, n), od, alld, n | 1)
        if not ok:
            print("bench.py: PARITY FAILURE on query %d: %s" % (i, msg), file=sys.stderr)
            sys.exit(4)
        checked += 1
        if time.time() - t0 > budget_s and checked >= min(args.min_check, args.check):
            break
    return checked


def cpu_baseline(wl, args, queries):
    """The oracle (CPU restatement of h:3731-3892, built -O3 like the reference) on a bounded sample of the
    same workload: 1 thread (the reference's own parallelism, main:328), all host cores (one query per C++
    thread), and the O_DIRECT variant (what `-task query` literally does: reopen + 4 KB reads per query)."""
    from oracle import dtc_oracle as O
    orc = O.Oracle()
    cb, payload, n, k = wl["codebook"], wl["payload"], wl["n_local"], args.topk
    t0 = time.time()
    orc.query_many(payload, n, cb, queries[:1], k, 1)
    per_query = max(time.time() - t0, 1e-4)
    n1 = int(min(len(queries), max(2, args.cpu_seconds / per_query)))
    t0 = time.time()
    orc.query_many(payload, n, cb, queries[:n1], k, 1)
    t1 = time.time() - t0
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:   # a container's CPU quota caps what the threads can use whatever the affinity mask says
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(period)
    except Exception:
        pass
    nall = int(min(len(queries), max(cores, cores * (args.cpu_seconds / 2) / (t1 / n1))))
    t0 = time.time()
    orc.query_many(payload, n, cb, queries[:nall], k, cores)
    tall = time.time() - t0
    od = None
    try:
        from deltapq_amd import synth
        d = tempfile.mkdtemp(prefix="dpq_bench_", dir=os.path.join(HERE, "gpurun_out") if os.path.isdir(os.path.join(HERE, "gpurun_out")) else None)
        path = synth.dtc_file_name(d, args.m, 256, n)
        synth.write_dtc_file(path, n, payload)
        honoured = orc.o_direct_supported(path)
        n_od = int(min(len(queries), max(2, 3.0 / (5 * t1 / n1))))
        t0 = time.time()
        for i in range(n_od):
            orc.query_o_direct(path, n, cb, queries[i], k)
        tod = time.time() - t0
        od = {"value": n_od / tod, "ms_per_query": 1e3 * tod / n_od, "queries": n_od, "o_direct_honoured_by_filesystem": honoured}
        os.remove(path)
        os.rmdir(d)
    except Exception as e:  # the baseline leg must not take the bench down
        od = {"error": repr(e)}
    return {"value": n1 / t1, "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": "first %d of the %d queries of batch 0, %s N=%d index, in-memory scan (query_im twin), %.1f s; oracle built -O3"
                      % (n1, len(queries), "full" if wl["whole"] else "this rank's part of the", n, t1),
            "ms_per_query": 1e3 * t1 / n1,
            "all_cores": {"value": nall / tall, "unit": "queries/s", "cores": cores, "cgroup_cpu_quota": quota, "queries": nall,
                          "how": "one query at a time per std::thread inside liboracle.so"},
            "o_direct_variant_1_thread": od}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reps", type=int, default=10, help="repetitions of the timed K-step region (median is the value)")
    ap.add_argument("--n", "--codes", dest="n", type=int, default=1_000_000,
                    help="codes in the whole index (under torch.distributed.run use --codes: its parser takes --n for its own)")
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--topk", type=int, default=100)
    ap.add_argument("--m", type=int, default=8)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--data", choices=["pipeline", "stream"], default="pipeline")
    ap.add_argument("--mean-diffs", type=float, default=3.0, help="changed bytes per node (--data stream)")
    ap.add_argument("--chunks-per-segment", type=int, default=0)
    ap.add_argument("--bootstrap", type=int, default=0, help="dpq_open_opts.bootstrap: 0 auto, 1 on, -1 off")
    ap.add_argument("--batch-decode", type=int, default=0,
                    help="dpq_open_opts.batch_decode: 0 auto (batches of >= 3 query groups decode once into a plain-code "
                         "scratch), 1 always, -1 never (decode inside the scan)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU time given to each leg of the oracle baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustain-seconds", type=float, default=2.0, help="length of the sustained block after the timed regions (0 = skip)")
    ap.add_argument("--host-steps", type=int, default=20, help="steps of the host-to-host leg through dpq_query_batch (N = 1; 0 = skip)")
    ap.add_argument("--min-check", type=int, default=8,
                    help="queries the parity gate verifies whatever its time budget (a 125 M-code oracle scan takes seconds per query)")
    ap.add_argument("--check", type=int, default=64, help="queries verified against the oracle before timing (time-bounded)")
    ap.add_argument("--shard", choices=["auto", "index", "query"], default="auto",
                    help="N > 1: 'index' (= auto) cuts the index into DFS ranges, every GPU answers the one batch on its "
                         "range, one all-gather + merge (strong scaling, the north-star decomposition); 'query' = every GPU "
                         "holds the whole index and answers its own batch (weak scaling, no collective)")
    ap.add_argument("--sharded-streams", type=int, default=2,
                    help="index shards: stream-ordered steps alternate between this many torch streams (1 or 2)")
    ap.add_argument("--index-dir", default="",
                    help="--data pipeline: cache the built index (codebook + DTC payload) in this directory and load it from there "
                         "when present: profiled runs then hold the query path only")
    ap.add_argument("--build-only", action="store_true", help="with --index-dir: build + cache the index and exit")
    ap.add_argument("--no-hbm-leg", action="store_true",
                    help="skip the HBM-regime leg of the default N = 1 run (one query per call on a 125 M-code index, in a child process)")
    ap.add_argument("--no-replicas", action="store_true",
                    help="N > 1 index shards of an index every rank could hold whole: skip the query-replica run timed beside it")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share GPUs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from deltapq_amd import api
    from deltapq_amd import dist as dpq_dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world),
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the DeltaPQ query path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    if args.backend == "gloo" or os.environ.get("DPQ_BENCH_SHARE_GPUS") == "1":   # rehearsals: more ranks than devices
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cpu_coll = args.backend == "gloo"   # gloo collectives take host tensors
    backend_note = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not cpu_coll:
            # RCCL, proved with one small collective before anything is built on it; if the node refuses it (no P2P
            # between the visible devices, IPC mode ...) every rank sees the failure here and the run goes on over
            # gloo with host-staged lists -- a slower exchange step, said so on the JSON line, instead of no line.
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize(dev)
                assert int(probe.item()) == world
            except Exception as e:   # noqa: BLE001 -- any failure of the collective layer
                backend_note = "RCCL unusable (%s: %s); exchange step over gloo with host staging" % (type(e).__name__, str(e)[:200])
                print("bench.py rank %d: %s" % (rank, backend_note), file=sys.stderr)
                try:
                    dist.destroy_process_group()
                except Exception:   # noqa: BLE001
                    pass
                cpu_coll = True
        if cpu_coll:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    by_query = world > 1 and args.shard == "query"
    nq, k = args.queries, args.topk
    wl = build_workload(args, local_rank, rank, world, by_query)
    if args.build_only:
        if rank == 0:
            print(json.dumps({"built": wl["desc"], "seconds": wl["gen_s"], "index_dir": args.index_dir}))
        return

    def open_index(**kw):
        idx = api.DeltaPQIndex.open_memory(wl["payload"], wl["n_local"], args.m, 256, device=local_rank,
                                           chunks_per_segment=args.chunks_per_segment, bootstrap=args.bootstrap,
                                           batch_decode=args.batch_decode, **kw)
        idx.set_codebook(wl["codebook"])
        return idx

    idx = open_index(**wl["open_kwargs"])
    info = idx.info()
    # N_BATCHES distinct query batches; index shards answer the SAME batch on every rank, replicas their own
    batches_np = [make_queries(args, 101 + b + (1000 * rank if by_query else 0)) for b in range(N_BATCHES)]
    batches = [torch.from_numpy(b).to(dev) for b in batches_np]
    ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    dists = torch.empty((nq, k), dtype=torch.float32, device=dev)

    # Batches that no collective follows are pipelined (dpq_query_batch_device_async: two lanes); sync() settles them
    # (dpq_finish: waits, checks the overflow words, reruns if needed).  Index shards consume the partial lists on
    # the device in stream order (select -> pack -> all-gather -> merge: dpq_query_batch_device_ordered), so a step
    # needs no host round trip either; if dpq_finish had to answer a batch again (a query overflowed its candidate
    # buffers: the merged list of that step was built from an incomplete partial list) the region is repeated with
    # one synchronous call per step.
    sharded = world > 1 and not by_query
    pipelined = not sharded
    state = {"ordered": sharded, "reruns": 0}

    # Stream-ordered sharded steps alternate between two torch streams: the library gives each caller stream one of
    # its two workspaces, so step i + 1's decode / table build run under step i's scan and its bootstrap beside step
    # i's select, while every step's exchange (pack -> all-gather -> merge) still follows its own batch in stream order.
    n_step_streams = max(1, min(2, args.sharded_streams)) if sharded else 1
    # every step in flight owns its output pair: two lanes = two batches in flight
    step_streams, step_out = [], [(ids, dists), (torch.empty_like(ids), torch.empty_like(dists))]
    if sharded and n_step_streams == 2:
        step_streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        for st_ in step_streams:
            st_.wait_stream(torch.cuda.current_stream(dev))   # the query batches are there

    def step(i, index=None):
        if pipelined:
            o_ids, o_dists = step_out[i & 1]
            (index or idx).query_batch_torch(batches[i % N_BATCHES], k, o_ids, o_dists, wait=False)
            return o_ids, o_dists   # this rank's batch; complete after sync()
        # index shards: the path's one exchange step -- all-gather of the partial lists
        # (nq * k * 8 B per rank) over RCCL, then the device merge
        if state["ordered"] and step_streams:
            o_ids, o_dists = step_out[i & 1]
            with torch.cuda.stream(step_streams[i & 1]):
                (index or idx).query_batch_torch(batches[i % N_BATCHES], k, o_ids, o_dists, wait=False, ordered=True)
                return dpq_dist.gather_and_merge(o_ids, o_dists)
        (index or idx).query_batch_torch(batches[i % N_BATCHES], k, ids, dists, wait=not state["ordered"], ordered=state["ordered"])
        return dpq_dist.gather_and_merge(ids, dists)

    def sync(index=None):
        state["reruns"] += (index or idx).finish()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # correctness gate before timing (rank 0, against the oracle)
    out_ids, out_dists = step(0)
    sync()
    if nq == 1 and args.check > 1:
        # one query per call: the gate wants several queries through that very path -- one call each
        gate_q = np.concatenate([b[:1] for b in batches_np] + [make_queries(args, 900 + i)[:1] for i in range(max(0, args.check - N_BATCHES))])[:args.check]
        g_ids, g_d = [], []
        for i in range(len(gate_q)):
            gi, gd = idx.query_batch(gate_q[i:i + 1], k)
            g_ids.append(gi[0]), g_d.append(gd[0])
        gate = (gate_q, np.stack(g_ids), np.stack(g_d))
    else:
        gate = None
    parity = 0
    if rank == 0 and args.check > 0:
        if wl["whole"] and gate is not None:
            parity = parity_gate(args, wl, gate[0], gate[1], gate[2])
        elif wl["whole"]:
            parity = parity_gate(args, wl, batches_np[0], out_ids.cpu().numpy(), out_dists.cpu().numpy())
        else:   # own part of a larger index: the rank's partial lists (still in ids / dists) against the oracle on the part
            parity = parity_gate(args, wl, batches_np[0], ids.cpu().numpy(), dists.cpu().numpy())
            m_ids = out_ids.cpu().numpy()
            assert np.all(np.diff(out_dists.cpu().numpy(), axis=1) >= 0) and m_ids.min() >= 0 and m_ids.max() <= args.n

    # ... and the overlapped path itself (two batches in flight on the two lanes, each with its own outputs) before it
    # is timed: the first rows of both against the oracle
    overlap_parity = 0
    if pipelined:
        o0, o1 = step(0), step(1)
        sync()
        if rank == 0 and args.check > 0 and wl["whole"]:
            for b, (oi, od) in enumerate((o0, o1)):
                overlap_parity += parity_gate(args, wl, batches_np[b][:max(2, args.check // 8)], oi.cpu().numpy(), od.cpu().numpy(),
                                              budget_s=8.0)

    def timed(index=None, reps=args.reps):
        """--reps repetitions of the K-step region; returns the per-repetition wall times (max over ranks)."""
        for i in range(args.warmup):
            step(i, index)
        sync(index)
        times = []
        for _ in range(max(1, reps)):
            sync(index)
            t0 = time.perf_counter()
            for i in range(args.steps):
                step(i, index)
            sync(index)
            times.append(time.perf_counter() - t0)
        t = torch.tensor(times, dtype=torch.float64, device=torch.device("cpu") if cpu_coll else dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.cpu().numpy()

    # The timed regions run without any profiling events.  Kernel times (the roofline's scan time among them) come
    # from a short pass afterwards: synchronous calls (one batch at a time on the caller's stream: with pipelined
    # batches on two lanes an event interval would include the wait for the other lane's scan) with HIP events
    # around every kernel.
    idx.profile_enable(0)
    state["reruns"] = 0
    times = timed()
    if sharded:   # did any rank answer a batch again?  Then its merged lists were stale: repeat with synchronous steps
        flag = torch.tensor([state["reruns"]], dtype=torch.int64, device=torch.device("cpu") if cpu_coll else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) > 0:
            state["ordered"] = False
            times = timed()
    # Sustained rate: the same steps back to back for >= --sustain-seconds (the K-step regions above last a few
    # milliseconds: boost clocks; MI355X_MICROARCH.md 'DVFS give-back').  Same work per step, nothing skipped.
    sustained = None
    if args.sustain_seconds > 0:
        sync()
        n_done, t0 = 0, time.perf_counter()
        while True:
            for i in range(args.steps):
                step(n_done + i)
            n_done += args.steps
            sync()
            el = time.perf_counter() - t0
            tt = torch.tensor([el], dtype=torch.float64, device=torch.device("cpu") if cpu_coll else dev)
            if world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            if float(tt.item()) >= args.sustain_seconds:
                el = float(tt.item())
                break
        sustained = {"value": (world if by_query else 1) * nq * n_done / el, "unit": "queries/s", "seconds": el, "steps": n_done,
                     "ms_per_step": 1e3 * el / n_done,
                     "note": "the same steps back to back for >= %.1f s, a host synchronisation every %d steps" % (args.sustain_seconds, args.steps)}

    # Host-to-host rate through the drop-in boundary (host pointers in and out, PCIe both ways), page-locked buffers; never
    # used for `value`.  Pipelined (dpq_query_batch_host_async: the copies of neighbouring batches beside a batch's
    # kernels, settled by dpq_finish every `steps` batches) and, beside it, one synchronous dpq_query_batch per batch.
    host_to_host = None
    if world == 1 and args.host_steps > 0:
        hq = [api.pin_host(np.ascontiguousarray(b)) for b in batches_np]
        NH = 16   # batches in flight own their result buffers (dpq_query_batch_host_async: up to sixteen)
        h_out = [(api.pin_host(np.empty((nq, k), dtype=np.int32)), api.pin_host(np.empty((nq, k), dtype=np.float32))) for _ in range(NH)]
        import ctypes as _ct
        from deltapq_amd import _lib as _l
        fn = _l.load().dpq_query_batch
        for i in range(2):
            _l.check(fn(idx._h, _ct.c_void_p(hq[i % N_BATCHES].ctypes.data), nq, k, _ct.c_void_p(h_out[0][0].ctypes.data),
                        _ct.c_void_p(h_out[0][1].ctypes.data)), "dpq_query_batch")
        t0 = time.perf_counter()
        for i in range(args.host_steps):
            _l.check(fn(idx._h, _ct.c_void_p(hq[i % N_BATCHES].ctypes.data), nq, k, _ct.c_void_p(h_out[0][0].ctypes.data),
                        _ct.c_void_p(h_out[0][1].ctypes.data)), "dpq_query_batch")
        el_sync = time.perf_counter() - t0
        for i in range(4):
            idx.query_batch_host_async(hq[i % N_BATCHES], k, *h_out[i % NH])
        idx.finish()
        h_reps = []
        for _ in range(5):
            t0 = time.perf_counter()
            for i in range(args.host_steps):
                idx.query_batch_host_async(hq[i % N_BATCHES], k, *h_out[i % NH])
            idx.finish()
            h_reps.append(time.perf_counter() - t0)
        el = float(np.median(h_reps))
        host_to_host = {"value": nq * args.host_steps / el, "unit": "queries/s", "ms_per_step": 1e3 * el / args.host_steps,
                        "steps": args.host_steps,
                        "note": "dpq_query_batch_host_async + dpq_finish: page-locked host queries in (%d KB), ids + distances out (%d KB) "
                                "per step, up to sixteen batches in flight (results written by the select kernel into the mapped buffers), median of 5 runs of %d steps"
                                % (nq * args.dim * 4 // 1024, nq * k * 8 // 1024, args.host_steps),
                        "synchronous_dpq_query_batch": {"value": nq * args.host_steps / el_sync, "ms_per_step": 1e3 * el_sync / args.host_steps}}
        if args.check > 0 and wl["whole"]:
            last = (args.host_steps - 1)
            parity_gate(args, wl, batches_np[last % N_BATCHES][:4], h_out[last % NH][0], h_out[last % NH][1], budget_s=5.0)
        for a in hq + [x for pair in h_out for x in pair]:
            api.unpin_host(a)

    aux_steps = max(1, min(args.steps, 8))
    idx.profile_enable(1)
    idx.profile_reset()
    for i in range(aux_steps):
        idx.query_batch_torch(batches[i % N_BATCHES], k, ids, dists, wait=True)
    torch.cuda.synchronize(dev)
    prof = idx.profile_read()
    prof_aux = prof
    idx.profile_enable(0)
    total_steps = aux_steps

    replicas = None
    if sharded and not args.no_replicas and wl["whole"]:
        ridx = open_index()
        pipelined = True
        rt = timed(ridx, reps=3)
        pipelined = False
        ridx.finish()
        replicas = {"note": "same GPUs, every rank holds the whole index and answers its own %d-query batch; no collective" % nq,
                    "value": world * nq * args.steps / float(np.median(rt)), "unit": "queries/s", "scaling": "weak",
                    "ms_per_step": 1e3 * float(np.median(rt)) / args.steps}
        ridx.close()

    QG = 64 if args.m <= 8 else 32   # queries served by one decode pass (8-bit filter entries: 128 KB of LDS tables)
    groups = (nq + QG - 1) // QG
    S = 64 * info["chunks_per_segment"]
    batch_decoded = prof_aux["decode_ms"] > 0.0
    decode_ms_step = prof_aux["decode_ms"] / aux_steps
    lds_bytes_step = float(info["n_segments"]) * S * groups * (4 if args.m <= 8 else 2) * args.m * 16   # NG * M * 16 B per node and group
    stats = torch.tensor([prof["scan_ms"], float(prof["scan_launches"]), float(info["algorithmic_bytes"]),
                          float(info["device_bytes"] + info["bootstrap_bytes"]), prof_aux["select_ms"], prof_aux["lut_ms"],
                          prof_aux["quantise_ms"], lds_bytes_step, float(prof_aux["exact_checks"]), float(prof_aux["candidates"]),
                          float(info["node_hi"] - info["node_lo"]), prof_aux["decode_ms"], float(info["strand_bytes"]),
                          float(prof_aux["stream_launches"]), float(prof_aux["strand_launches"]), float(prof_aux["strand1_launches"])],
                         dtype=torch.float64, device=torch.device("cpu") if cpu_coll else dev)
    if world > 1:
        all_stats = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(all_stats, stats)
        all_stats = torch.stack(all_stats).cpu().numpy()
    else:
        all_stats = stats.cpu().numpy()[None, :]

    if rank == 0:
        med = float(np.median(times))
        scan_ms_step = float(all_stats[:, 0].max()) / total_steps          # slowest rank
        launches_step = float(all_stats[0, 1]) / total_steps
        alg_bytes_total = float(all_stats[:, 2].sum())     # index shards: n_bytes of the payload; replicas: world x n_bytes
        lds_total = float(all_stats[:, 7].sum())
        lds_gbps = lds_total / (scan_ms_step * 1e-3) / 1e9 if scan_ms_step > 0 else 0.0
        traffic, pmc_launch_ms, pmc_name = pmc_traffic(args, world)
        avg_launch_ms = scan_ms_step / launches_step if launches_step else None
        prof_ms, prof_name = profile_launch_ms(args, world)
        global_q = (world if by_query else 1) * nq
        bytes_per_code = float(all_stats[:, 2].sum()) / max(1.0, float(all_stats[:, 10].sum()))
        result = {
            "metric": "queries/sec, SIFT1M-shaped m=%d k=256 topk=%d" % (args.m, k),
            "value": global_q * args.steps / med,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * med / args.steps,
            "higher_is_better": True,
            "scaling": "weak" if by_query else "strong",
            "vs_baseline": None,
            "dtype": "f64 sum of f32 table entries (u8 code decode; the reference's incremental f64 stack)",
            "data": "synthetic",
            "config": {
                "workload": "SIFT1M-shaped synthetic (%s): N=%d m=%d k=256 h=1 topk=%d, %d queries/step, %.2f B/code"
                            % (wl["desc"], args.n, args.m, k, nq, bytes_per_code),
                "n_codes": args.n, "queries_per_step": nq, "topk": k,
                "n_bytes": int(alg_bytes_total if not by_query else alg_bytes_total / world),
                "sharding": "one GPU, whole index" if world == 1 else
                            ("query replicas x%d: every GPU holds the whole index and answers its own %d-query batch" % (world, nq)
                             if by_query else "dfs-range index shards x%d, one %d-query batch, one all-gather + merge" % (world, nq)),
                "global_queries_per_step": global_q,
                "collectives": None if world == 1 else (backend_note or ("gloo (host tensors)" if cpu_coll else "RCCL (backend nccl)")),
                "query_batches_rotated": N_BATCHES,
                "step_pipelining": "two lanes (dpq_query_batch_device_async)" if pipelined or not sharded else
                                   ("stream-ordered steps (dpq_query_batch_device_ordered) alternating between %d stream(s), no host round trip; "
                                    "no batch had to be answered again" % n_step_streams
                                    if state["ordered"] else "one synchronous call per step (a batch overflowed in the stream-ordered run)"),
                "queries_per_decode_pass": (groups * QG if batch_decoded else QG),
                "decode": ("once per batch (decode_list_kernel), tile by tile, into %d MB of plain codes that the %d query groups' "
                           "filter passes read through L2 / Infinity Cache" % (info["batch_decode_mb"], groups)) if batch_decoded
                          else "inside the scan kernel, once per %d-query group" % QG,
                "threshold_bootstrap": "multi-index, stride %d, %.1f MB" % (info["bootstrap_stride"], info["bootstrap_bytes"] / 1e6)
                                       if info["bootstrap_bytes"] else "off (spread-sample cascade)",
            },
            "repetitions": {"count": int(len(times)), "steps_each": args.steps,
                            "ms_per_step_median": 1e3 * med / args.steps,
                            "ms_per_step_min": 1e3 * float(times.min()) / args.steps,
                            "ms_per_step_max": 1e3 * float(times.max()) / args.steps,
                            "value_min": global_q * args.steps / float(times.max()),
                            "value_max": global_q * args.steps / float(times.min())},
            "roofline": {
                "bound": "lds",
                "kernel": "scan_kernel",
                "achieved": lds_gbps,
                "peak": LDS_PEAK_GBPS * world,
                "unit": "GB/s",
                "frac": lds_gbps / (LDS_PEAK_GBPS * world),
                "traffic": traffic,
                "traffic_source": None if traffic is None else "committed profile profiles/%s (rocprofv3 --pmc passes of this workload on another run), not measured in this run" % pmc_name,
                "definition": "achieved = LDS bytes of the ADC table gathers the scan launches issue (per decoded node and "
                              "%d-query group: M x %d ds_read_b128 = %d B, i.e. %d B per (code, query) pair, padding slots "
                              "included) / scan-kernel time (HIP events on the launch stream); peak = 256 CUs x 256 B/clk x "
                              "2.4 GHz.  The scan is a gather/lookup kernel bound by the LDS array (bank-conflict replays and the "
                              "decode's ds_bpermute traffic come on top of the counted bytes), not by HBM: see `hbm`."
                              % (QG, 4 if args.m <= 8 else 2, (4 if args.m <= 8 else 2) * args.m * 16, args.m),
                "lds_gather_bytes_per_step": lds_total,
                "hbm": None if traffic is None else {
                    "achieved": traffic / ((pmc_launch_ms or avg_launch_ms) * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": traffic / ((pmc_launch_ms or avg_launch_ms) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "bytes_per_launch": traffic, "source": "profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
                                                           "2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction)" % pmc_name},
                "algorithmic_hbm": {
                    "GBps": (nq * alg_bytes_total) / ((scan_ms_step + decode_ms_step) * 1e-3) / 1e9 if scan_ms_step > 0 else 0.0,
                    "bytes_per_step": nq * alg_bytes_total,
                    "note": "SURVEY.md 8(d) figure: queries x DTC payload bytes / (scan + per-batch decode) time.  Every decoded "
                            "chunk serves %d queries, so this is reuse, not traffic: it is NOT a fraction of the HBM roof"
                            % (groups * QG if batch_decoded else QG)},
                "launches_per_step": launches_step,
                "avg_launch_ms": avg_launch_ms,
                # the same kernel in the committed rocprofv3 --kernel-trace --stats run of this workload (one lane, another box /
                # another run): what `frac` is when recomputed from profiles/ alone
                "from_profiles": None if not (prof_ms and launches_step) else {
                    "avg_launch_ms_kernel_trace": prof_ms, "source": "profiles/%s" % prof_name,
                    "frac": lds_total / launches_step / (prof_ms * 1e-3) / 1e9 / (LDS_PEAK_GBPS * world)},
                "scan_ms_per_step": scan_ms_step,
                "select_ms_per_step": float(all_stats[:, 4].max()) / aux_steps,
                "lut_ms_per_step": float(all_stats[:, 5].max()) / aux_steps,
                "quantise_ms_per_step": float(all_stats[:, 6].max()) / aux_steps,
                "decode_ms_per_step": float(all_stats[:, 11].max()) / aux_steps,
                "filter_survivors_per_query": float(all_stats[:, 8].sum()) / max(1, aux_steps * nq),
                "candidates_per_query": float(all_stats[:, 9].sum()) / max(1, aux_steps * nq),
                "event_note": "kernel times (scan, select incl. bootstrap, lut, quantise, per-batch decode) from %d synchronous steps run after the "
                              "timed regions with HIP events around every kernel on the launch stream; the timed regions carry no "
                              "events (pipelined batches overlap on two lanes there)" % aux_steps,
            },
            "parity_checked_queries": parity,
            "parity_checked_queries_overlapped_steps": overlap_parity,
            "sustained": sustained,
            "host_to_host": host_to_host,
            # SURVEY.md 8(d) "report alongside": whole-job rates of the timed regions (median repetition)
            "rates": {"code_query_pairs_per_s": global_q * args.steps / med * float(all_stats[:, 10].sum()) / (1 if sharded or world == 1 else world),
                      "raw_pq_equivalent_GBps": global_q * args.steps / med * float(all_stats[:, 10].sum()) / (1 if sharded or world == 1 else world) * args.m / 1e9,
                      "note": "queries/s x codes each query is compared with (every code, exactly once); x M bytes = the bandwidth "
                              "a plain PQ scan would need at one query per pass"},
            "index": {"device_bytes_rank0": int(all_stats[0, 3] + all_stats[0, 12]), "strand_bytes_rank0": int(all_stats[0, 12]),
                      "segments_rank0": info["n_segments"],
                      "codes_rank0": int(all_stats[0, 10]), "gen_seconds": wl["gen_s"]},
        }
        dev = os.environ.get("DPQ_DEV", "0") not in ("", "0")
        stream_max = int(os.environ.get("DPQ_STREAM_MAX_QUERIES", "4")) if dev else 4   # the library's measured switch-over
        if nq <= stream_max:
            # up to four queries: the stream pass -- 1, 2 or 4 queries per pass over the compressed image, every decoded node
            # against the queries' exact tables in LDS -- the mode in which the path is bound by the decode and by HBM,
            # not by the filter tables' LDS gathers
            r = result["roofline"]
            per_pass = 1 if nq <= 1 else 2 if nq <= 2 else 4
            passes = -(-nq // per_pass)
            # which kernels the stream pass's launches were (dpq_profile counts them): named as they ran, all of them
            # when a call's levels mix
            ran = {"stream_kernel": float(all_stats[0, 13]) / aux_steps, "strand_kernel": float(all_stats[0, 14]) / aux_steps,
                   "strand1_kernel": float(all_stats[0, 15]) / aux_steps}
            names = [n for n, c in ran.items() if c > 0]
            strands = any(n.startswith("strand") for n in names)
            alg = (passes * alg_bytes_total) / (scan_ms_step * 1e-3) / 1e9 if scan_ms_step > 0 else 0.0
            r.update({"bound": "hbm", "kernel": " + ".join(names) if names else "stream_kernel", "stream_pass_launches_per_step": ran,
                      "achieved": alg, "peak": HBM_PEAK_GBPS * world,
                      "frac": alg / (HBM_PEAK_GBPS * world),
                      "definition": "stream pass, %d quer%s per pass, %d pass(es): achieved = passes x DTC payload bytes (SURVEY.md 8(d): every "
                                    "pass streams the compressed index once) / kernel time of the pass's launches (HIP events on the launch "
                                    "stream); peak = 8 TB/s HBM3E.  `hbm` holds the physical bytes of the counters where a matching PMC "
                                    "summary is committed." % (per_pass, "y" if per_pass == 1 else "ies", passes),
                      "queries_per_pass": per_pass, "passes": passes})
            r.pop("lds_gather_bytes_per_step", None)
            r["algorithmic_hbm"]["note"] = ("queries x DTC payload bytes / kernel time: with %d quer%s per pass this is %d x the bytes the "
                                            "passes stream" % (per_pass, "y" if per_pass == 1 else "ies", per_pass))
            result["config"]["decode"] = ("lane per run of 64 nodes over the strand image (%s)" % " + ".join(names) if strands else
                                          "wavefront per 64-node chunk (stream_kernel)") + ", once per pass of %d quer%s" % (per_pass, "y" if per_pass == 1 else "ies")
            result["config"]["queries_per_decode_pass"] = per_pass
        if replicas:
            result["query_replicas"] = replicas
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(wl, args, batches_np[0])
            result["vs_cpu_baseline_1_thread"] = result["value"] / result["cpu_baseline"]["value"]
        default_workload = (args.n, args.queries, args.topk, args.m, args.data) == (1_000_000, 1000, 100, 8, "pipeline")
        if world == 1 and default_workload and not args.no_hbm_leg:
            result["hbm_regime"] = hbm_leg()
        print(json.dumps(result))
    idx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
