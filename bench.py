#!/usr/bin/env python3
"""bench.py -- queries/s of the DeltaPQ `-task query` hot path on MI355X.

Workload (BASELINE.json configs[1]): SIFT1M-shaped index, N = 1e6 codes, m = 8,
k = 256, top-k = 100, 1000 queries per step, synthetic data (seeded DeltaTree
stream + SIFT-shaped codebook/queries, deltapq_amd/synth.py).  A "step" answers
the whole 1000-query batch: LUT build + delta-decode/ADC scan cascade + select.
Queries and results stay in HBM (device tensors) inside the timed region.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

With N > 1 the index is sharded by DFS-position range (byte-balanced), every
rank answers all queries on its shard, the per-shard partial top-k lists are
all-gathered over RCCL and merged on the GPU ("scaling": "strong": the
database and the query batch are fixed as N grows).

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
`roofline` (scan kernel, HIP-event timed inside the library on the launch
stream) and `cpu_baseline` (the oracle restatement timed on this host).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
PMC_SUMMARY = os.path.join(HERE, "profiles", "pmc_summary_default_workload.json")


def measured_traffic(args):
    """HBM bytes per scan-kernel launch from the committed rocprofv3 --pmc passes
    (scripts/collect_pmc.sh: FETCH_SIZE and WRITE_SIZE in separate runs, gfx950
    x2 read correction).  Only valid for the default workload it was taken on."""
    default = (args.n, args.queries, args.topk, args.m, args.data, args.gpus) == (1_000_000, 1000, 100, 8, "pipeline", 1)
    if not default or not os.path.exists(PMC_SUMMARY):
        return None, None
    with open(PMC_SUMMARY) as f:
        s = json.load(f)
    return s.get("scan_kernel_hbm_bytes_per_launch"), s.get("tag")


def make_queries(args, seed):
    from deltapq_amd import synth
    if args.data == "pipeline":
        return synth.make_clustered_vectors(args.queries, args.dim, seed=seed, n_clusters=20000, spread=12.0, centre_seed=7)
    return synth.make_queries(args.queries, args.dim, seed=seed)


def build_workload(args, device, query_seed=101):
    """`pipeline` (default): SIFT-shaped vectors (mixture of 20 000 Gaussians, values 0..218) ->
    k-means codebook -> PQ codes (GPU encoder) -> DeltaTree (host builder, reference method 1) -> DTC.
    `stream`: random DeltaTree emitted directly as (depth, mask, bytes) triples."""
    from deltapq_amd import api, synth
    t0 = time.time()
    if args.data == "pipeline":
        base = synth.make_clustered_vectors(args.n, args.dim, seed=100, n_clusters=20000, spread=12.0, centre_seed=7)
        queries = make_queries(args, query_seed)
        cb = synth.kmeans_codebook(base, args.m, 256, iters=6, seed=102)
        codes = api.encode_pq(base, cb, device=device)
        del base
        tree = api.DeltaTree(codes, codebook=cb, device=device)      # edge search on the GPU, layout on the host
        payload = tree.payload()
        n_bytes = len(payload)
        uniq = len(np.unique(codes.view("V%d" % args.m))) / args.n
        desc = "vectors->kmeans->PQ encode->built DeltaTree, %.1f%% unique codes" % (100 * uniq)
        tree.close()
    else:
        cb = synth.make_codebook(args.m, 256, args.dim // args.m, seed=100)
        queries = make_queries(args, query_seed)
        tree = synth.synth_tree(args.n, args.m, seed=102, mean_diffs=args.mean_diffs)
        payload, n_bytes = synth.encode_dtc(tree)
        desc = "random (depth, mask, bytes) stream"
    return dict(codebook=cb, queries=queries, payload=payload, n_bytes=n_bytes, gen_s=time.time() - t0, desc=desc)


def cpu_baseline(wl, args):
    """Oracle (CPU restatement of h:3731-3892) on a bounded sample of the same
    workload: 1 thread (the reference's parallelism) and all host cores."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import dtc_oracle as O
    orc = O.Oracle()
    cb, qs, payload = wl["codebook"], wl["queries"], wl["payload"]
    n1 = min(len(qs), args.cpu_queries)
    t0 = time.time()
    for i in range(n1):
        orc.query_in_memory(payload, args.n, cb, qs[i], args.topk)
    t1 = time.time() - t0
    cores = os.cpu_count() or 1
    nall = min(len(qs), max(n1, cores * 8))
    with ThreadPoolExecutor(cores) as ex:   # ctypes releases the GIL; one query per task
        t0 = time.time()
        list(ex.map(lambda i: orc.query_in_memory(payload, args.n, cb, qs[i], args.topk), range(nall)))
        tall = time.time() - t0
    return {"value": n1 / t1, "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": "first %d of the %d queries, full N=%d index, in-memory scan (query_im twin), %.1f s"
                      % (n1, len(qs), args.n, t1),
            "ms_per_query": 1e3 * t1 / n1,
            "all_cores": {"value": nall / tall, "cores": cores, "queries": nall}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", "--codes", dest="n", type=int, default=1_000_000,
                    help="codes in the index (under torch.distributed.run use --codes: its parser takes --n for its own)")
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--topk", type=int, default=100)
    ap.add_argument("--m", type=int, default=8)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--data", choices=["pipeline", "stream"], default="pipeline")
    ap.add_argument("--mean-diffs", type=float, default=3.0, help="changed bytes per node (--data stream)")
    ap.add_argument("--chunks-per-segment", type=int, default=0)
    ap.add_argument("--cpu-queries", type=int, default=700, help="queries timed on the CPU oracle (1 thread)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", type=int, default=4, help="queries verified against the oracle before timing")
    ap.add_argument("--shard", choices=["auto", "query", "index"], default="auto",
                    help="N > 1: 'query' = every GPU holds the whole index and answers its own batch of --queries "
                         "(independent units, no data-path collective, weak scaling); 'index' = the index is cut into "
                         "DFS ranges, every GPU answers the one batch on its range, one all-gather + merge (strong "
                         "scaling; what an index beyond one GPU's HBM needs); auto = query while the index fits")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share GPUs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from deltapq_amd import api

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
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cpu_coll = args.backend == "gloo"   # gloo collectives take host tensors
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if cpu_coll:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # N > 1 decomposition.  Queries are independent units: while the whole index fits one GPU (4.5 MB at
    # 1 M codes, 4.5 GB at 1 B codes, of 288 GB) every GPU keeps a replica and answers its own batch --
    # no data-path collective, per-GPU work fixed ("weak").  Index sharding (DFS ranges, one all-gather
    # of the partial lists + merge, "strong") is what larger indexes need; it is measured beside it.
    shard_mode = args.shard
    if shard_mode == "auto":
        shard_mode = "query" if (5 * args.n) < (64 << 30) else "index"
    weak = shard_mode == "query"   # the label the N > 1 runs of this mode carry; at N = 1 both modes are the same run
    by_query = weak and world > 1
    wl = build_workload(args, local_rank, query_seed=101 + (rank if by_query else 0))
    idx = api.DeltaPQIndex.open_memory(wl["payload"], args.n, args.m, 256, device=local_rank,
                                       shard_rank=0 if by_query else rank, shard_count=1 if by_query else world,
                                       chunks_per_segment=args.chunks_per_segment)
    idx.set_codebook(wl["codebook"])
    info = idx.info()
    q_dev = torch.from_numpy(wl["queries"]).to(dev)
    nq, k = args.queries, args.topk
    ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    dists = torch.empty((nq, k), dtype=torch.float32, device=dev)
    from deltapq_amd import dist as dpq_dist

    # Batches that no collective follows are pipelined (dpq_query_batch_device_async): a step enqueues its
    # batch, sync() settles them all (dpq_finish: waits, checks the overflow words, reruns if needed).
    pipelined = by_query or world == 1

    def step():
        idx.query_batch_torch(q_dev, k, ids, dists, wait=not pipelined)
        if pipelined:
            return ids, dists   # this rank's batch; complete after sync()
        # index shards: the path's one exchange step -- all-gather of the partial lists
        # (nq*k*8 B per rank) over RCCL, then the device merge
        return dpq_dist.gather_and_merge(ids, dists)

    def sync():
        idx.finish()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # correctness gate before timing (rank 0, a few queries, against the oracle)
    out_ids, out_dists = step()
    sync()
    parity = None
    if rank == 0 and args.check > 0:
        from oracle import dtc_oracle as O
        orc = O.Oracle()
        hi, hd = out_ids.cpu().numpy(), out_dists.cpu().numpy()
        parity = True
        for i in range(min(args.check, nq)):
            lut = orc.build_lut(wl["codebook"], wl["queries"][i])
            oi, od, alld, _ = orc.scan_lut(wl["payload"], args.n, lut, k, want_all=True)
            ok, msg = O.tie_aware_equal(hi[i], hd[i], oi, od, alld, args.n)
            if not ok:
                parity = False
                print("bench.py: PARITY FAILURE on query %d: %s" % (i, msg), file=sys.stderr)
        if not parity:
            sys.exit(4)

    for _ in range(args.warmup):
        step()
    # Timed region: HIP events bracket the scan launches only (the roofline kernel); an event pair
    # costs ~4 us of stream time, so the other kernels are timed in a short untimed pass afterwards.
    idx.profile_enable(0 if os.environ.get("DPQ_BENCH_NOPROF") else 2)
    idx.profile_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    prof = idx.profile_read()
    aux_steps = max(1, min(args.steps, 5))
    idx.profile_enable(1)
    idx.profile_reset()
    for _ in range(aux_steps):
        step()
    sync()
    prof_aux = idx.profile_read()
    idx.profile_enable(0)
    prof["exact_checks_per_query"] = prof_aux["exact_checks"] / max(1, aux_steps * args.queries)
    prof["candidates_per_query"] = prof_aux["candidates"] / max(1, aux_steps * args.queries)
    prof["select_ms"] = prof_aux["select_ms"] * args.steps / aux_steps
    prof["lut_ms"] = prof_aux["lut_ms"] * args.steps / aux_steps

    # N > 1 with replicas: also time the index-sharded decomposition of ONE batch (rank 0's queries), outside
    # the timed region, so both ways of using the GPUs are on record
    alt = None
    if world > 1 and by_query:
        q0 = torch.from_numpy(make_queries(args, 101)).to(dev)
        sidx = api.DeltaPQIndex.open_memory(wl["payload"], args.n, args.m, 256, device=local_rank, shard_rank=rank,
                                            shard_count=world, chunks_per_segment=args.chunks_per_segment)
        sidx.set_codebook(wl["codebook"])

        def sstep():
            sidx.query_batch_torch(q0, k, ids, dists)
            return dpq_dist.gather_and_merge(ids, dists)

        m_ids, _ = sstep()
        for _ in range(args.warmup):
            sstep()
        sync()
        ta = time.perf_counter()
        for _ in range(args.steps):
            sstep()
        sync()
        alt = time.perf_counter() - ta
        sidx.close()

    cdev = torch.device("cpu") if cpu_coll else dev
    t = torch.tensor([elapsed, alt or 0.0], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, alt = float(t[0].item()), float(t[1].item())
    # per-rank scan figures -> rank 0 (sum of algorithmic bytes, max of kernel time)
    stats = torch.tensor([prof["scan_ms"], float(prof["scan_launches"]), float(info["algorithmic_bytes"]),
                          float(info["device_bytes"]), prof["select_ms"], prof["lut_ms"]], dtype=torch.float64,
                         device=cdev)
    if world > 1:
        all_stats = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(all_stats, stats)
        all_stats = torch.stack(all_stats).cpu().numpy()
    else:
        all_stats = stats.cpu().numpy()[None, :]

    if rank == 0:
        steps = max(1, args.steps)
        scan_ms_step = float(all_stats[:, 0].max()) / steps              # slowest rank
        launches_step = float(all_stats[0, 1]) / steps
        alg_bytes_total = float(all_stats[:, 2].sum())   # index shards: n_bytes of the payload; replicas: world x n_bytes
        achieved = (nq * alg_bytes_total) / (scan_ms_step * 1e-3) / 1e9 if scan_ms_step > 0 else 0.0
        peak = HBM_PEAK_GBPS * world
        traffic, traffic_tag = measured_traffic(args)
        result = {
            "metric": "queries/sec, SIFT1M-shaped m=%d k=256 topk=%d" % (args.m, k),
            "value": (world if by_query else 1) * nq * steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "f64-sum-of-f32 (u8 code decode)",
            "data": "synthetic",
            "config": {
                "workload": "SIFT1M-shaped synthetic (%s): N=%d m=%d k=256 h=1 topk=%d, %d queries/step, "
                            "%.2f B/code, %.2f diffs/node" % (wl["desc"], args.n, args.m, k, nq, wl["n_bytes"] / args.n,
                                                              (wl["n_bytes"] - args.m) / args.n - (1.5 if args.m <= 8 else 2.5)),
                "n_codes": args.n, "queries_per_step": nq, "topk": k, "n_bytes": int(wl["n_bytes"]),
                "sharding": ("query replicas x%d: every GPU holds the whole index and answers its own %d-query batch"
                             % (world, nq)) if weak else "dfs-range index shards x%d, one batch" % world,
                "global_queries_per_step": (world if by_query else 1) * nq,
                "queries_per_decode_pass": 64 if args.m <= 8 else 16,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "scan_kernel",
                "achieved": achieved,
                "peak": peak,
                "unit": "GB/s",
                "frac": achieved / peak,
                "traffic": traffic,
                "traffic_note": ("HBM bytes per scan launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                 "passes (%s), 2x gfx950 read correction; compare with algorithmic_bytes_per_launch"
                                 % traffic_tag) if traffic else "not collected for this workload",
                "algorithmic_bytes_per_launch": nq * alg_bytes_total / launches_step if launches_step else None,
                "algorithmic_bytes_per_step": nq * alg_bytes_total,
                "launches_per_step": launches_step,
                "avg_launch_ms": scan_ms_step / launches_step if launches_step else None,
                "scan_ms_per_step": scan_ms_step,
                "select_ms_per_step": float(all_stats[:, 4].max()) / steps,
                "lut_ms_per_step": float(all_stats[:, 5].max()) / steps,
                "filter_survivors_per_query": prof["exact_checks_per_query"],
                "candidates_per_query": prof["candidates_per_query"],
                "event_note": "timed region: HIP events around the scan launches only; select/lut figures from %d "
                              "untimed steps run afterwards with events around every kernel" % aux_steps,
                "note": "achieved = queries x DTC payload bytes / scan-kernel time (HIP events on the launch "
                        "stream, all cascade levels of a step summed); each decoded chunk serves 64 queries (16 at m=16), so "
                        "physical traffic is a small fraction of this figure and frac can exceed 1 (see DESIGN.md)",
            },
            "parity_checked_queries": min(args.check, nq) if parity else 0,
            "index": {"device_bytes_rank0": int(all_stats[0, 3]), "segments_rank0": info["n_segments"],
                      "gen_seconds": wl["gen_s"]},
        }
        if alt:
            result["index_sharded"] = {
                "note": "same GPUs, the index cut into %d DFS ranges, ONE %d-query batch answered by all of them "
                        "(all-gather of the partial lists + merge); %d steps timed after the main region" % (world, nq, steps),
                "value": nq * steps / alt, "unit": "queries/s", "ms_per_step": 1e3 * alt / steps, "scaling": "strong"}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(wl, args)
        print(json.dumps(result))
    idx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
