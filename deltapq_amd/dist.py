"""Multi-GPU sharding of the query path.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo"
in the CPU tests).  The index is cut into `world` contiguous DFS-position
ranges at segment boundaries, balanced by payload bytes (done by the C-ABI
transcoder: dpq_open_opts.shard_rank/shard_count).  Every rank answers the
whole query batch on its shard; the only exchange is ONE all-gather of the
per-shard partial top-k lists (nq * k * 8 bytes per rank), followed by a merge
by (distance, id).  No all-reduce, no per-query collective.
"""
import numpy as np

from . import api


def shard_ranges(payload, n_codes, world, M=8, chunks_per_segment=0):
    """[(node_lo, node_hi, algorithmic_bytes)] of every shard (host only)."""
    out = []
    for r in range(world):
        soa = api.HostSoA(payload, n_codes, M, shard_rank=r, shard_count=world,
                          chunks_per_segment=chunks_per_segment)
        out.append((soa.info["node_lo"], soa.info["node_hi"], soa.info["algorithmic_bytes"]))
    return out


def pack_lists(ids, dists):
    """[nq][k] int32 ids + [nq][k] float32 distances -> one [nq][2k] int32 tensor (distance bits as they are)."""
    import torch
    return torch.cat((ids.contiguous(), dists.contiguous().view(torch.int32)), dim=1).contiguous()


def unpack_lists(gathered, k):
    """[world][nq][2k] int32 -> ([world][nq][k] int32 ids, [world][nq][k] float32 distances)."""
    import torch
    return gathered[:, :, :k].contiguous(), gathered[:, :, k:].contiguous().view(torch.float32)


def gather_and_merge(ids, dists, group=None):
    """All-gather the per-shard partial top-k lists and merge them.

    ids/dists: [nq][k] torch tensors (int32 / float32) holding this rank's
    partial result with GLOBAL DFS positions (padding rows: id -1, dist +inf).
    CUDA tensors take the device path (RCCL all-gather + dpq_merge_topk_device),
    CPU tensors the host path (gloo all-gather + dpq_merge_topk_host).
    Returns (ids, dists) of the merged top-k on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return ids, dists
    nq, k = ids.shape
    if ids.is_cuda and dist.get_backend(group) != "gloo":
        # ONE collective: ids and distance bits travel together as [nq][2k] int32 (the exchange is
        # latency-bound: nq * k * 8 B per rank)
        gathered = torch.empty((world, nq, 2 * k), dtype=torch.int32, device=ids.device)
        dist.all_gather_into_tensor(gathered, pack_lists(ids, dists), group=group)
        return api.merge_topk_packed_torch(gathered, k)     # the merge kernel reads the packed rows as they are
    if ids.is_cuda:
        # rehearsal mode (gloo with device tensors, e.g. several ranks sharing one GPU):
        # exchange through host memory, merge on the device as the RCCL path does
        h_ids = torch.empty((world, nq, k), dtype=ids.dtype)
        h_dists = torch.empty((world, nq, k), dtype=dists.dtype)
        dist.all_gather(list(h_ids.unbind(0)), ids.cpu().contiguous(), group=group)
        dist.all_gather(list(h_dists.unbind(0)), dists.cpu().contiguous(), group=group)
        packed = torch.cat((h_ids, h_dists.view(torch.int32)), dim=2).contiguous().to(ids.device)   # the RCCL path's layout
        return api.merge_topk_packed_torch(packed, k)
    g_ids = torch.empty((world, nq, k), dtype=ids.dtype)
    g_dists = torch.empty((world, nq, k), dtype=dists.dtype)
    dist.all_gather(list(g_ids.unbind(0)), ids.contiguous(), group=group)
    dist.all_gather(list(g_dists.unbind(0)), dists.contiguous(), group=group)
    mi, md = api.merge_topk_host(g_ids.numpy(), g_dists.numpy())
    return torch.from_numpy(mi), torch.from_numpy(md)


def partial_topk_rows(all_ids, all_dists, k):
    """Helper for tests: the k best (distance, id) rows of arbitrary candidate
    arrays, padded with (-1, +inf) -- the shape a shard's partial list has."""
    order = np.lexsort((all_ids, all_dists.view(np.uint32)))[:k]
    ids = np.full(k, -1, dtype=np.int32)
    dists = np.full(k, np.inf, dtype=np.float32)
    ids[:len(order)] = all_ids[order]
    dists[:len(order)] = all_dists[order]
    return ids, dists
