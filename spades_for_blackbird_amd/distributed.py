"""Multi-GPU sharding of the k-mer set: one process per GPU, k-mers owned by
owner(key) = mulhi(mix64(key words), nranks); ONE all_to_all_single carries every distinct
canonical k-mer to its owner after the local count.  The collective goes through
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests);
everything else is the C ABI.
"""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_SEED = np.uint64(0x9E3779B97F4A7C15)


def owner_mix(keys):
    """numpy mirror of csrc/kmer_ops.h owner_mix (keys: uint64[n, W])."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    x = np.full(keys.shape[0], _SEED, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for i in range(keys.shape[1]):
            x = x ^ keys[:, i]
            x = x ^ (x >> np.uint64(30))
            x = x * _M1
            x = x ^ (x >> np.uint64(27))
            x = x * _M2
            x = x ^ (x >> np.uint64(31))
    return x


def owner_of(keys, nranks):
    """owner rank of every key: (mix * nranks) >> 64."""
    h = owner_mix(keys)
    hi, lo = h >> np.uint64(32), h & np.uint64(0xFFFFFFFF)
    n = np.uint64(nranks)
    # (h * n) >> 64 with 32-bit limbs (nranks < 2^32)
    t = (lo * n) >> np.uint64(32)
    return ((hi * n + t) >> np.uint64(32)).astype(np.int64)


def exchange_by_owner(send, send_counts, words, group=None):
    """send: int64 tensor [n, words] grouped by owner; send_counts: per-owner record counts.
    Returns (recv tensor [m, words], recv_counts list).  Two collectives: the tiny count matrix and
    the payload (the only data-path collective)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = send.device
    if dist.get_backend(group) == "gloo" and send.is_cuda:
        # gloo has no device all_to_all: stage through the host (tests of the sharded path on one GPU)
        recv, rcl = exchange_by_owner(send.cpu(), send_counts, words, group)
        return recv.to(dev), rcl
    rank = dist.get_rank(group)
    sc = torch.tensor([int(x) for x in send_counts], dtype=torch.int64, device=dev)
    rc = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(rc, sc, group=group)
    rcl = [int(x) for x in rc.tolist()]
    send = send.contiguous()
    # The rank's own segment never enters the collective: it is copied on the device, and the collective carries
    # the 7 peer segments only (one message per xGMI link).  The received records are a set, so the own segment
    # simply goes last.  (A one-rank RCCL all_to_all of the whole buffer returned before the data had arrived.)
    own = int(send_counts[rank])
    before = sum(int(x) for x in send_counts[:rank])
    n_peers = sum(rcl) - rcl[rank]
    recv = torch.empty((n_peers + own, words), dtype=torch.int64, device=dev)
    out_split = [x * words for x in rcl]
    in_split = [int(x) * words for x in send_counts]
    out_split[rank] = 0
    in_split[rank] = 0
    # input = the send buffer without the own segment (views, no copy, when the own segment is first or last)
    if own == 0:
        src = send.view(-1)
    elif before == 0:
        src = send[own:].view(-1)
    elif before + own == send.shape[0]:
        src = send[:before].view(-1)
    else:
        src = torch.cat((send[:before], send[before + own:])).view(-1)
    if world > 1:
        dist.all_to_all_single(recv[:n_peers].view(-1), src, out_split, in_split, group=group)
    if own:
        recv[n_peers:].copy_(send[before:before + own])
    rcl_out = list(rcl)
    return recv, rcl_out


def sharded_count(ctx, reads, k, both_strands=True, group=None, reference_order=False):
    """Count on every rank, exchange distinct canonical k-mers by owner, merge-unique the shard and
    (optionally) expand it to both strands.  Returns the rank's KMerSet shard."""
    import torch
    import torch.distributed as dist
    from . import engine as E
    world = dist.get_world_size(group)
    nw = E.words(k)
    dev = torch.device("cuda", ctx.device)
    # local dedup only (hash-bucket order): the owner partition does not need sorted input
    c = ctx.count(reads, k, E.CANONICAL | E.UNSORTED)
    send = torch.empty((len(c), nw), dtype=torch.int64, device=dev)
    counts = c.export_by_owner(world, dst_keys=send)
    c.free()
    recv, rcl = exchange_by_owner(send, counts, nw, group)
    torch.cuda.current_stream().synchronize()
    # merge-unique of the shard; sorted only if it is the final product
    shard = ctx.kmerset_from_device(recv, sum(rcl), k, flags=E.UNSORTED if both_strands else 0)
    if not both_strands:
        return shard
    # reference_order: the shard is built in the final_kmers order (16 XXH3 buckets, ascending inside)
    both = shard.both_strands(E.REFERENCE_ORDER if reference_order else 0)
    shard.free()
    return both
