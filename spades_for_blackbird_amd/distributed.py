"""Multi-GPU sharding of the k-mer set and of the extension index: one process per GPU, k-mers owned
by owner(key) = mulhi(mix64(key words), nranks); ONE all_to_all carries every distinct canonical k-mer -- with its
multiplicity or its InOutMask bits when asked -- to its owner after the local count.  The collective goes through
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests); everything else is the
C ABI.

Message size.  RCCL 2.26.6 (the build torch 2.10+rocm7.0 ships) silently drops the tail of a message above 1 GiB on
its self-copy path: a single-rank all_to_all_single of n bytes delivers the first ~n/2 for every n > 2^30 and all of it
for n <= 2^30 (profiles/r02_rccl_selfcopy_1GiB_diag.log; deterministic, independent of streams and of who wrote the
buffer -- this is what round 1 saw as "returned before the data had arrived").  Two rules follow: the rank's own segment
never enters the collective (a device copy is also cheaper), and no message exceeds MAX_MSG_BYTES: larger segments
travel in several rounds of the same collective.
"""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_SEED = np.uint64(0x9E3779B97F4A7C15)

MAX_MSG_BYTES = 256 << 20  # per (source, destination) message of one round; far below the 1 GiB that RCCL mishandles


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


def _exchange_counts(send_counts, dev, group):
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sc = torch.tensor([int(x) for x in send_counts], dtype=torch.int64, device=dev)
    rc = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(rc, sc, group=group)
    # rounds: every rank must run the same number of collectives
    mx = torch.tensor([max([0] + [int(c) for i, c in enumerate(send_counts) if i != dist.get_rank(group)])],
                      dtype=torch.int64, device=dev)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    return [int(x) for x in rc.tolist()], int(mx.item())


def ensure_headroom(ctx, need_bytes, what=""):
    """The engine's arena keeps the device memory of its last calls mapped (that is what makes the next call fast);
    torch allocates the exchange buffers beside it.  Before a large torch allocation: if the driver's free memory does
    not cover it with 10 % to spare, the arena's free chunks are handed back (bbk_ctx_trim; re-mapping them later costs
    up to 30 ms/GiB, so this is not done unconditionally); if it still does not fit, say so instead of dying in the
    allocator."""
    import torch
    free, total = torch.cuda.mem_get_info(ctx.device)
    if free < need_bytes * 1.1:
        ctx.trim()
        torch.cuda.empty_cache()
        free, total = torch.cuda.mem_get_info(ctx.device)
        if free < need_bytes:
            st = ctx.memory_stats()
            raise MemoryError("rank needs %.1f GB for %s but the device has %.1f GB free (%.1f GB mapped by the engine's "
                              "arena after a trim, %.1f GB reserved by torch)"
                              % (need_bytes / 1e9, what or "the exchange", free / 1e9, st["mapped_now"] / 1e9,
                                 torch.cuda.memory_reserved(ctx.device) / 1e9))


def exchange_by_owner(send, send_counts, words, group=None, payload=None, max_msg_bytes=None):
    """send: int64 tensor [n, words] grouped by owner; send_counts: per-owner record counts; payload: optional int32
    tensor [n] that travels with the records (multiplicities or mask bits).
    Returns (recv [m, words], recv_counts list) or (recv, recv_payload, recv_counts) with a payload.  The received
    records are a multiset in no particular order (the own segment comes last)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = send.device
    if dist.get_backend(group) == "gloo" and send.is_cuda:
        # gloo has no device all_to_all: stage through the host (tests of the sharded path on one GPU)
        out = exchange_by_owner(send.cpu(), send_counts, words, group, None if payload is None else payload.cpu(),
                                max_msg_bytes)
        return tuple(x.to(dev) if hasattr(x, "to") else x for x in out)
    rank = dist.get_rank(group)
    cap_bytes = max_msg_bytes or MAX_MSG_BYTES
    cap = max(1, cap_bytes // (8 * words))  # records per message
    rcl, max_seg = _exchange_counts(send_counts, dev, group)
    rounds = max(1, -(-max_seg // cap))
    send = send.contiguous()
    sc = [int(x) for x in send_counts]
    starts = np.concatenate([[0], np.cumsum(sc)]).astype(np.int64)
    own = sc[rank]
    n_peers = sum(rcl) - rcl[rank]
    recv = torch.empty((n_peers + own, words), dtype=torch.int64, device=dev)
    rpay = None
    if payload is not None:
        payload = payload.contiguous()
        rpay = torch.empty(n_peers + own, dtype=payload.dtype, device=dev)
    off = 0
    for t in range(rounds):
        lo = t * cap
        in_cnt = [0 if j == rank else max(0, min(sc[j] - lo, cap)) for j in range(world)]
        out_cnt = [0 if j == rank else max(0, min(rcl[j] - lo, cap)) for j in range(world)]
        n_out = sum(out_cnt)
        pieces = [(int(starts[j]) + lo, in_cnt[j]) for j in range(world) if in_cnt[j]]
        # input of this round: the slices of the peer segments (views, no copy, when they happen to be adjacent)
        def gather(t2d):
            if not pieces:
                return t2d[:0]
            if all(pieces[i][0] + pieces[i][1] == pieces[i + 1][0] for i in range(len(pieces) - 1)):
                return t2d[pieces[0][0]:pieces[-1][0] + pieces[-1][1]]
            return torch.cat([t2d[a:a + c] for a, c in pieces])
        if world > 1:
            src = gather(send)
            dist.all_to_all_single(recv[off:off + n_out].view(-1), src.reshape(-1), [x * words for x in out_cnt],
                                   [x * words for x in in_cnt], group=group)
            if payload is not None:
                dist.all_to_all_single(rpay[off:off + n_out], gather(payload), out_cnt, in_cnt, group=group)
        off += n_out
    assert off == n_peers
    if own:
        a = int(starts[rank])
        recv[n_peers:].copy_(send[a:a + own])
        if payload is not None:
            rpay[n_peers:].copy_(payload[a:a + own])
    if payload is not None:
        return recv, rpay, list(rcl)
    return recv, list(rcl)


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
    # send + receive buffers (the shard is about as large as the local set) + the merge's working space
    ensure_headroom(ctx, 3 * len(c) * nw * 8, "the k-mer exchange")
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


def sharded_extindex(ctx, reads, k, group=None):
    """The extension index sharded by k-mer owner ("ext records follow their k-mer's owner, no second
    exchange"): every rank reduces its reads to (canonical k-mer, OR of mask bits) records, the records go to the
    k-mer's owner in the same all_to_all that carries the keys, and the owner ORs what it receives.  Returns this
    rank's ExtIndex shard (ascending canonical k-mers it owns + their complete InOutMask)."""
    import torch
    import torch.distributed as dist
    from . import engine as E
    world = dist.get_world_size(group)
    nw = E.words(k)
    dev = torch.device("cuda", ctx.device)
    c = ctx.count(reads, k, E.CANONICAL | E.UNSORTED | E.WITH_MASKS)
    ensure_headroom(ctx, 3 * len(c) * (nw * 8 + 4), "the extension-index exchange")
    send = torch.empty((len(c), nw), dtype=torch.int64, device=dev)
    pay = torch.empty(len(c), dtype=torch.int32, device=dev)
    counts = c.export_by_owner(world, dst_keys=send, dst_counts=pay)
    c.free()
    recv, rpay, rcl = exchange_by_owner(send, counts, nw, group, payload=pay)
    torch.cuda.current_stream().synchronize()
    return ctx.extindex_from_device(recv, rpay, sum(rcl), k)


def _p2p_chunks(n_rec, words, max_msg_bytes):
    cap = max(1, (max_msg_bytes or MAX_MSG_BYTES) // (8 * words))
    return [(a, min(n_rec, a + cap)) for a in range(0, n_rec, cap)] or []


def gather_extindex(ctx, shard, k, dst=0, group=None, max_msg_bytes=None):
    """The shards of the extension index gathered on rank `dst` for the unitig stage (the compaction
    walks across owners, so the (k-mer, mask) table goes to a GPU that holds it).  Point-to-point messages of at most
    MAX_MSG_BYTES; the shards are disjoint, so the merge on `dst` is one ordering pass.  Returns the complete ExtIndex
    on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    from . import engine as E
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nw = E.words(k)
    dev = torch.device("cuda", ctx.device)
    staged = dist.get_backend(group) == "gloo"
    n = len(shard)
    sizes = [torch.zeros(1, dtype=torch.int64, device="cpu" if staged else dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n], dtype=torch.int64, device="cpu" if staged else dev), group=group)
    sizes = [int(s.item()) for s in sizes]
    keys = torch.empty((n, nw), dtype=torch.int64, device=dev)
    masks = torch.empty(n, dtype=torch.int32, device=dev)
    shard.export_to_u32(keys, masks)
    torch.cuda.current_stream().synchronize()
    if rank != dst:
        kk, mm = (keys.cpu(), masks.cpu()) if staged else (keys, masks)
        for a, b in _p2p_chunks(n, nw, max_msg_bytes):
            dist.send(kk[a:b].reshape(-1), dst, group=group)
            dist.send(mm[a:b], dst, group=group)
        return None
    total = sum(sizes)
    allk = torch.empty((total, nw), dtype=torch.int64, device=dev)
    allm = torch.empty(total, dtype=torch.int32, device=dev)
    off = 0
    for src in range(world):
        m = sizes[src]
        if src == dst:
            allk[off:off + m].copy_(keys)
            allm[off:off + m].copy_(masks)
        else:
            for a, b in _p2p_chunks(m, nw, max_msg_bytes):
                if staged:
                    tk = torch.empty((b - a) * nw, dtype=torch.int64)
                    tm = torch.empty(b - a, dtype=torch.int32)
                    dist.recv(tk, src, group=group)
                    dist.recv(tm, src, group=group)
                    allk[off + a:off + b].copy_(tk.view(b - a, nw))
                    allm[off + a:off + b].copy_(tm)
                else:
                    dist.recv(allk[off + a:off + b].view(-1), src, group=group)
                    dist.recv(allm[off + a:off + b], src, group=group)
        off += m
    torch.cuda.current_stream().synchronize()
    return ctx.extindex_from_device(allk, allm, total, k)


def sharded_gfa(ctx, reads, k, path, dst=0, group=None):
    """spades-gbuilder --gfa over N ranks: sharded extension index -> gathered on `dst` -> unitigs + links -> GFA
    written by `dst`.  Returns the Unitigs object on dst (None elsewhere)."""
    import torch.distributed as dist
    shard = sharded_extindex(ctx, reads, k, group)
    full = gather_extindex(ctx, shard, k, dst, group)
    shard.free()
    if dist.get_rank(group) != dst:
        return None
    u = ctx.unitigs(full)
    u.write_gfa(path)
    full.free()
    return u
