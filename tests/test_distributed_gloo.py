"""World-size-2 test of the multi-GPU exchange logic on CPU (gloo): owner partition ->
all_to_all_single -> merge-unique per shard.  Device compute is replaced by numpy here (the oracle
produces each rank's local k-mers); the owner function is the numpy mirror of the device one
(checked against the device in tests/test_gpu_count.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spades_for_blackbird_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from tests.helpers import synth_reads
        reads = synth_reads(400, read_len=100, genome_len=3000, seed=21)
        mine = reads[rank::world]
        local = O.kmercount(mine, k, 16, 1)                       # this rank's local distinct k-mers
        own = D.owner_of(local, world)
        order = np.argsort(own, kind="stable")
        send = torch.from_numpy(local[order].view(np.int64).copy())
        counts = np.bincount(own, minlength=world)
        recv, rcl = D.exchange_by_owner(send, counts, local.shape[1])
        got = np.unique(recv.numpy().view(np.uint64), axis=0)     # merge-unique of the shard
        assert np.all(D.owner_of(got, world) == rank)
        q.put((rank, got))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,world", [(21, 2), (33, 2), (21, 3), (21, 8)])
def test_multi_rank_exchange(k, world):
    """world 3 also covers the rank whose own segment sits in the middle of its send buffer."""
    from oracle import oracle as O
    from tests.helpers import synth_reads
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    shards = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    reads = synth_reads(400, read_len=100, genome_len=3000, seed=21)
    full = O.kmercount(reads, k, 16, 1)
    exp = set(map(tuple, full.tolist()))
    sets = [set(map(tuple, shards[r].tolist())) for r in range(world)]
    for i in range(world):
        for j in range(i + 1, world):
            assert not (sets[i] & sets[j])
    assert set().union(*sets) == exp
    assert min(len(x) for x in sets) > 0.6 * len(exp) / world  # hash owner keeps shards balanced


def test_owner_of_is_a_partition():
    rng = np.random.default_rng(0)
    keys = rng.integers(0, 2 ** 62, size=(10000, 2), dtype=np.uint64)
    for n in (1, 2, 3, 8):
        o = D.owner_of(keys, n)
        assert o.min() >= 0 and o.max() < n
        assert np.bincount(o, minlength=n).min() > 10000 / n * 0.8
