"""GPU: the N>1 sharded count end to end with two ranks sharing the one GPU of the test box (gloo for
the exchange, staged through the host).  The union of the two shards must be the single-rank result."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import spades_for_blackbird_amd as B
        from spades_for_blackbird_amd import distributed as D
        torch.cuda.set_device(0)
        ctx = B.Context(0, stream=torch.cuda.current_stream())
        reads = ctx.reads_synth(20000, read_len=150, genome_len=120000, seed_genome=42, seed_reads=43 + rank)
        # built in the final_kmers order (what bench.py --gpus N does); the ascending export sorts a copy
        shard = D.sharded_count(ctx, reads, 21, both_strands=True, reference_order=True)
        ptr, order = shard.device_keys()
        asc, ref = shard.export(B.ORDER_SORTED), shard.export(B.ORDER_REFERENCE_BUCKETS16)
        from oracle import oracle as O
        bk = np.array([O.bucket(asc[i], 16) for i in range(len(asc))])
        ref_ok = order == B.ORDER_REFERENCE_BUCKETS16 and np.array_equal(ref, asc[np.argsort(bk, kind="stable")])
        q.put((rank, asc, reads.to_list(), bool(ref_ok)))
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu():
    from oracle import oracle as O
    world = 2
    c = mp.get_context("spawn")
    q = c.Queue()
    port = _free_port()
    procs = [c.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (k, rd, ok)) for r, k, rd, ok in (q.get(timeout=300) for _ in range(world)))
    assert all(v[2] for v in res.values()), "a shard is not in the final_kmers order"
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res[0][0], res[1][0]
    sa, sb = set(int(x) for x in a[:, 0]), set(int(x) for x in b[:, 0])
    assert not (sa & sb)
    exp = O.kmercount(res[0][1] + res[1][1], 21, 16, 4)
    assert (sa | sb) == set(int(x) for x in exp[:, 0])
    assert len(sa) + len(sb) == len(exp)
