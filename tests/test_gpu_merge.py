"""GPU: bbk_kmerset_from_device (merge of raw key arrays, the N>1 receive side) on key sets chosen to stress the
bucket kernels: heavy repeats of single keys and tight clusters of consecutive keys crowd single bins of the
distribution sort, which must hand the bucket to the radix kernel (second chance) and still produce the exact
sorted distinct set with summed counts.  Checked against numpy (integer work: bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _expect(keys, counts):
    # word order, word 0 most significant (reference adt/array_vector.hpp:114-123)
    order = np.lexsort(tuple(keys[:, w] for w in range(keys.shape[1] - 1, -1, -1)))
    ks, cs = keys[order], counts[order]
    head = np.ones(len(ks), dtype=bool)
    head[1:] = np.any(ks[1:] != ks[:-1], axis=1)
    idx = np.cumsum(head) - 1
    out_c = np.zeros(int(head.sum()), dtype=np.uint64)
    np.add.at(out_c, idx, cs.astype(np.uint64))
    return ks[head], out_c


@pytest.mark.parametrize("k,n_rand", [(21, 300_000), (31, 50_000), (55, 200_000), (33, 7_000)])
def test_merge_skewed_keys(k, n_rand):
    import torch
    import spades_for_blackbird_amd as B
    ctx = B.Context(0, stream=torch.cuda.current_stream())
    nw = B.engine.words(k)
    rng = np.random.default_rng(1234 + k)
    top_bits = 2 * k - 64 * (nw - 1)
    keys = rng.integers(0, 2**63, size=(n_rand, nw), dtype=np.uint64)
    keys[:, nw - 1] &= np.uint64((1 << top_bits) - 1) if top_bits < 64 else np.uint64(2**64 - 1)
    # one key repeated 3000 times, one 400 times
    rep = np.concatenate([np.repeat(keys[:1], 3000, axis=0), np.repeat(keys[1:2], 400, axis=0)])
    # clusters of consecutive keys (differ in the low bits of the LAST word only: same word 0 for wide keys)
    base = keys[2:3].copy()
    if top_bits >= 12:
        base[0, nw - 1] &= np.uint64(~np.uint64(0xFFF))
        clus = np.repeat(base, 1500, axis=0)
        clus[:, nw - 1] += np.arange(1500, dtype=np.uint64)
    else:  # too few populated bits in the last word (k=33: one base): repeat instead
        clus = np.repeat(base, 1500, axis=0)
    # and a cluster in word 0's low bits
    base0 = keys[3:4].copy()
    base0[0, 0] &= np.uint64(~np.uint64(0x3FF))
    if nw == 1 and top_bits < 64:
        base0[0, 0] &= np.uint64((1 << top_bits) - 1)
    clus0 = np.repeat(base0, 700, axis=0)
    clus0[:, 0] += np.arange(700, dtype=np.uint64)
    allk = np.concatenate([keys, rep, clus, clus0, keys[:5000]])
    allk = allk[rng.permutation(len(allk))]
    cnt = rng.integers(1, 50, size=len(allk)).astype(np.uint32)
    dk = torch.from_numpy(allk.view(np.int64)).cuda()
    dc = torch.from_numpy(cnt.view(np.int32)).cuda()
    torch.cuda.synchronize()
    s = ctx.kmerset_from_device(dk, len(allk), k, d_counts=dc)
    got_k, got_c = s.export(B.ORDER_SORTED, with_counts=True)
    exp_k, exp_c = _expect(allk, cnt)
    assert len(s) == len(exp_k)
    assert np.array_equal(got_k, exp_k)
    assert np.array_equal(got_c.astype(np.uint64), exp_c)
    # without counts: same distinct set
    s2 = ctx.kmerset_from_device(dk, len(allk), k)
    assert np.array_equal(s2.export(B.ORDER_SORTED), exp_k)
    ctx.close()


def test_k32_all_T_is_a_legal_key():
    """k=32: the k-mer T x 32 is the all-ones word, the value the LDS hash table of the 8-byte dedup kernel uses as its
    empty marker -- that kernel is only selected for 2k < 64, so the key must survive both the unsorted (hash) and the
    sorted merge of non-canonical input (bbk.h allows any k-mer records here)."""
    import torch
    import spades_for_blackbird_amd as B
    ctx = B.Context(0, stream=torch.cuda.current_stream())
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 2**63, size=(200_000, 1), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(200_000, 1), dtype=np.uint64)
    allT = np.full((3, 1), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    allk = np.concatenate([keys, allT, keys[:1000], np.zeros((2, 1), dtype=np.uint64)])
    allk = allk[rng.permutation(len(allk))]
    dk = torch.from_numpy(allk.view(np.int64)).cuda()
    torch.cuda.synchronize()
    exp = np.unique(allk[:, 0])
    s = ctx.kmerset_from_device(dk, len(allk), 32)
    got = s.export(B.ORDER_SORTED)[:, 0]
    assert np.array_equal(got, exp) and got[-1] == np.uint64(0xFFFFFFFFFFFFFFFF)
    u = ctx.kmerset_from_device(dk, len(allk), 32, flags=B.UNSORTED)
    uk, _ = u.export_by_owner(1)
    assert np.array_equal(np.sort(uk[:, 0]), exp)
    # k=31 (the widest key the hash-table kernel takes): T x 31 = 2^62 - 1 must survive too
    k31 = np.concatenate([keys & np.uint64((1 << 62) - 1), np.full((2, 1), (1 << 62) - 1, dtype=np.uint64)])
    d31 = torch.from_numpy(k31.view(np.int64)).cuda()
    torch.cuda.synchronize()
    u31 = ctx.kmerset_from_device(d31, len(k31), 31, flags=B.UNSORTED)
    g31, _ = u31.export_by_owner(1)
    assert np.array_equal(np.sort(g31[:, 0]), np.unique(k31[:, 0]))
    ctx.close()
