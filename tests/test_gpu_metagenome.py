"""GPU: BASELINE configs[4] workload shape -- skewed-coverage synthetic metagenome (many genomes,
log-uniform lengths, log-normal(sigma=2) abundances), multi-k {21, 33, 55} -- on a small instance against the oracle,
plus the properties of the generator itself."""
import numpy as np
import pytest

import spades_for_blackbird_amd as B
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = B.Context(0)
    yield c
    c.close()


def test_generator_shape(ctx):
    r, gl, ab = ctx.reads_synth_meta(50_000, read_len=150, n_genomes=200, min_len=500_000, max_len=8_000_000, sigma=2.0,
                                     seed=44, want_community=True)
    assert len(r) == 50_000 and r.bases == 50_000 * 150
    assert gl.min() >= 500_000 and gl.max() <= 8_000_000
    assert np.log10(gl.max() / gl.min()) > 0.9                      # lengths spread over the whole log range
    assert np.log10(ab.max() / ab.min()) >= 3.0                     # >= 3 decades of coverage skew (sigma = 2)
    r2 = ctx.reads_synth_meta(50_000, seed=44)
    assert r.to_ascii()[0] == r2.to_ascii()[0]                      # deterministic in the seed
    r3 = ctx.reads_synth_meta(50_000, seed=45)
    assert r.to_ascii()[0] != r3.to_ascii()[0]


@pytest.mark.parametrize("k", [21, 33, 55])
def test_skewed_community_vs_oracle(ctx, k):
    """20 small genomes with abundances over three decades: the abundant ones pile their k-mers hundreds of times
    deep while the rare ones are singletons -- counts, final_kmers order and the extension index must equal the
    oracle bit for bit."""
    r = ctx.reads_synth_meta(6000, read_len=150, n_genomes=20, min_len=2000, max_len=30000, sigma=2.0, sub_rate=0.005,
                             seed=44)
    reads = r.to_list()
    got, gc = ctx.count(r, k, B.BOTH_STRANDS | B.WITH_COUNTS).export(B.ORDER_REFERENCE_BUCKETS16, with_counts=True)
    exp, ec = O.kmercount(reads, k, 16, 4, with_counts=True)
    assert np.array_equal(got, exp) and np.array_equal(gc, ec)
    assert gc.max() > 20 * np.median(gc)                            # the skew is really there
    ref = ctx.count(r, k, B.BOTH_STRANDS | B.REFERENCE_ORDER).export(B.ORDER_REFERENCE_BUCKETS16)
    assert np.array_equal(ref, exp)
    x = ctx.extindex(r, k)
    ox = O.ExtIndex(reads, k, 1)
    order = np.lexsort([ox.kmers[:, j] for j in range(ox.kmers.shape[1] - 1, -1, -1)])
    gk, gm = x.export()
    assert np.array_equal(gk, ox.kmers[order]) and np.array_equal(gm, ox.masks[order])


def test_skew_at_slot_mode_scale(ctx):
    """2 M reads of the full-shape community (200 genomes): large enough for the histogram-free slot mode of stage A;
    multiplicities must add up to the k-mer positions and the set must be strictly ascending, whatever overflowed."""
    r = ctx.reads_synth_meta(2_000_000, seed=44)
    ctx.profile(True)
    ctx.profile_reset()
    s = ctx.count(r, 21, B.CANONICAL | B.WITH_COUNTS)
    ctx.profile(False)
    st = {f: ctx.profile_get(f) for f in ("stat_slot_records", "stat_slot_spilled", "stat_slot_reprocessed")}
    assert st["stat_slot_records"]["launches"] >= 1, "the slot mode did not run"
    assert st["stat_slot_records"]["bytes"] == 2_000_000 * 130
    keys, cnt = s.export(B.ORDER_SORTED, with_counts=True)
    assert int(cnt.astype(np.uint64).sum()) == 2_000_000 * 130
    assert np.all(keys[1:, 0] > keys[:-1, 0])
    runs, eq, _ = s.verify_order()
    assert runs == 1 and eq == 0
