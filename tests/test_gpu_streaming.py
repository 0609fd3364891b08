"""GPU: the streaming entry points (bbk_count_begin / push / finish, bbk_extindex_begin / push / finish) and the
range passes of both stages.

What the reference gets from bounded cells + repeated DumpBuffers rounds + MergeKMers
(common/utils/kmer_mph/kmer_splitter.hpp:73-167, kmer_index_builder.hpp:281-365) must give the same bytes as one
pass over everything: the same reads pushed in >= 3 batches (a merge forced after every push, every stage forced
into range passes by a tiny BBK_PASS_LIMIT) have to produce byte-identical final_kmers and extension index to the
one-shot calls and to the oracle, for all four key widths.  The knobs are read once per process, hence subprocesses.
"""
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(bool(os.environ.get("BBK_DISABLE_MSD")), reason="tests of the MSD path's modes")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import numpy as np, sys
sys.path.insert(0, %(root)r)
import spades_for_blackbird_amd as B
from oracle import oracle as O
from tests.helpers import synth_reads
ctx = B.Context(0)
reads = synth_reads(%(n)d, read_len=150, genome_len=%(g)d, sub_rate=0.01, seed=11, n_rate=0.002)
reads += ["A" * 150] * 40 + ["ACGT" * 37] * 30 + ["AC"] + ["ACGTN" * 30] + [""]
cuts = [0, len(reads) // 5, len(reads) // 2, len(reads) - 7, len(reads)]
import re
runs = [max(re.findall("[ACGTacgt]+", x) or [""], key=len) for x in reads]   # LongestValid
def n_inst(k):
    return sum(max(0, len(x) - k + 1) for x in runs)
for k in %(ks)r:
    whole = ctx.reads_from_ascii(reads)
    exp, ec = O.kmercount(reads, k, 16, 2, with_counts=True)
    # one shot (its stages may themselves run in range passes under the forced limits)
    one, oc = ctx.count(whole, k, B.BOTH_STRANDS | B.WITH_COUNTS).export(B.ORDER_REFERENCE_BUCKETS16, with_counts=True)
    assert np.array_equal(one, exp) and np.array_equal(oc, ec), ("one-shot", k)
    # pushed in 4 batches, with counts
    c = ctx.counter(k, B.BOTH_STRANDS | B.WITH_COUNTS)
    for a, b in zip(cuts[:-1], cuts[1:]):
        r = ctx.reads_from_ascii(reads[a:b])
        c.push(r)
        r.free()
    assert c.instances == n_inst(k), ("instances", k)
    s = c.finish()
    got, gc = s.export(B.ORDER_REFERENCE_BUCKETS16, with_counts=True)
    assert np.array_equal(got, exp) and np.array_equal(gc, ec), ("pushed+counts", k)
    s.free()
    # pushed as ASCII, final_kmers order built in place, no counts (the spades-kmercount configuration)
    c = ctx.counter(k, B.BOTH_STRANDS | B.REFERENCE_ORDER)
    for a, b in zip(cuts[:-1], cuts[1:]):
        c.push_ascii(reads[a:b])
    s = c.finish()
    assert np.array_equal(s.export(B.ORDER_REFERENCE_BUCKETS16), exp), ("pushed ref order", k)
    assert s.instances == 2 * n_inst(k), ("instances both strands", k)
    s.free()
    # canonical, ascending
    c = ctx.counter(k, B.CANONICAL | B.WITH_COUNTS)
    for a, b in zip(cuts[:-1], cuts[1:]):
        r = ctx.reads_from_ascii(reads[a:b]); c.push(r); r.free()
    s = c.finish()
    ck, cc = s.export(B.ORDER_SORTED, with_counts=True)
    one_c = ctx.count(whole, k, B.CANONICAL | B.WITH_COUNTS)
    ok, occ = one_c.export(B.ORDER_SORTED, with_counts=True)
    assert np.array_equal(ck, ok) and np.array_equal(cc, occ), ("canonical", k)
    if k %% 2 == 1 and k + 1 < 128:
        ox = O.ExtIndex(reads, k, 1)
        order = np.lexsort([ox.kmers[:, j] for j in range(ox.kmers.shape[1] - 1, -1, -1)])
        x1 = ctx.extindex(whole, k)
        k1, m1 = x1.export()
        assert np.array_equal(k1, ox.kmers[order]) and np.array_equal(m1, ox.masks[order]), ("ext one-shot", k)
        xb = ctx.extbuilder(k)
        for a, b in zip(cuts[:-1], cuts[1:]):
            r = ctx.reads_from_ascii(reads[a:b]); xb.push(r); r.free()
        x2 = xb.finish()
        k2, m2 = x2.export()
        assert np.array_equal(k2, k1) and np.array_equal(m2, m1), ("ext pushed", k)
    whole.free()
print("STREAMING-OK")
"""


def _run(env_extra, n, g, ks):
    env = dict(os.environ, BBK_VERBOSE="1", **env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "n": n, "g": g, "ks": ks}], capture_output=True,
                       text=True, env=env, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "STREAMING-OK" in r.stdout
    return r.stderr


def test_pushed_batches_equal_one_shot_all_widths():
    """merge after every push; default pass limits"""
    _run({"BBK_MERGE_MIN": "0"}, 2500, 15000, (21, 55, 77, 127))


def test_pushed_batches_with_forced_range_passes():
    """every stage in range passes: hash ranges from reads (stage A) and from key arrays (the merges), key ranges
    in stage B (KEYS prefix: tagged 8-byte keys and the extension index; REF prefix: 16-byte keys)"""
    err = _run({"BBK_MERGE_MIN": "0", "BBK_PASS_LIMIT": "30000"}, 2500, 15000, (21, 32, 55, 77))
    assert "hash ranges" in err and "key ranges" in err
    assert "prefix mode 2" in err, "the REF-prefix range passes (16-byte keys, final_kmers order) did not run"


def test_range_passes_with_slot_mode():
    """forced range passes AND the histogram-free slot modes on small inputs: the materialised ranges of stage B (level-0
    partition of the both-strand set) take the key slots of the ordering pass -- and whatever does not hold there (this
    input has poly-A and ACGT-repeat reads: an uneven key space) goes back to the exact mode, range by range"""
    err = _run({"BBK_MERGE_MIN": "0", "BBK_PASS_LIMIT": "30000", "BBK_SLOTS_MIN": "0"}, 2500, 15000, (21, 32, 55))
    assert "level 0" in err, "no level-0 partition ran"
    assert "msd key slots" in err, "the key slots were never tried"


def test_slot_mode_with_pushed_batches():
    """the histogram-free slot mode + spill reprocessing on every pushed batch and on the merges"""
    _run({"BBK_MERGE_MIN": "0", "BBK_SLOTS_MIN": "0"}, 3000, 20000, (21, 33))


def test_narrow_records_all_k():
    """stage A with 4-byte records between the partition levels (msd.hip "narrow records", 17 <= k <= 21): every k of
    that range (2k - 32 = 2 .. 10 key bits carried by the segment), with multiplicities, mask payloads (extension
    index, odd k), heavy repeats that overflow slots, and k = 16 / 22 on either side of the range"""
    err = _run({"BBK_MERGE_MIN": "0", "BBK_SLOTS_MIN": "0"}, 3000, 20000, (16, 17, 18, 19, 20, 21, 22))
    assert "narrow" in err


def test_table_give_up_on_half_empty_slots_with_poisoned_memory():
    """Slot mode, a flagged bucket whose slot is NOT full (its LDS table gave up: BBK_HASH_MAX_PROBES=0 makes every
    collision a give-up), device memory poisoned (BBK_POOL_POISON: every allocation is filled with 0xCD, so the unwritten
    tail of a slot holds plausible-looking garbage keys): only the records the level-2 cursor says were written may be
    reprocessed -- the round-1 code copied the whole slot and turned stale memory into k-mers.  Counts and mask payloads
    against the oracle, narrow (k=21) and 8-byte (k=25) hash kernels."""
    err = _run({"BBK_MERGE_MIN": "0", "BBK_SLOTS_MIN": "0", "BBK_HASH_MAX_PROBES": "0", "BBK_POOL_POISON": "1"}, 2500, 15000,
               (21, 25))
    assert any("over_bkt=" in l and "over_bkt=0" not in l for l in err.splitlines() if "msd slots" in l), \
        "no bucket was flagged: the give-up path did not run"


def _superk_lines(err):
    return [l for l in err.splitlines() if "superk" in l]


def test_superk_records_all_wide_k():
    """stage A through super-k-mer records (superk.hip; 16-, 24- and 32-byte keys): every pushed batch and the one-shot calls,
    multiplicities and mask payloads (extension index) against the oracle.  k = 33 / 64 / 65 / 96 / 97 / 127 are the ends
    of the three key widths (k = 64: no padding bits in the second word; k = 96: the window is longer than a segment;
    k = 127: records of 24 k-mers in five words).  Poisoned memory: slots and tables may only hold what the kernels wrote."""
    err = _run({"BBK_MERGE_MIN": "0", "BBK_SUPERK_MIN": "0", "BBK_POOL_POISON": "1"}, 2500, 15000,
               (33, 55, 63, 64, 65, 77, 96, 97, 127))
    lines = _superk_lines(err)
    assert any("distinct" in l for l in lines), "the super-k-mer path did not run"
    assert not any("declines" in l for l in lines), lines[:5]


def test_superk_records_in_hash_range_passes():
    """the same with 40 buckets per pass: several passes over ranges of the minimizer hash, the output regrown"""
    err = _run({"BBK_MERGE_MIN": "0", "BBK_SUPERK_MIN": "0", "BBK_SUPERK_BUCKETS": "40"}, 2500, 15000, (55, 77))
    lines = _superk_lines(err)
    assert any("passes=" in l and "passes=1 " not in l for l in lines), lines[:5]
    assert not any("declines" in l for l in lines), lines[:5]


def test_superk_second_chance_table():
    """buckets planned at 1.5x the first table's slots: most of them are given up and finished by the second-chance
    geometry (8192 slots, 1024 threads)"""
    err = _run({"BBK_MERGE_MIN": "0", "BBK_SUPERK_MIN": "0", "BBK_SUPERK_FILL": "1.5"}, 2500, 15000, (55, 77))
    lines = _superk_lines(err)
    assert any("second-chance table" in l for l in lines), lines[:5]
    assert not any("declines" in l for l in lines), lines[:5]


@pytest.mark.parametrize("k,meta", [(55, False), (77, False), (55, True), (33, True)])
def test_superk_equals_kmer_path_at_size(k, meta, monkeypatch):
    """3 M x 150 bp (0.29 G instances: the super-k-mer path's default threshold is 4 M, its buckets, slots and tables are
    the planned sizes, several thousand buckets go through the second chance with a second batch planned for the first
    one's multiplicity): the canonical set with multiplicities and the extension index (keys + edge masks) must be
    identical, element by element, to what the k-mer path (BBK_NO_SUPERK=1) gives -- compared on the device."""
    import torch
    import spades_for_blackbird_amd as B
    ctx = B.Context(0, stream=torch.cuda.current_stream())
    n_reads, L = 3_000_000, 150
    if meta:  # the skewed metagenome of BASELINE configs[4]: 200 genomes, log-normal abundances (coverage 0.01x .. 1000x)
        r = ctx.reads_synth_meta(n_reads, read_len=L)
    else:
        r = ctx.reads_synth(n_reads, read_len=L, genome_len=n_reads * L // 20, seed_genome=7, seed_reads=8)
    W = (k + 31) // 32

    def run():
        out = []
        for _ in range(2):  # the second call is planned from the multiplicity the first one saw
            s = ctx.count(r, k, B.CANONICAL | B.WITH_COUNTS)
            n = len(s)
            assert s.instances == n_reads * (L - k + 1)
            keys = torch.empty((n, W), dtype=torch.int64, device="cuda")
            cnt = torch.empty(n, dtype=torch.int32, device="cuda")
            s.export_to(keys, B.ORDER_SORTED, cnt)
            s.free()
            out.append((keys, cnt))
        x = ctx.extindex(r, k)
        xk = torch.empty((len(x), W), dtype=torch.int64, device="cuda")
        xm = torch.empty(len(x), dtype=torch.int32, device="cuda")
        x.export_to_u32(xk, xm)
        return out, xk, xm

    ctx.profile(True)
    ctx.profile_reset()
    a, ak, am = run()
    assert ctx.profile_get("k_sk_dedup")["launches"] >= 3, "the super-k-mer path did not run"
    assert ctx.profile_get("stat_superk_declined")["launches"] == 0
    monkeypatch.setenv("BBK_NO_SUPERK", "1")
    ctx.profile_reset()
    b, bk, bm = run()
    assert ctx.profile_get("k_sk_dedup")["launches"] == 0
    for (k1, c1), (k2, c2) in zip(a, b):
        assert k1.shape == k2.shape and bool(torch.equal(k1, k2)) and bool(torch.equal(c1, c2))
        assert int(c1.sum(dtype=torch.int64).item()) == n_reads * (L - k + 1)
    assert bool(torch.equal(a[0][0], a[1][0])) and bool(torch.equal(a[0][1], a[1][1]))
    assert ak.shape == bk.shape and bool(torch.equal(ak, bk)) and bool(torch.equal(am, bm))
    assert bool(torch.equal(ak, a[0][0]))  # the extension index holds exactly the canonical k-mers
    ctx.close()


def test_superk_gives_up_and_the_kmer_path_takes_over():
    """buckets planned 64x over the table: the first chance lists them, the second chance fails too, the call is
    given up and the batch is run on the k-mer path -- same results, and the next batch starts from scratch"""
    err = _run({"BBK_MERGE_MIN": "0", "BBK_SUPERK_MIN": "0", "BBK_SUPERK_FILL": "64"}, 2500, 15000, (55,))
    lines = _superk_lines(err)
    assert any("declines" in l for l in lines), lines[:5]


def test_superk_spilled_slots_and_hot_buckets_take_the_kmer_path():
    """level-1 slots far below their load: the records that do not fit go to the spill list, the buckets they belong
    to are marked hot and skipped by both tables, and everything of those buckets (slot part + spilled part) is expanded
    into k-mers and deduplicated by the k-mer path, then appended -- the batch is NOT given up (the allowance for the
    k-mer path's share is lifted for the test)"""
    err = _run({"BBK_MERGE_MIN": "0", "BBK_SUPERK_MIN": "0", "BBK_SUPERK_SLOT_SCALE": "0.12", "BBK_SUPERK_FALLBACK_MAX": "4"},
               2500, 15000, (55, 77, 127))
    lines = _superk_lines(err)
    assert any("spilled from full level-1 slots" in l for l in lines), lines[:5]
    assert any("records to the k-mer path" in l for l in lines), lines[:5]
    assert not any("declines" in l for l in lines), [l for l in lines if "declines" in l][:3]


def test_superk_buckets_beyond_both_tables_take_the_kmer_path():
    """buckets planned 8x over the first table: most of them overflow the second-chance table too and are finished by
    the k-mer path, bucket by bucket"""
    err = _run({"BBK_MERGE_MIN": "0", "BBK_SUPERK_MIN": "0", "BBK_SUPERK_FILL": "8", "BBK_SUPERK_FALLBACK_MAX": "4"},
               2500, 15000, (55, 96))
    lines = _superk_lines(err)
    assert any("records to the k-mer path" in l and " 0 spilled" in l for l in lines), lines[:5]
    assert not any("declines" in l for l in lines), [l for l in lines if "declines" in l][:3]


def test_superk_low_complexity_reads(monkeypatch, capfd):
    """250 k reads of a random genome + 50 k reads that share a 35 bp poly-A stretch between random flanks: the
    poly-A 24-mer is the minimizer of ~a million distinct 55-mers -- one level-1 slot runs over (spill list), one bucket
    is far beyond both tables.  Default thresholds: the batch must not be given up, the hot bucket goes to the k-mer
    path, and the result equals the k-mer path's (BBK_NO_SUPERK=1) element by element."""
    import random
    import torch
    import spades_for_blackbird_amd as B
    rnd = random.Random(5)
    genome = "".join(rnd.choice("ACGT") for _ in range(1_500_000))
    reads = []
    for _ in range(250_000):
        p = rnd.randrange(len(genome) - 150)
        reads.append(genome[p:p + 150])
    for _ in range(50_000):
        reads.append("".join(rnd.choice("ACGT") for _ in range(40)) + "A" * 35 + "".join(rnd.choice("ACGT") for _ in range(75)))
    rnd.shuffle(reads)
    k, W = 55, 2
    ctx = B.Context(0, stream=torch.cuda.current_stream())
    r = ctx.reads_from_ascii(reads)
    monkeypatch.setenv("BBK_VERBOSE", "1")

    def run():
        s = ctx.count(r, k, B.CANONICAL | B.WITH_COUNTS)
        keys = torch.empty((len(s), W), dtype=torch.int64, device="cuda")
        cnt = torch.empty(len(s), dtype=torch.int32, device="cuda")
        s.export_to(keys, B.ORDER_SORTED, cnt)
        x = ctx.extindex(r, k)
        xk = torch.empty((len(x), W), dtype=torch.int64, device="cuda")
        xm = torch.empty(len(x), dtype=torch.int32, device="cuda")
        x.export_to_u32(xk, xm)
        return keys, cnt, xk, xm

    a = run()
    err = capfd.readouterr().err
    lines = [l for l in err.splitlines() if "superk" in l]
    assert any("records to the k-mer path" in l for l in lines), lines[:6]
    assert not any("declines" in l for l in lines), [l for l in lines if "declines" in l][:3]
    monkeypatch.setenv("BBK_NO_SUPERK", "1")
    b = run()
    for x, y in zip(a, b):
        assert x.shape == y.shape and bool(torch.equal(x, y))
    assert int(a[1].sum(dtype=torch.int64).item()) == len(reads) * (150 - k + 1)
    ctx.close()


def test_one_call_is_cut_into_read_pieces():
    """8-byte keys: a call above one stage-A pass is cut into pieces of reads (each deduplicated on its own, merged
    once) instead of hash ranges that re-extract every k-mer per range; forced here with 60 k bases per piece"""
    err = _run({"BBK_READ_CHUNK": "60000"}, 2500, 15000, (16, 21, 31, 32))
    assert "pieces of" in err


@pytest.mark.parametrize("k", [33, 55, 97])
def test_superk_ragged_reads(k, monkeypatch, capfd):
    """reads of very different lengths through the super-k-mer path: 20 kbp reads (hundreds of segments each, tiles that
    lie inside one read), runs of reads shorter than k (no segment at all) between them, reads of exactly k and k+1
    bases, a read that is one long homopolymer; counts and extension index against the oracle"""
    import random
    import numpy as np
    import spades_for_blackbird_amd as B
    from oracle import oracle as O
    rnd = random.Random(k)
    rs = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))
    long_reads = [rs(20_000) for _ in range(4)]
    reads = []
    for lr in long_reads:
        reads.append(lr)
        reads += [rs(rnd.randrange(1, k)) for _ in range(300)]           # no k-mer at all
        reads += [lr[p:p + 150] for p in range(0, 19_000, 97)]            # overlap the long read: duplicates
    reads += [rs(k), rs(k + 1), "C" * 700, "", "ACGTN" * 40, rs(5_000) + "N" + rs(30)]
    monkeypatch.setenv("BBK_SUPERK_MIN", "0")
    monkeypatch.setenv("BBK_VERBOSE", "1")
    ctx = B.Context(0)
    r = ctx.reads_from_ascii(reads)
    exp, ec = O.kmercount(reads, k, 16, 2, with_counts=True)
    got, gc = ctx.count(r, k, B.BOTH_STRANDS | B.WITH_COUNTS).export(B.ORDER_REFERENCE_BUCKETS16, with_counts=True)
    assert np.array_equal(got, exp) and np.array_equal(gc, ec)
    ox = O.ExtIndex(reads, k, 1)
    order = np.lexsort([ox.kmers[:, j] for j in range(ox.kmers.shape[1] - 1, -1, -1)])
    xk, xm = ctx.extindex(r, k).export()
    assert np.array_equal(xk, ox.kmers[order]) and np.array_equal(xm, ox.masks[order])
    err = capfd.readouterr().err
    assert "superk:" in err and "distinct" in err and "declines" not in err
    ctx.close()


@pytest.mark.parametrize("k", [21, 33, 55])
def test_count_and_extindex_from_one_pass(k):
    """bbk_count_extindex / bbk_extindex_finish_with_set (BASELINE configs[2]: count + DeBruijnExtensionIndex of the same
    reads): one stage A with the mask payload; results equal the two separate calls and the oracle -- including reads of
    length exactly k, whose k-mers the index drops (no extension bit, kmer_splitters.hpp:160-180) and spades-kmercount
    keeps."""
    import numpy as np
    import spades_for_blackbird_amd as B
    from oracle import oracle as O
    from tests.helpers import synth_reads
    ctx = B.Context(0)
    reads = synth_reads(2500, read_len=120, genome_len=9000, sub_rate=0.01, seed=40 + k, n_rate=0.002)
    reads += ["ACGTTGCA" * 20][0:1] + [("ACGTTGCATTGACCA" * 10)[:k], ("TTGACCAGGTACCAT" * 10)[:k], "", "ACG"]
    r = ctx.reads_from_ascii(reads)
    s0 = ctx.count(r, k, B.BOTH_STRANDS | B.REFERENCE_ORDER).export(B.ORDER_REFERENCE_BUCKETS16)
    k0, m0 = ctx.extindex(r, k).export()
    s1, x1 = ctx.count_extindex(r, k, B.BOTH_STRANDS | B.REFERENCE_ORDER)
    assert np.array_equal(s1.export(B.ORDER_REFERENCE_BUCKETS16), s0)
    k1, m1 = x1.export()
    assert np.array_equal(k1, k0) and np.array_equal(m1, m0)
    assert np.array_equal(s0, O.kmercount(reads, k, 16, 2))
    # streamed in three pushes
    xb = ctx.extbuilder(k)
    for part in (reads[:900], reads[900:1800], reads[1800:]):
        xb.push(ctx.reads_from_ascii(part))
    s2, x2 = xb.finish_with_set(B.BOTH_STRANDS)
    assert np.array_equal(s2.export(B.ORDER_REFERENCE_BUCKETS16), s0)
    k2, m2 = x2.export()
    assert np.array_equal(k2, k0) and np.array_equal(m2, m0)
    ctx.close()
