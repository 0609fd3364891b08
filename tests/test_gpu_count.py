"""GPU parity tests of the k-mer counting path (through the C ABI) against the CPU oracle and
the reference's golden vectors.  Bit-exact: integer/byte work."""
import hashlib
import os

import numpy as np
import pytest

import spades_for_blackbird_amd as B
from oracle import oracle as O
from tests.helpers import rc, read_fastq_gz, synth_reads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = B.Context(0)
    yield c
    c.close()


def gpu_final_kmers(ctx, reads, k, with_counts=False):
    r = ctx.reads_from_ascii(reads)
    s = ctx.count(r, k, B.BOTH_STRANDS | (B.WITH_COUNTS if with_counts else 0))
    return s.export(B.ORDER_REFERENCE_BUCKETS16, with_counts=with_counts)


def test_toy_md5(ctx, golden, golden_dir):
    g = golden["toy_kmercount"]
    reads = []
    for f in g["files"]:
        reads += read_fastq_gz(os.path.join(golden_dir, f))
    for k, key in ((21, "k21"), (55, "k55")):
        a = gpu_final_kmers(ctx, reads, k)
        assert len(a) == g[key]["n_kmers"]
        assert hashlib.md5(a.tobytes()).hexdigest() == g[key]["md5"]
    a = gpu_final_kmers(ctx, read_fastq_gz(os.path.join(golden_dir, g["files"][0])), 77)
    assert len(a) == g["k77"]["n_kmers_r1_only"]


def test_write_final_kmers(ctx, golden, golden_dir, tmp_path):
    g = golden["toy_kmercount"]
    reads = []
    for f in g["files"]:
        reads += read_fastq_gz(os.path.join(golden_dir, f))
    s = ctx.count(ctx.reads_from_ascii(reads), 21, B.BOTH_STRANDS)
    p = str(tmp_path / "final_kmers")
    s.write_final_kmers(p)
    with open(p, "rb") as f:
        assert hashlib.md5(f.read()).hexdigest() == g["k21"]["md5"]


@pytest.mark.parametrize("k", [1, 2, 4, 5, 16, 21, 22, 31, 32, 33, 55, 63, 64, 65, 77, 96, 97, 127])
def test_vs_oracle_all_k(ctx, k):
    reads = synth_reads(400, read_len=150, genome_len=3000, sub_rate=0.01, seed=k, n_rate=0.002)
    reads += ["", "A", "ACGT" * 40, "N" * 50, "acgtnACGTTGCA" * 12, "T" * 150]
    exp, expc = O.kmercount(reads, k, 16, 2, with_counts=True)
    got, gotc = gpu_final_kmers(ctx, reads, k, with_counts=True)
    assert got.shape == exp.shape
    assert np.array_equal(got, exp)
    assert np.array_equal(gotc, expc)


def test_sorted_order_and_canonical(ctx):
    reads = synth_reads(2000, read_len=100, genome_len=20000, sub_rate=0.005, seed=11)
    for k in (21, 33):
        r = ctx.reads_from_ascii(reads)
        s = ctx.count(r, k, B.CANONICAL | B.WITH_COUNTS)
        keys, cnt = s.export(B.ORDER_SORTED, with_counts=True)
        x = O.ExtIndex(reads, k - 1, 1)  # canonical k-mers = the (k-1)+1-mers of the ext index build
        exp = x.kp1
        order = np.lexsort([exp[:, j] for j in range(exp.shape[1] - 1, -1, -1)])
        assert np.array_equal(keys, exp[order])
        assert np.array_equal(cnt, x.kp1_count[order])
        assert s.instances == sum(max(0, len(t) - k + 1) for t in reads)


def test_empty_and_short(ctx):
    for reads in ([], [""], ["ACG"], ["NNNN", "AC"]):
        a = gpu_final_kmers(ctx, reads, 5)
        assert a.shape == (0, 1)
    a, c = gpu_final_kmers(ctx, ["ACGTA"], 5, with_counts=True)
    exp, expc = O.kmercount(["ACGTA"], 5, 16, 1, with_counts=True)
    assert np.array_equal(a, exp) and np.array_equal(c, expc)


def test_palindromes_even_k(ctx):
    reads = ["ACGTACGTACGT", "AATT", "GGCC" * 5]
    for k in (2, 4, 6):
        a, c = gpu_final_kmers(ctx, reads, k, with_counts=True)
        exp, expc = O.kmercount(reads, k, 16, 1, with_counts=True)
        assert np.array_equal(a, exp) and np.array_equal(c, expc)


def test_synth_reads_roundtrip_and_parity(ctx):
    r = ctx.reads_synth(3000, read_len=150, genome_len=9000, sub_rate=0.005, seed_genome=42, seed_reads=43)
    reads = r.to_list()
    assert len(reads) == 3000 and all(len(x) == 150 for x in reads)
    r2 = ctx.reads_synth(3000, read_len=150, genome_len=9000, sub_rate=0.005, seed_genome=42, seed_reads=43)
    assert r2.to_list() == reads  # deterministic
    s = ctx.count(r, 21, B.BOTH_STRANDS)
    a = s.export(B.ORDER_REFERENCE_BUCKETS16)
    exp = O.kmercount(reads, 21, 16, 4)
    assert np.array_equal(a, exp)
    # coverage ~50x: far fewer distinct than instances
    assert len(a) < s.instances // 4


def test_larger_batch_properties(ctx):
    """Size-independent properties at a size the oracle would take long for."""
    r = ctx.reads_synth(400000, read_len=150, genome_len=1200000)
    k = 21
    s = ctx.count(r, k, B.BOTH_STRANDS | B.WITH_COUNTS)
    keys, cnt = s.export(B.ORDER_SORTED, with_counts=True)
    assert s.instances == 2 * 400000 * (150 - k + 1)
    assert int(cnt.sum(dtype=np.uint64)) == s.instances          # multiplicities add up
    assert np.all(keys[1:, 0] > keys[:-1, 0])                      # strictly ascending = distinct
    ref = s.export(B.ORDER_REFERENCE_BUCKETS16)
    assert np.array_equal(np.sort(ref[:, 0]), keys[:, 0])           # same multiset in both orders
    # closed under reverse complement (spot check on a sample)
    sample = keys[:: max(1, len(keys) // 2000), 0]
    kset = set(int(x) for x in keys[:, 0])
    for x in sample:
        s_ = "".join("ACGT"[(int(x) >> (2 * i)) & 3] for i in range(k))
        assert O.kmer_words(rc(s_))[0] in kset
    # bucket ids ascend along the reference order
    b = [O.bucket(row, 16) for row in ref[:: max(1, len(ref) // 5000)]]
    assert b == sorted(b)
    # a set built directly in the reference order (tagged sort for 8-byte keys, REF prefix for 16-byte keys) equals
    # the ascending set reordered by the one stable bucket pass
    for kk in (21, 55):
        direct = ctx.count(r, kk, B.BOTH_STRANDS | B.REFERENCE_ORDER).export(B.ORDER_REFERENCE_BUCKETS16)
        generic = ctx.count(r, kk, B.BOTH_STRANDS).export(B.ORDER_REFERENCE_BUCKETS16)
        assert np.array_equal(direct, generic), kk


def test_owner_partition(ctx):
    reads = synth_reads(500, read_len=100, genome_len=5000, seed=3)
    s = ctx.count(ctx.reads_from_ascii(reads), 21, B.CANONICAL)
    keys = s.export(B.ORDER_SORTED)
    for nranks in (1, 2, 8):
        part, counts = s.export_by_owner(nranks)
        assert int(counts.sum()) == len(keys)
        assert np.array_equal(np.sort(part[:, 0]), keys[:, 0])
        # owner segments are disjoint sets whose union is everything; each sorted inside
        off = 0
        for c in counts:
            seg = part[off:off + int(c), 0]
            assert np.all(seg[1:] > seg[:-1])
            off += int(c)


def test_owner_matches_host_mirror(ctx):
    """Device owner function == numpy mirror used by the host-side exchange logic."""
    from spades_for_blackbird_amd import distributed as D
    reads = synth_reads(300, read_len=100, genome_len=3000, seed=4)
    for k in (21, 55):
        s = ctx.count(ctx.reads_from_ascii(reads), k, B.CANONICAL)
        for nranks in (2, 8):
            part, counts = s.export_by_owner(nranks)
            own = D.owner_of(part, nranks)
            exp = np.repeat(np.arange(nranks), counts.astype(np.int64))
            assert np.array_equal(own, exp)


def test_many_tiny_reads(ctx):
    """Thousands of reads with one or two k-mers each: a partition tile then spans more reads than
    its LDS cursor table holds (the slow path of the fused extraction)."""
    rng = np.random.default_rng(5)
    k = 21
    reads = ["".join("ACGT"[i] for i in rng.integers(0, 4, size=k + int(rng.integers(0, 2)))) for _ in range(30000)]
    reads += ["ACGTAC"] * 50 + ["A" * 400]
    exp, expc = O.kmercount(reads, k, 16, 2, with_counts=True)
    got, gotc = gpu_final_kmers(ctx, reads, k, with_counts=True)
    assert np.array_equal(got, exp) and np.array_equal(gotc, expc)
    x = ctx.extindex(ctx.reads_from_ascii(reads), k)
    ox = O.ExtIndex(reads, k, 1)
    order = np.lexsort([ox.kmers[:, 0]])
    gk, gm = x.export()
    assert np.array_equal(gk, ox.kmers[order]) and np.array_equal(gm, ox.masks[order])


def test_high_multiplicity_overflow(ctx):
    """One k-mer repeated far more often than a bucket holds (poly-A reads) next to normal reads:
    exercises the oversize-bucket paths of the MSD sort."""
    reads = synth_reads(3000, read_len=150, genome_len=20000, seed=12) + ["A" * 150] * 400 + ["ACGT" * 37] * 300
    for k in (21, 33):
        exp, expc = O.kmercount(reads, k, 16, 2, with_counts=True)
        got, gotc = gpu_final_kmers(ctx, reads, k, with_counts=True)
        assert np.array_equal(got, exp) and np.array_equal(gotc, expc)


def test_unsorted_sets(ctx):
    """BBK_UNSORTED: dedup only (hash-bucket order) -- same set, may be owner-partitioned, expanded and
    re-merged, refuses a key-ordered export."""
    reads = synth_reads(1500, read_len=120, genome_len=8000, seed=9)
    for k in (21, 33):
        r = ctx.reads_from_ascii(reads)
        ref = ctx.count(r, k, B.CANONICAL | B.WITH_COUNTS)
        rk, rc_ = ref.export(B.ORDER_SORTED, with_counts=True)
        u = ctx.count(r, k, B.CANONICAL | B.UNSORTED | B.WITH_COUNTS)
        assert len(u) == len(ref)
        if not os.environ.get("BBK_DISABLE_MSD"):  # the general (LSD) path always yields a sorted set
            with pytest.raises(B.BBKError):
                u.export(B.ORDER_SORTED)
        part, counts = u.export_by_owner(4)
        order = np.lexsort([part[:, j] for j in range(part.shape[1] - 1, -1, -1)])
        assert np.array_equal(part[order], rk)
        both = u.both_strands()
        exp, expc = O.kmercount(reads, k, 16, 2, with_counts=True)
        got, gotc = both.export(B.ORDER_REFERENCE_BUCKETS16, with_counts=True)
        assert np.array_equal(got, exp) and np.array_equal(gotc, expc)


def test_wide_keys_shared_first_word(ctx):
    """16-byte keys whose first 32 bases coincide: the bucket sort orders runs of equal word 0 by the
    other words; > 48 such keys force its all-words radix fallback."""
    rng = np.random.default_rng(17)
    prefix = "".join("ACGT"[i] for i in rng.integers(0, 4, size=32))
    reads = [prefix + "".join("ACGT"[i] for i in rng.integers(0, 4, size=40)) for _ in range(400)]
    reads += synth_reads(500, read_len=120, genome_len=3000, seed=18)
    for k in (33, 41, 55, 63):
        exp, expc = O.kmercount(reads, k, 16, 2, with_counts=True)
        got, gotc = gpu_final_kmers(ctx, reads, k, with_counts=True)
        assert np.array_equal(got, exp) and np.array_equal(gotc, expc), k


def test_abi_error_paths(ctx):
    """Errors surface as status codes + messages, never as crashes (no exception crosses the C ABI)."""
    import ctypes as C
    L = B.load_library()
    r = ctx.reads_from_ascii(["ACGTACGTACGTACGTACGTACGTA"])
    h = C.c_void_p()
    assert L.bbk_count(ctx._h, r._h, 0, B.BOTH_STRANDS, C.byref(h)) == -1      # k out of range
    assert b"out of range" in L.bbk_last_error()
    assert L.bbk_count(ctx._h, r._h, 128, B.BOTH_STRANDS, C.byref(h)) == -1
    assert L.bbk_count(ctx._h, r._h, 21, 0, C.byref(h)) == -1                  # neither strand mode
    assert L.bbk_count(ctx._h, r._h, 21, B.BOTH_STRANDS | B.CANONICAL, C.byref(h)) == -1
    assert L.bbk_count(None, r._h, 21, B.BOTH_STRANDS, C.byref(h)) == -1
    assert L.bbk_extindex_build(ctx._h, r._h, 127, C.byref(h)) == -1           # k+1 must be < 128
    with pytest.raises(B.BBKError):
        ctx.count(r, 21, B.BOTH_STRANDS).write_final_kmers("/nonexistent_dir/x/final_kmers")
    c2 = C.c_void_p()
    assert L.bbk_ctx_create(9999, C.byref(c2)) == -1


def test_spades_binary_read_cache(ctx, tmp_path):
    """SPAdes binary read cache (.seq/.off): written from the device arrays and read back; the layout is
    checked field by field against the reference's writer (binary_converter.cpp:50-113; no cache file ships
    with the reference, so its bytes are "parity unpinned")."""
    import struct
    reads = synth_reads(250, read_len=100, genome_len=3000, seed=41) + ["ACGT", "A" * 33, "ACGTNNNNACGTAC"]
    r = ctx.reads_from_ascii(reads)
    prefix = str(tmp_path / "lib")
    r.write_spades_binary(prefix)
    data = open(prefix + ".seq", "rb").read()
    n, max_len, total = struct.unpack_from("<QQQ", data, 0)
    kept = r.to_list()
    assert (n, max_len, total) == (len(kept), max(len(x) for x in kept), sum(len(x) for x in kept))
    pos, offs = 24, []
    for i, s in enumerate(kept):
        if i % 100 == 0:
            offs.append(pos)
        (size,) = struct.unpack_from("<Q", data, pos)
        assert size == len(s)
        nw = (size + 31) // 32
        words = struct.unpack_from("<%dQ" % nw, data, pos + 8)
        dec = "".join("ACGT"[(words[j // 32] >> (2 * (j % 32))) & 3] for j in range(size))
        assert dec == s
        pos += 8 + 8 * nw + 4
    assert pos == len(data)
    assert list(struct.unpack("<%dQ" % len(offs), open(prefix + ".off", "rb").read())) == offs
    r2 = ctx.reads_from_spades_binary(prefix + ".seq")
    assert r2.to_list() == kept
    a = ctx.count(r, 21, B.BOTH_STRANDS).export(B.ORDER_REFERENCE_BUCKETS16)
    b = ctx.count(r2, 21, B.BOTH_STRANDS).export(B.ORDER_REFERENCE_BUCKETS16)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("k", [4, 21, 30, 31, 32, 55, 77])
def test_reference_order_flag(ctx, golden, golden_dir, k):
    """BBK_REFERENCE_ORDER stores the set in the final_kmers order: exporting in that order is a copy and must
    equal the oracle (and, for the toy data, the reference's own md5); the ascending export must still be
    ascending.  k <= 30 takes the tagged-sort path, larger k the generic reorder."""
    reads = synth_reads(600, read_len=150, genome_len=5000, sub_rate=0.01, seed=100 + k, n_rate=0.001)
    reads += ["", "ACGT" * 40, "T" * 150]
    exp, expc = O.kmercount(reads, k, 16, 2, with_counts=True)
    for wc in (False, True):
        s = ctx.count(ctx.reads_from_ascii(reads), k, B.BOTH_STRANDS | B.REFERENCE_ORDER | (B.WITH_COUNTS if wc else 0))
        ptr, order = s.device_keys()
        assert order == B.ORDER_REFERENCE_BUCKETS16 and ptr
        if wc:
            got, gotc = s.export(B.ORDER_REFERENCE_BUCKETS16, with_counts=True)
            assert np.array_equal(gotc, expc)
        else:
            got = s.export(B.ORDER_REFERENCE_BUCKETS16)
        assert np.array_equal(got, exp)
        asc = s.export(B.ORDER_SORTED)
        ref = ctx.count(ctx.reads_from_ascii(reads), k, B.BOTH_STRANDS).export(B.ORDER_SORTED)
        assert np.array_equal(asc, ref)
    if k == 21:
        g = golden["toy_kmercount"]
        toy = []
        for f in g["files"]:
            toy += read_fastq_gz(os.path.join(golden_dir, f))
        s = ctx.count(ctx.reads_from_ascii(toy), 21, B.BOTH_STRANDS | B.REFERENCE_ORDER)
        assert hashlib.md5(s.export(B.ORDER_REFERENCE_BUCKETS16).tobytes()).hexdigest() == g["k21"]["md5"]
    with pytest.raises(B.BBKError):
        ctx.count(ctx.reads_from_ascii(reads), k, B.CANONICAL | B.REFERENCE_ORDER | B.UNSORTED)


def test_heavy_hitters_at_default_thresholds(ctx):
    """300 k reads (39 M k-mer instances: the histogram-free slot mode is on by default at this size) of which a
    fifth are poly-A / tandem repeats: single k-mers with millions of instances overflow their segment slot, the
    spill list and the exact pass take over; counts must equal the oracle."""
    reads = synth_reads(240_000, read_len=150, genome_len=600_000, sub_rate=0.005, seed=77)
    reads += ["A" * 150] * 30_000 + ["ACGT" * 37 + "AC"] * 20_000 + ["AAC" * 50] * 10_000
    r = ctx.reads_from_ascii(reads)
    exp, expc = O.kmercount(reads, 21, 16, 8, with_counts=True)
    s = ctx.count(r, 21, B.BOTH_STRANDS | B.WITH_COUNTS)
    got, gotc = s.export(B.ORDER_REFERENCE_BUCKETS16, with_counts=True)
    assert np.array_equal(got, exp)
    assert np.array_equal(gotc, expc)
    assert int(gotc.max()) > 3_000_000
    s2 = ctx.count(r, 21, B.BOTH_STRANDS | B.REFERENCE_ORDER)
    assert np.array_equal(s2.export(B.ORDER_REFERENCE_BUCKETS16), exp)
    x = ctx.extindex(r, 21)
    assert len(x) > 0


@pytest.mark.parametrize("k", [21, 33, 77])
def test_median_multiplicity_read_filter(ctx, k):
    """bbk_reads_median_filter (the device side of spades-read-filter: CoverageFilter::CheckMedianMlt,
    coverage_filtering_read_wrapper.hpp:22-76) against a direct evaluation: per read the upper median
    (nth_element at nk/2) of the canonical k-mer multiplicities of the whole read set, compared with the threshold.
    Odd k only in the oracle comparison (no self-reverse-complementary k-mers, whose both-strand count doubles)."""
    deep = synth_reads(1500, read_len=150, genome_len=4000, sub_rate=0.01, seed=5, n_rate=0.002)     # ~55x
    shallow = synth_reads(300, read_len=150, genome_len=30000, sub_rate=0.01, seed=6, n_rate=0.002)  # ~1.5x
    reads = deep + shallow + ["", "ACGT", "ACGTTGCA" * 12, "N" * 80, "T" * 150]
    r = ctx.reads_from_ascii(reads)
    cset = ctx.count(r, k, B.CANONICAL | B.WITH_COUNTS)
    both, cnt = O.kmercount(reads, k, 16, 2, with_counts=True)
    mult = {tuple(row): int(c) for row, c in zip(both.tolist(), cnt.tolist())}
    for thr in (1, 3, 8):
        got = ctx.median_filter(r, cset, thr)
        exp = np.zeros(len(reads), dtype=np.uint8)
        for i, s in enumerate(reads):
            a, b = O.longest_valid(s)
            seg = s[a:b].upper()
            nk = len(seg) - k + 1
            if nk <= 0:
                continue  # median 0 < thr
            m = sorted(mult[tuple(O.kmer_words(seg[p:p + k]))] for p in range(nk))
            exp[i] = 1 if m[nk // 2] >= thr else 0
        assert np.array_equal(got, exp), (k, thr)
        assert 0 < int(got.sum()) < len(reads)
    with pytest.raises(B.BBKError):
        ctx.median_filter(r, ctx.count(r, k, B.BOTH_STRANDS | B.WITH_COUNTS), 2)  # needs the canonical set


def test_full_size_properties():
    """BASELINE.json configs[1] at full size (10 M x 150 bp, k=21: 2.6 G k-mer instances), checked through
    size-independent properties on the device: strictly ascending = distinct, multiplicities add up to the instance
    count, the set is closed under reverse complement (sample), the reference-order set is the same multiset with
    ascending XXH3 buckets (sample), and the sharded building blocks (unsorted canonical set) agree in size."""
    import torch
    ctx = B.Context(0, stream=torch.cuda.current_stream())
    k, n_reads, L = 21, 10_000_000, 150
    r = ctx.reads_synth(n_reads, read_len=L, genome_len=n_reads * L // 50, seed_genome=42, seed_reads=43)
    s = ctx.count(r, k, B.BOTH_STRANDS | B.WITH_COUNTS)
    n = len(s)
    assert s.instances == 2 * n_reads * (L - k + 1)
    keys = torch.empty((n, 1), dtype=torch.int64, device="cuda")
    cnt = torch.empty(n, dtype=torch.int32, device="cuda")
    s.export_to(keys, B.ORDER_SORTED, cnt)
    kv = keys.view(-1)
    assert bool(torch.all(kv[1:] > kv[:-1]))                       # 42-bit keys: int64 order == uint64 order
    assert int(cnt.sum(dtype=torch.int64).item()) == s.instances   # every instance counted once
    assert int(cnt.min().item()) >= 1
    ref = ctx.count(r, k, B.BOTH_STRANDS | B.REFERENCE_ORDER)
    assert len(ref) == n
    rk = torch.empty((n, 1), dtype=torch.int64, device="cuda")
    ref.export_to(rk, B.ORDER_REFERENCE_BUCKETS16)
    assert bool(torch.equal(torch.sort(rk.view(-1)).values, kv))    # same multiset
    step = max(1, n // 4000)
    sample = rk.view(-1)[::step].cpu().numpy().view(np.uint64)
    b = [O.bucket(np.array([x], dtype=np.uint64), 16) for x in sample]
    assert b == sorted(b) and b[0] == 0 and b[-1] == 15
    # closure under reverse complement on a sample (binary search on the device-sorted keys)
    samp = kv[:: max(1, n // 2000)].cpu().numpy().view(np.uint64)
    rcs = []
    for x in samp:
        s_ = "".join("ACGT"[(int(x) >> (2 * i)) & 3] for i in range(k))
        rcs.append(O.kmer_words(rc(s_))[0])
    q = torch.from_numpy(np.array(rcs, dtype=np.uint64).view(np.int64)).cuda()
    pos = torch.searchsorted(kv, q)
    assert bool(torch.all(kv[pos.clamp(max=n - 1)] == q))
    u = ctx.count(r, k, B.CANONICAL | B.UNSORTED)
    assert 2 * len(u) == n                                           # odd k: no self-reverse-complementary k-mers
    ctx.close()


def test_direct_sorted_output_withdrawn(ctx, monkeypatch, capfd):
    """Odd k: the both-strand set of a distinct canonical set holds no duplicates, so the sorting kernels write the
    dense result directly (no compaction pass).  A thousand k-mers that differ only in their first five bases crowd
    one bin of the in-bucket distribution sort: that bucket is left to the second-chance kernel, the direct output
    is withdrawn and the pass redone in place -- the result must not change."""
    import random
    rnd = random.Random(11)
    suffix = "".join(rnd.choice("ACGT") for _ in range(16))
    reads = synth_reads(1200, read_len=100, genome_len=6000, sub_rate=0.01, seed=12)
    reads += ["".join(rnd.choice("ACGT") for _ in range(30)) + suffix for _ in range(4000)]
    monkeypatch.setenv("BBK_VERBOSE", "1")
    for flags in (B.BOTH_STRANDS, B.BOTH_STRANDS | B.REFERENCE_ORDER | B.WITH_COUNTS):
        exp, expc = O.kmercount(reads, 21, 16, 2, with_counts=True)
        s = ctx.count(ctx.reads_from_ascii(reads), 21, flags)
        if flags & B.WITH_COUNTS:
            got, gotc = s.export(B.ORDER_REFERENCE_BUCKETS16, with_counts=True)
            assert np.array_equal(gotc, expc)
        else:
            got = s.export(B.ORDER_REFERENCE_BUCKETS16)
        assert np.array_equal(got, exp)
    err = capfd.readouterr().err
    if not any(os.environ.get(v) for v in ("BBK_DISABLE_MSD", "BBK_NO_DIRECT", "BBK_NO_DIST")):
        assert "direct output withdrawn" in err


def test_device_info_and_xcd_switches(monkeypatch):
    """bbk_ctx_device_info: an MI355X in SPX mode shows 256 CUs and 8 XCDs (what turns the XCD-local fill fronts on); the
    result of a count does not depend on those switches (placement only)."""
    import numpy as np
    import spades_for_blackbird_amd as B
    ctx = B.Context(0)
    info = ctx.device_info()
    assert info["num_cus"] > 0 and 1 <= info["num_xcds"] <= 16, info
    import torch
    if "MI355" in torch.cuda.get_device_name(0):
        assert info == {"num_cus": 256, "num_xcds": 8}, info
    r = ctx.reads_synth(300_000, read_len=150, genome_len=900_000, seed_reads=5)
    monkeypatch.setenv("BBK_SLOTS_MIN", "0")
    a = ctx.count(r, 21, B.BOTH_STRANDS | B.WITH_COUNTS).export(B.ORDER_REFERENCE_BUCKETS16, with_counts=True)
    b55 = ctx.count(r, 55, B.BOTH_STRANDS).export(B.ORDER_REFERENCE_BUCKETS16)
    assert len(a[0]) > 1_000_000
    ctx.close()
    # the switches are read once per process: the other setting runs in a child
    import subprocess, sys, os, hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib; sys.path.insert(0, %r); import spades_for_blackbird_amd as B; ctx = B.Context(0); "
            "r = ctx.reads_synth(300000, read_len=150, genome_len=900000, seed_reads=5); "
            "a = ctx.count(r, 21, B.BOTH_STRANDS | B.WITH_COUNTS).export(B.ORDER_REFERENCE_BUCKETS16, with_counts=True); "
            "b = ctx.count(r, 55, B.BOTH_STRANDS).export(B.ORDER_REFERENCE_BUCKETS16); "
            "print(hashlib.md5(a[0].tobytes()).hexdigest(), hashlib.md5(a[1].tobytes()).hexdigest(), hashlib.md5(b.tobytes()).hexdigest())" % root)
    env = dict(os.environ, BBK_XCD_SLOTS="0", BBK_XCD_TILES="0", BBK_SLOTS_MIN="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    want = "%s %s %s" % (hashlib.md5(a[0].tobytes()).hexdigest(), hashlib.md5(a[1].tobytes()).hexdigest(),
                         hashlib.md5(b55.tobytes()).hexdigest())
    assert out.stdout.strip().splitlines()[-1] == want
