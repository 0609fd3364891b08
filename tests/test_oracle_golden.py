"""The CPU oracle against every golden vector the reference holds for this path
(tests/golden/reference_kats.json; sources cited there).  CPU-only."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from oracle import oracle as O
from spades_for_blackbird_amd.tools import gfa_canon
from tests.helpers import rc, read_fastq_gz


def test_rtseq_selector(golden):
    for seq, i, exp in golden["rtseq_selector"]["cases"]:
        km = O.kmer_from_ascii(seq)
        assert "ACGT"[O.lib().orc_kmer_get(C.byref(km), i)] == exp


def test_rtseq_shift_left(golden):
    for seq, c, exp in golden["rtseq_shift_left"]["cases"]:
        assert O.kmer_shl(seq, c) == exp


def test_rtseq_reverse_complement(golden):
    for seq, exp in golden["rtseq_reverse_complement"]["cases"]:
        assert O.kmer_rc(seq) == exp
        assert exp == rc(seq)


def test_is_minimal_matches_definition():
    rng = np.random.default_rng(7)
    for k in (1, 2, 5, 21, 22, 31, 32, 33, 55, 64, 65, 127):
        for _ in range(50):
            s = "".join("ACGT"[i] for i in rng.integers(0, 4, size=k))
            assert O.kmer_is_minimal(s) == (s <= rc(s))
    assert O.kmer_is_minimal("ACGT")  # palindrome -> minimal


def test_final_kmers_encoding(golden):
    g = golden["final_kmers_encoding"]
    w = np.array(O.kmer_words(g["kmer"]), dtype=np.uint64)
    assert list(w.tobytes()) == g["bytes"]


def test_xxh3_against_python_xxhash_and_vendored_header():
    import xxhash
    rng = np.random.default_rng(3)
    ref = None
    so = os.path.join(os.path.dirname(O.__file__), "_ref", "libxxh3_ref.so")
    if os.path.exists(so):
        ref = C.CDLL(so)
        ref.ref_xxh3_64_with_seed.restype = C.c_uint64
        ref.ref_xxh3_64_with_seed.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
    for nw in (1, 2, 3, 4):
        for _ in range(300):
            w = rng.integers(0, 2 ** 64, size=nw, dtype=np.uint64)
            h = O.xxh3_64(w)
            assert h == xxhash.xxh3_64_intdigest(w.tobytes())
            if ref is not None:
                assert h == ref.ref_xxh3_64_with_seed(w.ctypes.data, nw * 8, 0)
            assert O.bucket(w, 16) == (h * 16) >> 64


def test_longest_valid(golden):
    g = golden["longest_valid"]
    f, t = O.longest_valid(g["reads"][0])
    assert g["reads"][0][f:t].upper() == g["kept_run"]
    assert O.longest_valid("NNNN") == (0, 0)
    assert O.longest_valid("ACNGT") == (0, 2)  # first wins on ties
    got = O.kmercount(g["reads"], g["k"])
    assert len(got) == g["n_kmers"]
    run = g["kept_run"]
    exp = set()
    for s in (run, rc(run)):
        for i in range(len(s) - g["k"] + 1):
            exp.add(O.kmer_words(s[i:i + g["k"]])[0])
    assert set(int(x) for x in got[:, 0]) == exp


def test_toy_kmercount_md5(golden, golden_dir):
    g = golden["toy_kmercount"]
    reads = []
    for f in g["files"]:
        reads += read_fastq_gz(os.path.join(golden_dir, f))
    for k, key in ((21, "k21"), (55, "k55")):
        for T in (1, 3):
            a = O.kmercount(reads, k, 16, T)
            assert len(a) == g[key]["n_kmers"]
            assert a.nbytes == g[key]["bytes"]
            assert hashlib.md5(a.tobytes()).hexdigest() == g[key]["md5"]
    a = O.kmercount(read_fastq_gz(os.path.join(golden_dir, g["files"][0])), 77, 16, 2)
    assert len(a) == g["k77"]["n_kmers_r1_only"]
    b = [O.bucket(r, 16) for r in a]
    assert b == sorted(b)


def test_kmercount_counts_sum():
    reads = ["ACGTTGCATGCA", "ACGTTGCAAGCA", "TTTTTTTTTTTT"]
    k = 4
    a, c = O.kmercount(reads, k, 16, 2, with_counts=True)
    assert int(c.sum()) == sum(2 * (len(r) - k + 1) for r in reads)
    assert len(np.unique(a[:, 0])) == len(a)


def test_toy_gbuilder(golden, golden_dir):
    g = golden["toy_gbuilder"]
    reads = read_fastq_gz(os.path.join(golden_dir, g["file"]))
    for T in (1, 4):
        x = O.ExtIndex(reads, 21, T)
        assert (x.n_kp1, x.n_k) == (g["k21"]["n_kp1"], g["k21"]["n_k"])
        u = x.unitigs()
        assert u.n == g["k21"]["n_unitigs"] and u.n_loops == 0
        assert sorted(len(s) for s in u.seqs) == sorted(g["k21"]["unitig_lengths"])
        txt, nv, nl = u.gfa(with_cov=True)
        assert (nv, nl) == (g["k21"]["n_vertices"], g["k21"]["n_links"])
        S = sorted(l.split("\t")[2] for l in txt.splitlines() if l.startswith("S"))
        assert hashlib.md5(("\n".join(S) + "\n").encode()).hexdigest() == g["k21"]["sorted_S_sequences_md5"]
        assert sorted(u.kc) == sorted(g["k21"]["KC"])
        dp = sorted(l.split("\t")[3][5:] for l in txt.splitlines() if l.startswith("S"))
        assert dp == sorted(g["k21"]["DP"])
    # T = 1 reproduces the recorded unitig order too
    u = O.ExtIndex(reads, 21, 1).unitigs()
    assert [len(s) for s in u.seqs] == g["k21"]["unitig_lengths"]
    assert u.kc == g["k21"]["KC"]
    x = O.ExtIndex(reads, 55, 2)
    u = x.unitigs()
    _, nv, _ = u.gfa()
    assert (x.n_kp1, x.n_k, u.n, nv) == (g["k55"]["n_kp1"], g["k55"]["n_k"], g["k55"]["n_unitigs"],
                                         g["k55"]["n_vertices"])


def test_pair_equals_r1(golden, golden_dir):
    g = golden["toy_kmercount"]
    r1 = read_fastq_gz(os.path.join(golden_dir, g["files"][0]))
    r2 = read_fastq_gz(os.path.join(golden_dir, g["files"][1]))
    a = O.ExtIndex(r1, 21, 2).unitigs().gfa()[0]
    b = O.ExtIndex(r1 + r2, 21, 2).unitigs().gfa()[0]
    assert gfa_canon.canon_md5(a) == gfa_canon.canon_md5(b)


def test_construction_kats(golden):
    g = golden["construction_unitigs_k5"]
    for c in g["cases"]:
        u = O.ExtIndex(c["reads"], g["k"], 1).unitigs()
        got = set(u.seqs) | set(rc(s) for s in u.seqs)
        exp = set(c["edges"]) | set(rc(s) for s in c["edges"])
        assert got == exp, c["name"]
        for s in u.seqs:
            assert not (s < rc(s))


def test_coverage_kat(golden):
    g = golden["construction_coverage_k3"]
    u = O.ExtIndex(g["reads"], g["k"], 1).unitigs()
    got = {}
    for s, kc in zip(u.seqs, u.kc):
        got[s] = kc
        got[rc(s)] = kc
    for e, cov in g["coverage"].items():
        assert got[e] == cov
    assert set(got) == set(g["edges"]) | set(rc(e) for e in g["edges"])


def test_loop_golden(golden):
    g = golden["loop_k5"]
    for T in (1, 2, 8):
        u = O.ExtIndex(g["reads"], g["k"], T).unitigs()
        assert (u.n - u.n_loops, u.n_loops) == (g["n_paths"], g["n_loops"])
        txt = u.gfa()[0]
        exp = "S\t3\t%s\tDP:f:0\tKC:i:0\nL\t3\t-\t3\t-\t5M\n" % g["S"][0]
        assert gfa_canon.canon_text(txt) == gfa_canon.canon_text(exp)


def test_self_rc_edge_golden(golden):
    g = golden["self_rc_edge_k5"]
    u = O.ExtIndex(g["reads"], g["k"], 1).unitigs()
    assert sorted(u.seqs) == sorted(g["S"])
    txt = u.gfa()[0]
    name = {l.split("\t")[2]: l.split("\t")[1] for l in txt.splitlines() if l.startswith("S")}
    exp = txt_S = "".join("S\t%s\t%s\n" % (name[s], s) for s in g["S"])
    for a, oa, b, ob in g["links_by_seq"]:
        exp += "L\t%s\t%s\t%s\t%s\t5M\n" % (name[a], oa, name[b], ob)
    assert gfa_canon.canon_text(txt) == gfa_canon.canon_text(exp)
    assert "L\t5\t-\t3\t+\t5M" in txt and "L\t5\t+\t3\t+\t5M" in txt


def test_split_loop_golden(golden):
    g = golden["split_loop_k5"]
    x = O.ExtIndex(g["reads"], g["k"], 1)
    assert (x.n_kp1, x.n_k) == (g["n_kp1"], g["n_k"])
    u = x.unitigs()
    assert (u.n - u.n_loops, u.n_loops) == (g["n_paths"], g["n_loops"])
    assert u.seqs == g["S"]
    txt = u.gfa()[0]
    assert "L\t5\t+\t3\t+\t5M" in txt and txt.count("\nL\t") == 1


def oriented_edge_cover(seqs, k):
    """Oriented (k+1)-mers of every segment and of the RC of every non-self-RC segment."""
    out = []
    for s in seqs:
        for t in ((s,) if s == rc(s) else (s, rc(s))):
            out += [t[i:i + k + 1] for i in range(len(t) - k)]
    return out


def test_unitig_invariants_random():
    """Invariant behind SURVEY 8(a): every oriented (k+1)-mer of E u rc(E) is spelled exactly
    once by the segments and their reverse complements (a self-RC segment counts once)."""
    from tests.helpers import synth_reads
    reads = synth_reads(300, read_len=60, genome_len=2000, sub_rate=0.01, seed=5)
    for k in (5, 9, 21, 33):
        x = O.ExtIndex(reads, k, 2)
        E = set()
        for r in x.kp1:
            km = O._Kmer()
            for i, w in enumerate(r):
                km.w[i] = int(w)
            e = O.kmer_str(km, k + 1)
            E.add(e)
            E.add(rc(e))
        u = x.unitigs()
        if u.n_loops:
            continue
        cover = oriented_edge_cover(u.seqs, k)
        assert len(cover) == len(set(cover))
        assert set(cover) == E


def test_oracle_early_tip_clipper_semantics():
    """orc_extindex_clip_tips (restated EarlyTipClipperProcessor, early_simplification.hpp:37-160) on a case worked
    by hand: one read with a substitution 5 bases before its end adds a dead-end branch of 5 k-mers next to the true
    path; clipping isolates exactly those and restores the masks of the error-free index.  (The reference ships no
    fixture for this step; this pins the restatement to the algorithm's definition.)"""
    import random
    rnd = random.Random(7)
    k, L = 21, 100
    genome = "".join(rnd.choice("ACGT") for _ in range(400))
    clean = [genome[i:i + L] for i in range(0, len(genome) - L + 1, 10)]
    bad = list(genome[100:100 + L])
    pos = L - 5
    bad[pos] = {"A": "C", "C": "G", "G": "T", "T": "A"}[bad[pos]]
    ox_clean = O.ExtIndex(clean, k, 1)
    ox = O.ExtIndex(clean + ["".join(bad)], k, 1)
    assert ox.n_k == ox_clean.n_k + 5
    removed, links = ox.clip_tips(L - k)
    assert (removed, links) == (5, 1)
    ref = {tuple(r): m for r, m in zip(ox_clean.kmers.tolist(), ox_clean.masks.tolist())}
    extra = 0
    for r, m in zip(ox.kmers.tolist(), ox.masks.tolist()):
        if tuple(r) in ref:
            assert m == ref[tuple(r)]
        else:
            assert m == 0
            extra += 1
    assert extra == 5
    # a bound shorter than the tip leaves it alone
    ox2 = O.ExtIndex(clean + ["".join(bad)], k, 1)
    assert ox2.clip_tips(3) == (0, 0)


class _Seq(C.Structure):
    _fields_ = [("data", C.c_char_p), ("from_", C.c_size_t), ("size", C.c_size_t), ("rtl", C.c_int)]


def _seq_api():
    L = O.lib()
    L.orc_seq_make.restype = _Seq
    L.orc_seq_make.argtypes = [C.c_char_p]
    L.orc_seq_rc.restype = _Seq
    L.orc_seq_rc.argtypes = [C.POINTER(_Seq)]
    L.orc_seq_subseq.restype = _Seq
    L.orc_seq_subseq.argtypes = [C.POINTER(_Seq), C.c_size_t, C.c_size_t]
    L.orc_seq_at.argtypes = [C.POINTER(_Seq), C.c_size_t]
    L.orc_seq_less.argtypes = [C.POINTER(_Seq), C.POINTER(_Seq)]
    L.orc_seq_eq.argtypes = [C.POINTER(_Seq), C.POINTER(_Seq)]
    L.orc_seq_str.argtypes = [C.POINTER(_Seq), C.c_char_p]
    L.orc_seq_concat.argtypes = [C.POINTER(_Seq), C.POINTER(_Seq), C.c_char_p]
    L.orc_nucl.restype = C.c_char
    return L


def _seq_str(L, s):
    buf = C.create_string_buffer(s.size + 1)
    L.orc_seq_str(C.byref(s), buf)
    return buf.value.decode()


def test_sequence_kats(golden):
    """test/include_test/sequence_test.cpp:12-45 restated as data: Sequence's selector, operator+, str and operator!
    -- the operators `if (s < !s) continue` (debruijn_graph_constructor.hpp:279) is written in."""
    g = golden["sequence_kats"]
    L = _seq_api()
    keep = []

    def mk(t):
        b = t.encode()
        keep.append(b)
        return L.orc_seq_make(b)
    for seq, i, exp in g["selector"]:
        s = mk(seq)
        assert "ACGT"[L.orc_seq_at(C.byref(s), i)] == exp
    for seq, n in g["zero_length"]:
        assert mk(seq).size == n
    for seq, exp in g["null_value"] + g["reverse_complement"]:
        s = mk(seq)
        r = L.orc_seq_rc(C.byref(s))
        assert _seq_str(L, r) == exp == rc(seq)
        rr = L.orc_seq_rc(C.byref(r))
        assert _seq_str(L, rr) == seq
    for a, b, exp in g["sum"]:
        sa, sb = mk(a), mk(b)
        buf = C.create_string_buffer(len(a) + len(b) + 1)
        L.orc_seq_concat(C.byref(sa), C.byref(sb), buf)
        assert buf.value.decode() == exp
    for seq in g["str"]:
        assert _seq_str(L, mk(seq)) == seq


def test_sequence_views_against_strings():
    """Subseq and operator< of views (incl. reversed ones) against plain string arithmetic; and the orientation rule
    of the unitig extractor: a string is kept iff not (s < !s) <=> s >= rc(s) as strings."""
    L = _seq_api()
    rng = np.random.default_rng(11)
    for _ in range(300):
        n = int(rng.integers(1, 80))
        t = "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
        b = t.encode()
        s = L.orc_seq_make(b)
        r = L.orc_seq_rc(C.byref(s))
        a0, a1 = sorted(int(v) for v in rng.integers(0, n + 1, size=2))
        assert _seq_str(L, L.orc_seq_subseq(C.byref(s), a0, a1)) == t[a0:a1]
        assert _seq_str(L, L.orc_seq_subseq(C.byref(r), a0, a1)) == rc(t)[a0:a1]
        assert bool(L.orc_seq_less(C.byref(s), C.byref(r))) == (t < rc(t))
        assert bool(L.orc_seq_eq(C.byref(s), C.byref(r))) == (t == rc(t))
        pre = L.orc_seq_subseq(C.byref(s), 0, a1)
        assert bool(L.orc_seq_less(C.byref(pre), C.byref(s))) == (t[:a1] < t)  # shorter prefix first
    # kept orientation of every unitig the oracle emits: s >= rc(s)
    from tests.helpers import synth_reads
    u = O.ExtIndex(synth_reads(200, read_len=100, genome_len=1500, seed=3), 21, 1).unitigs()
    for sq in u.seqs:
        b = sq.encode()
        s = L.orc_seq_make(b)
        r = L.orc_seq_rc(C.byref(s))
        assert not L.orc_seq_less(C.byref(s), C.byref(r))


def test_nucl_kats(golden):
    """test/include_test/nucl_test.cpp:10-33"""
    g = golden["nucl_kats"]
    L = _seq_api()
    for code, ch in g["nucl"]:
        assert L.orc_nucl(code).decode() == ch
    for ch, code in g["dignucl"]:
        assert L.orc_dignucl(C.c_char(ch.encode())) == code
    for a, b in g["complement"]:
        assert L.orc_complement(a) == b
    for ch in g["is_nucl_true"]:
        assert L.orc_is_nucl(C.c_char(ch.encode()))
    for ch in g["is_nucl_false"]:
        assert not L.orc_is_nucl(C.c_char(ch.encode()))
