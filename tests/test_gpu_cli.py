"""GPU: the two CLI drop-ins end to end (argv -> files), against the golden md5s and the oracle."""
import hashlib
import os
import subprocess

import pytest

from oracle import oracle as O
from spades_for_blackbird_amd import build, build_host
from spades_for_blackbird_amd.tools import gfa_canon
from tests.helpers import read_fastq_gz

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bins():
    build.build()
    return {os.path.basename(p): p for p in build_host.build()}


def test_kmercount_cli_toy(bins, golden, golden_dir, tmp_path):
    g = golden["toy_kmercount"]
    files = [os.path.join(golden_dir, f) for f in g["files"]]
    for k, key in ((21, "k21"), (55, "k55")):
        wd = tmp_path / ("w%d" % k)
        r = subprocess.run([bins["spades-kmercount"], "-k", str(k), "-t", "4", "-w", str(wd)] + files,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "There are %d kmers in total" % g[key]["n_kmers"] in r.stdout
        data = open(wd / "final_kmers", "rb").read()
        assert len(data) == g[key]["bytes"]
        assert hashlib.md5(data).hexdigest() == g[key]["md5"]
    # YAML dataset gives the same file
    y = tmp_path / "toy.yaml"
    y.write_text("- left reads: [%s]\n  orientation: fr\n  right reads: [%s]\n  type: paired-end\n" % tuple(files))
    wd = tmp_path / "wy"
    r = subprocess.run([bins["spades-kmercount"], "-k", "21", "-d", str(y), "-w", str(wd)], capture_output=True,
                       text=True)
    assert r.returncode == 0, r.stderr
    assert hashlib.md5(open(wd / "final_kmers", "rb").read()).hexdigest() == g["k21"]["md5"]


def test_gbuilder_cli_toy(bins, golden, golden_dir, tmp_path):
    g = golden["toy_gbuilder"]
    f = os.path.join(golden_dir, g["file"])
    out = tmp_path / "g.gfa"
    r = subprocess.run([bins["spades-gbuilder"], f, str(out), "-k", "21", "--gfa"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "%d sequences extracted" % g["k21"]["n_unitigs"] in r.stdout
    txt = open(out).read()
    S = sorted(l.split("\t")[2] for l in txt.splitlines() if l.startswith("S"))
    assert hashlib.md5(("\n".join(S) + "\n").encode()).hexdigest() == g["k21"]["sorted_S_sequences_md5"]
    exp = O.ExtIndex(read_fastq_gz(f), 21, 1).unitigs().gfa()[0]
    assert gfa_canon.canon_md5(txt) == gfa_canon.canon_md5(exp)
    assert txt.splitlines()[0].startswith("S\t3\t") and txt.splitlines()[0].endswith("\tDP:f:0\tKC:i:0")
    out2 = tmp_path / "u.fa"
    r = subprocess.run([bins["spades-gbuilder"], f, str(out2), "-k", "21"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    fa = open(out2).read()
    assert fa.startswith(">EDGE_1_length_")
    assert fa.count(">") == g["k21"]["n_unitigs"]


def test_kmer_estimating_cli_toy(bins, golden, golden_dir, tmp_path):
    """spades-kmer-estimating drop-in: the reference prints a HyperLogLog ESTIMATE of the number of distinct k-mers
    (strands identified); the engine prints the exact number, checked here against the oracle's canonical set (the
    both-strand set of the toy data minus its self-reverse-complementary k-mers, halved)."""
    import numpy as np
    g = golden["toy_kmercount"]
    files = [os.path.join(golden_dir, f) for f in g["files"]]
    y = tmp_path / "toy.yaml"
    y.write_text("- left reads: [%s]\n  orientation: fr\n  right reads: [%s]\n  type: paired-end\n" % tuple(files))
    reads = []
    for f in files:
        reads += read_fastq_gz(f)
    for k in (21, 22, 55):
        both = O.kmercount(reads, k, 16, 2)
        canon = set()  # one representative per {k-mer, reverse complement} class
        for row in both.tolist():
            s = "".join("ACGT"[(row[i // 32] >> (2 * (i % 32))) & 3] for i in range(k))  # RtSeq layout
            canon.add(min(s, rcs(s)))
        r = subprocess.run([bins["spades-kmer-estimating"], "-k", str(k), "-d", str(y), "-t", "4"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "Kmer number estimation: %d" % len(canon) in r.stdout
    r = subprocess.run([bins["spades-kmer-estimating"], "-k", "21"], capture_output=True, text=True)
    assert r.returncode == 1 and "SYNOPSIS" in r.stdout  # -d is required


def rcs(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def test_cli_streaming_blocks_and_modes(bins, golden, golden_dir, tmp_path):
    """The CLIs stream their input block by block (-b = bytes of input text per block, -t = parser threads): tiny blocks
    (dozens of pushes + merges) must give the same final_kmers and the same GFA as one block; -c and --spades go through
    the streaming path too (coverage from a (k+1)-mer counter fed by the same blocks)."""
    g = golden["toy_kmercount"]
    files = [os.path.join(golden_dir, f) for f in g["files"]]
    for b in ("20000", "536870912"):
        wd = tmp_path / ("w" + b)
        r = subprocess.run([bins["spades-kmercount"], "-k", "21", "-t", "3", "-b", b, "-w", str(wd)] + files,
                           capture_output=True, text=True, env=dict(os.environ, BBK_MERGE_MIN="0", BBK_PHASES="1"))
        assert r.returncode == 0, r.stderr
        assert hashlib.md5(open(wd / "final_kmers", "rb").read()).hexdigest() == g["k21"]["md5"]
        ph = [l for l in r.stdout.splitlines() if l.startswith("BBK_PHASES ")]
        import json
        nblocks = json.loads(ph[0][len("BBK_PHASES "):])["blocks"]
        assert (nblocks == len(files)) if b != "20000" else (nblocks > 2 * len(files))  # one block per file, or many
    gg = golden["toy_gbuilder"]
    f = os.path.join(golden_dir, gg["file"])
    texts = []
    for b in ("15000", "536870912"):
        out = tmp_path / ("c%s.gfa" % b)
        r = subprocess.run([bins["spades-gbuilder"], f, str(out), "-k", "21", "--gfa", "-c", "-b", b, "-t", "2"],
                           capture_output=True, text=True, env=dict(os.environ, BBK_MERGE_MIN="0"))
        assert r.returncode == 0, r.stderr
        texts.append(open(out).read())
    assert gfa_canon.canon_text(texts[0], 21, with_kc=True) == gfa_canon.canon_text(texts[1], 21, with_kc=True)
    kcs = sorted(int(t[5:]) for l in texts[0].splitlines() if l.startswith("S") for t in l.split("\t") if t.startswith("KC:i:"))
    assert kcs == sorted(gg["k21"]["KC"])
    base = tmp_path / "bin_graph"
    r = subprocess.run([bins["spades-gbuilder"], f, str(base), "-k", "21", "--spades", "-c"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(str(base) + ".grseq") > 24 and os.path.getsize(str(base) + ".cvr") == 12 * gg["k21"]["n_unitigs"] + 8


def test_read_filter_cli(bins, golden, golden_dir, tmp_path):
    """spades-read-filter drop-in (projects/kmercount/read_filter.cpp): reads whose median k-mer multiplicity is <= -c
    are dropped, a pair stays when either mate passes; names and qualities pass through.  The reference's
    multiplicities are approximate (counting quotient filter), the engine's exact: the expectation is computed from the
    oracle's exact canonical counts with the reference's own median rule (CountMedianMlt,
    io/reads/coverage_filtering_read_wrapper.hpp:36-50: the element at index size/2 of the sorted multiplicities)."""
    import gzip
    import re
    import numpy as np
    g = golden["toy_kmercount"]
    f1, f2 = [os.path.join(golden_dir, f) for f in g["files"]]
    k, thr = 21, 20

    def records(path):
        out = []
        with gzip.open(path, "rt") as f:
            while True:
                h = f.readline()
                if not h:
                    break
                s = f.readline().rstrip("\n")
                f.readline()
                q = f.readline().rstrip("\n")
                out.append((h.rstrip("\n")[1:], s, q))
        return out
    r1, r2 = records(f1), records(f2)
    # a third, single-end library: a few reads of r1 plus junk that never repeats
    single = tmp_path / "single.fq"
    rng = np.random.default_rng(3)
    with open(single, "w") as f:
        for n, s, q in r1[:50]:
            f.write("@%s\n%s\n+\n%s\n" % (n, s, q))
        for j in range(20):
            s = "".join("ACGT"[i] for i in rng.integers(0, 4, size=100))
            f.write("@junk%d\n%s\n+\n%s\n" % (j, s, "I" * 100))
    y = tmp_path / "ds.yaml"
    y.write_text("- left reads: [%s]\n  orientation: fr\n  right reads: [%s]\n  type: paired-end\n"
                 "- single reads: [%s]\n  type: single\n" % (f1, f2, single))
    out = tmp_path / "out"
    r = subprocess.run([bins["spades-read-filter"], "-k", str(k), "-c", str(thr), "-d", str(y), "-o", str(out), "-t", "4"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # expectation from exact canonical multiplicities
    sing = [l for l in open(single).read().split("\n")]
    srecs = [(sing[i][1:], sing[i + 1], sing[i + 3]) for i in range(0, len(sing) - 1, 4)]
    allseq = [s for _, s, _ in r1 + r2 + srecs]
    both, cnt = O.kmercount(allseq, k, 16, 2, with_counts=True)
    mult = {}
    for row, c in zip(both.tolist(), cnt.tolist()):
        s = "".join("ACGT"[(row[i // 32] >> (2 * (i % 32))) & 3] for i in range(k))
        mult[min(s, rcs(s))] = c if s != rcs(s) else c // 2   # both-strand counts: a k-mer and its RC share the count
    def passes(seq):
        runs = re.findall("[ACGT]+", seq.upper())
        run = max(runs, key=len) if runs else ""
        if len(run) < k:
            return 0 >= thr + 1
        ms = sorted(mult[min(run[i:i + k], rcs(run[i:i + k]))] for i in range(len(run) - k + 1))
        return ms[len(ms) // 2] >= thr + 1
    keep_pairs = [i for i in range(len(r1)) if passes(r1[i][1]) or passes(r2[i][1])]
    keep_single = [i for i in range(len(srecs)) if passes(srecs[i][1])]
    assert 0 < len(keep_pairs) <= len(r1) and 0 < len(keep_single) < len(srecs)   # the threshold really splits the data
    def fq(recs, idx):
        return "".join("@%s\n%s\n+\n%s\n" % recs[i] for i in idx)
    assert open(out / "1.1.fastq").read() == fq(r1, keep_pairs)
    assert open(out / "1.2.fastq").read() == fq(r2, keep_pairs)
    assert open(out / "2.s.fastq").read() == fq(srecs, keep_single)
    assert "Total %d reads processed, %d reads left after filtering" % (len(r1), len(keep_pairs)) in r.stdout
    ds = open(out / "dataset.yaml").read()
    assert "1.1.fastq" in ds and "1.2.fastq" in ds and "2.s.fastq" in ds and "paired-end" in ds
    # the written dataset is readable by the tools again; --drop-quality writes FASTA
    out2 = tmp_path / "out2"
    r = subprocess.run([bins["spades-read-filter"], "-k", str(k), "-c", str(thr), "-d", str(out / "dataset.yaml"), "-o",
                        str(out2), "--drop-quality"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    fa = open(out2 / "1.1.fasta").read()
    assert fa.startswith(">" + r1[keep_pairs[0]][0] + "\n") and "+" not in fa.split("\n")[2]
    r = subprocess.run([bins["spades-read-filter"], "-k", "21"], capture_output=True, text=True)
    assert r.returncode == 1 and "SYNOPSIS" in r.stdout  # -d is required


# ---- several ranks in one process (--devices): SURVEY 8e / north_star "host code kept in C++ ... single RCCL all-to-all" ----
def _fasta(path, reads):
    with open(path, "w") as f:
        for i, s in enumerate(reads):
            f.write(">r%d\n%s\n" % (i, s))


@pytest.mark.parametrize("k", [21, 33, 77])
def test_kmercount_cli_devices(bins, tmp_path, k):
    """One process, one host thread + context per rank, owner-hash shards, one exchange, N-way bucket merge into ONE
    final_kmers: byte-identical to the single-device tool and to the oracle.  A one-GPU box cannot hold two RCCL ranks
    (RCCL refuses a device listed twice), so the N-rank logic runs with --exchange copy on devices 0,0[,0] (the same
    segments, rounds and merge; peer copies instead of ncclSend/ncclRecv) and the RCCL code path with one rank."""
    from tests.helpers import synth_reads
    import numpy as np
    reads = synth_reads(6000, read_len=150, genome_len=30000, sub_rate=0.01, seed=k, n_rate=0.002)
    fa = tmp_path / "r.fa"
    _fasta(fa, reads)
    exp = O.kmercount(reads, k, 16, 2)
    outs = {}
    for name, extra, env in (("single", [], {}),
                             ("rccl1", ["--devices", "0"], {}),
                             ("copy2", ["--devices", "0,0", "--exchange", "copy"], {}),
                             ("copy3_small_messages", ["--devices", "0,0,0", "--exchange", "copy"], {"BBK_GROUP_MAX_MSG": "4096"})):
        wd = tmp_path / name
        e = dict(os.environ)
        e.update(env)
        # -b small: several blocks, so that every rank gets some
        r = subprocess.run([bins["spades-kmercount"], "-k", str(k), "-t", "4", "-b", "200000", "-w", str(wd), str(fa)] + extra,
                           capture_output=True, text=True, env=e)
        assert r.returncode == 0, (name, r.stdout[-2000:], r.stderr[-2000:])
        assert "There are %d kmers in total" % len(exp) in r.stdout, (name, r.stdout[-1000:])
        outs[name] = open(wd / "final_kmers", "rb").read()
    want = np.ascontiguousarray(exp).tobytes()
    for name, data in outs.items():
        assert data == want, name


def test_gbuilder_cli_devices(bins, tmp_path):
    from tests.helpers import synth_reads
    k = 21
    reads = synth_reads(4000, read_len=150, genome_len=12000, sub_rate=0.005, seed=9)
    fa = tmp_path / "r.fa"
    _fasta(fa, reads)
    exp = O.ExtIndex(reads, k, 1).unitigs().gfa(with_cov=True)[0]
    for name, extra in (("rccl1", ["--devices", "0"]), ("copy2", ["--devices", "0,0", "--exchange", "copy"]),
                        ("copy3", ["--devices", "0,0,0", "--exchange", "copy"])):
        out = tmp_path / (name + ".gfa")
        r = subprocess.run([bins["spades-gbuilder"], str(fa), str(out), "-k", str(k), "--gfa", "-c", "-b", "150000"] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, (name, r.stdout[-2000:], r.stderr[-2000:])
        assert gfa_canon.canon_md5(open(out).read(), k, with_kc=True) == gfa_canon.canon_md5(exp, k, with_kc=True), name
