"""CPU-only: the C-ABI library loads and exports every symbol include/bbk.h declares, the host
programs honour the reference's argv contracts, and the host FASTA/FASTQ reader follows kseq."""
import gzip
import os
import re
import subprocess

import pytest

import spades_for_blackbird_amd as B
from spades_for_blackbird_amd import build, build_host, engine
from tests.helpers import read_fastq_gz

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bins():
    build.build()
    return {os.path.basename(p): p for p in build_host.build()}


def test_library_exports_every_declared_symbol():
    build.build()
    L = B.load_library()
    hdr = open(os.path.join(ROOT, "include", "bbk.h")).read()
    declared = set(re.findall(r"\b(bbk_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    for sym in sorted(declared):
        assert hasattr(L, sym), "libbbk.so does not export %s" % sym
    assert declared == set(engine.SYMBOLS)
    assert L.bbk_words(21) == 1 and L.bbk_words(33) == 2 and L.bbk_words(127) == 4
    assert b"gfx950" in L.bbk_version()


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly (never route through the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(B.BBKError):
        B.Context(0)
    src = ""
    for dp, _, fs in os.walk(os.path.join(ROOT, "spades_for_blackbird_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src += open(os.path.join(dp, f), errors="replace").read()
    assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src


def test_kmercount_argv_contract(bins):
    exe = bins["spades-kmercount"]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 255 and "No input files were specified" in r.stderr
    r = subprocess.run([exe, "-h"], capture_output=True, text=True)
    assert r.returncode == 0 and "--kmer" in r.stdout and "final_kmers" in r.stdout
    r = subprocess.run([exe, "--bogus", "x.fa"], capture_output=True, text=True)
    assert r.returncode == 1
    r = subprocess.run([exe, "-k", "notanumber", "x.fa"], capture_output=True, text=True)
    assert r.returncode == 1


def test_gbuilder_argv_contract(bins):
    exe = bins["spades-gbuilder"]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "--gfa" in r.stdout
    r = subprocess.run([exe, "in.fa", "out.gfa", "-k", "22", "--gfa"], capture_output=True, text=True)
    assert r.returncode == 255 and "k-mer size must be odd" in r.stderr
    r = subprocess.run([exe, "in.fa", "out.gfa", "-k", "129"], capture_output=True, text=True)
    assert r.returncode == 255 and "too high" in r.stderr
    r = subprocess.run([exe, "in.fa", "out.gfa", "--gfa", "--fastg"], capture_output=True, text=True)
    assert r.returncode == 1
    r = subprocess.run([exe, "/nonexistent/in.fa", "out.gfa", "--gfa"], capture_output=True, text=True)
    assert r.returncode == 255 and "does not exist" in r.stderr


def _dump(bins, path):
    r = subprocess.run([bins["bbk-fastx-dump"], path], capture_output=True, text=True)
    assert r.returncode == 0
    return r.stdout.split("\n")[:-1]


def test_fastx_reader(bins, golden_dir, tmp_path):
    p = os.path.join(golden_dir, "ecoli_1K_1.fq.gz")
    assert _dump(bins, p) == read_fastq_gz(p)
    fa = tmp_path / "a.fa"
    fa.write_text(">r1 some comment\nACGT\nacgtnn\n\nGG\n>r2\nTTTT\n>empty\n>r3\nAC\n")
    assert _dump(bins, str(fa)) == ["ACGTACGTNNGG", "TTTT", "", "AC"]
    fq = tmp_path / "b.fq"
    fq.write_text("@a\nACGT\n+\nIIII\n@b\nGGCC\nTT\n+b\nIIII\nII\n@c\nACGTAC\n+\nIII\n@d\nAAAA\n+\nIIII\n")
    # record c has a truncated quality string: the reference's parser stops there (kseq -2 -> eof)
    assert _dump(bins, str(fq)) == ["ACGT", "GGCCTT"]
    gz = tmp_path / "c.fa.gz"
    with gzip.open(gz, "wt") as f:
        f.write(">x\nACGTNACGT\n")
    assert _dump(bins, str(gz)) == ["ACGTNACGT"]


def test_dataset_yaml(bins, golden_dir, tmp_path):
    """YAML forms written by spades.py / the reference's configs (assembler/configs/debruijn/toy.yaml)."""
    y = tmp_path / "toy.yaml"
    y.write_text("- left reads: [%s/ecoli_1K_1.fq.gz]\n  orientation: fr\n  right reads: [%s/ecoli_1K_2.fq.gz]\n"
                 "  type: paired-end\n" % (golden_dir, golden_dir))
    exe = bins["spades-kmercount"]
    r = subprocess.run([exe, "-d", str(y), "-w", str(tmp_path)], capture_output=True, text=True)
    # parsing succeeded iff we get as far as opening the device (no GPU here) or finishing (GPU box)
    assert "ecoli_1K_1.fq.gz" in r.stdout or "bbk_ctx_create" in r.stderr


def test_no_kernel_uses_scratch():
    """Build hygiene (scratch computes correctly on this pool; it costs occupancy): "no scratch, no VGPR
    spills" is enforced on the code-object metadata of every kernel of the built library."""
    from spades_for_blackbird_amd import build as b
    b.build()
    res = b.check_resources()
    if res is None:
        pytest.skip("llvm-objdump / llvm-readelf not installed")
    assert len(res) > 150  # all instantiations of all eight translation units are seen
    assert all(r[2] == 0 and r[4] == 0 for r in res)


def test_final_kmers_merge_of_shards_fuzz(tmp_path):
    """host/multi.hpp: write_final_kmers_merged (the writer of `spades-kmercount --devices`) against a plain sort, for
    1..8 shards, 1..4-word records, empty shards / buckets; host-only, AddressSanitizer + UBSan."""
    from spades_for_blackbird_amd import build as b
    b.build()
    exe = str(tmp_path / "merge_fuzz")
    lib = os.path.join(ROOT, "spades_for_blackbird_amd")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fopenmp", "-fsanitize=address,undefined", "-o", exe,
                           os.path.join(ROOT, "tests", "merge_fuzz.cpp"), "-L" + lib, "-lbbk", "-lz", "-Wl,-rpath," + lib])
    for seed in (1, 2, 3):
        r = subprocess.run([exe, str(seed), str(tmp_path / "merged.bin")], capture_output=True, text=True,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert r.returncode == 0 and "MERGE-FUZZ-OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
