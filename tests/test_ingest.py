"""CPU: the parallel block parser of the CLIs (host/ingest.hpp) against the serial reader (host/fastx.hpp, whose
kseq corner cases are pinned in test_host_and_abi.py) and against hand-worked kseq results: FASTA / FASTQ, single and
multi-line, CRLF, empty records, '@' inside quality lines, truncated quality, gz, for several thread counts and block
sizes down to a few bytes (every record then crosses a block boundary and is carried over)."""
import gzip
import os
import random
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dump():
    from spades_for_blackbird_amd import build_host
    exe = [e for e in build_host.build() if e.endswith("bbk-fastx-dump")][0]

    def run(args):
        r = subprocess.run([exe] + args, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return r.stdout.split("\n")[:-1], r.stderr
    return run


def lv(s):
    runs = re.findall("[ACGTacgt]+", s)
    best = ""
    for r in runs:
        if len(r) > len(best):
            best = r
    return best.upper()


def check(dump, path, expect=None):
    from oracle import oracle as O
    serial, _ = dump(["--serial-lv", path])
    if expect is not None:
        assert serial == expect, "serial reader"
    # the oracle's reader (a restatement of kseq_read itself) must agree with the host readers
    assert [lv(x) for x in O.read_fastx_list(path)] == serial, "oracle reader"
    for threads, block in ((1, 1 << 20), (4, 1 << 20), (3, 7), (8, 64), (2, 1), (5, 300)):
        fast, _ = dump(["--fast", str(threads), str(block), path])
        assert fast == serial, (threads, block)
    return serial


def test_kseq_corner_cases(dump, tmp_path):
    fa = tmp_path / "a.fa"
    fa.write_text(">r1 some comment\nACGT\nacgtnn\n\nGG\n>r2\nTTTT\n>empty\n>r3\nAC\n")
    check(dump, str(fa), ["ACGTACGT", "TTTT", "", "AC"])
    fq = tmp_path / "b.fq"
    fq.write_text("@a\nACGT\n+\nIIII\n@b\nGGCC\nTT\n+b\nIIII\nII\n@c\nACGTAC\n+\nIII\n@d\nAAAA\n+\nIIII\n")
    check(dump, str(fq), ["ACGT", "GGCCTT"])  # record c: truncated quality -> the stream ends
    # quality lines that start with '@' and '>' ; a name that contains '@'
    fq2 = tmp_path / "c.fq"
    fq2.write_text("@r@1\nACGTA\n+\n@IIII\n@r2\nCCCCC\n+\n>>>>>\n@r3\nGGGNGGGG\n+\n@@@@@@@@\n")
    check(dump, str(fq2), ["ACGTA", "CCCCC", "GGGG"])
    # CRLF: one trailing CR per line is dropped (multi-line records stay in one run)
    crlf = tmp_path / "d.fa"
    crlf.write_bytes(b">x\r\nACGT\r\nGGCC\r\n>y\r\nTT\r\n")
    check(dump, str(crlf), ["ACGTGGCC", "TT"])
    # no trailing newline; header only; garbage before the first record; lower case
    t = tmp_path / "e.fa"
    t.write_text("garbage line\n>x\nacgtNNacgtacg")
    check(dump, str(t), ["ACGTACG"])
    t2 = tmp_path / "f.fa"
    t2.write_text(">x\nACGT\n>")
    check(dump, str(t2), ["ACGT"])
    t3 = tmp_path / "g.fq"
    t3.write_text("@x\nACGT\n+")
    check(dump, str(t3), [])
    t4 = tmp_path / "h.fa"
    t4.write_text(">only header")
    check(dump, str(t4), [""])
    e = tmp_path / "empty.fa"
    e.write_text("")
    check(dump, str(e), [])
    gz = tmp_path / "c.fa.gz"
    with gzip.open(gz, "wt") as f:
        f.write(">x\nACGTNACGTT\n>y\nAC\nGT\n")
    check(dump, str(gz), ["ACGTT", "ACGT"])


@pytest.mark.parametrize("kind", ["fasta", "fasta_multi", "fastq", "fastq_multi", "fastq_gz"])
def test_random_files(dump, tmp_path, kind):
    rng = random.Random(hash(kind) & 0xFFFF)
    recs = []
    for i in range(3000):
        L = rng.choice([0, 1, 31, 32, 33, 64, 100, 150, 151, 300])
        s = "".join(rng.choice("ACGTacgtNn.") if rng.random() < 0.02 else rng.choice("ACGT") for _ in range(L))
        recs.append(s)

    def wrap(s, w):
        return "\n".join(s[i:i + w] for i in range(0, len(s), w)) if s else ""
    out = []
    for i, s in enumerate(recs):
        if kind.startswith("fasta"):
            body = wrap(s, 60) if kind == "fasta_multi" else s
            out.append(">r%d desc\n%s\n" % (i, body))
        else:
            q = "".join(rng.choice("@>+IJ#5") for _ in s)
            if kind == "fastq_multi":
                out.append("@r%d\n%s\n+\n%s\n" % (i, wrap(s, 70), wrap(q, 70)))
            else:
                out.append("@r%d\n%s\n+r%d\n%s\n" % (i, s, i, q))
    text = "".join(out)
    p = tmp_path / ("x." + kind)
    if kind == "fastq_gz":
        with gzip.open(p, "wt") as f:
            f.write(text)
    else:
        p.write_text(text)
    serial, _ = dump(["--serial-lv", str(p)])
    from oracle import oracle as O
    assert [lv(x) for x in O.read_fastx_list(str(p))] == serial, "oracle reader"
    if kind != "fastq_multi":
        # (multi-line FASTQ: a wrapped quality line that starts with '@'/'+'/'>' is legal input whose kseq reading is what
        # the serial reader gives; the plain layouts have an obvious expectation)
        assert serial == [lv(s) for s in recs]
    for threads, block in ((1, 1 << 22), (8, 1 << 22), (7, 50000), (4, 4096)):
        fast, err = dump(["--fast", str(threads), str(block), str(p)])
        assert fast == serial, (kind, threads, block)


def test_two_files_and_stop(dump, tmp_path):
    a = tmp_path / "a.fq"
    a.write_text("@a\nACGT\n+\nIIII\n@b\nACGTAC\n+\nII\n@c\nGGGG\n+\nIIII\n")  # stops after a
    b = tmp_path / "b.fa"
    b.write_text(">x\nTTTT\n")
    fast, _ = dump(["--fast", "4", "16", str(a), str(b)])
    assert fast == ["ACGT", "TTTT"]  # the truncated record ends file a only; file b is read


def test_parsers_under_asan_ubsan(tmp_path):
    """Host sanitizer build (AddressSanitizer + UBSan) of the serial and the parallel parser on the corner cases and on
    tiny blocks (carry-over, chunk boundaries inside records): no report, same output."""
    from spades_for_blackbird_amd import build_host
    exe = build_host.build_sanitized()
    files = {}
    files["a.fa"] = ">r1 c\nACGT\nacgtnn\n\nGG\n>r2\nTTTT\n>empty\n>r3\nAC\n"
    files["b.fq"] = "@a\nACGT\n+\nIIII\n@b\nGGCC\nTT\n+b\nIIII\nII\n@c\nACGTAC\n+\nIII\n@d\nAAAA\n+\nIIII\n"
    files["c.fq"] = "@r@1\nACGTA\n+\n@IIII\n@r2\nCCCCC\n+\n>>>>>\n@r3\nGGGNGGGG\n+\n@@@@@@@@\n"
    files["d.fa"] = ">x\r\nACGT\r\nGGCC\r\n>y\r\nTT\r\n"
    files["e.fa"] = "garbage\n>x\nacgtNNacgtacg"
    files["f.fq"] = "@x\nACGT\n+"
    files["g.fa"] = ">only header"
    rng = random.Random(3)
    files["h.fa"] = "".join(">r%d\n%s\n" % (i, "".join(rng.choice("ACGTN") for _ in range(rng.choice([0, 1, 33, 150, 400]))))
                            for i in range(500))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for name, text in files.items():
        p = tmp_path / name
        p.write_bytes(text.encode())
        ser = subprocess.run([exe, "--serial-lv", str(p)], capture_output=True, text=True, env=env)
        assert ser.returncode == 0 and "Sanitizer" not in ser.stderr and "runtime error" not in ser.stderr, ser.stderr[-2000:]
        for threads, block in ((1, 1 << 20), (4, 5), (3, 64)):
            r = subprocess.run([exe, "--fast", str(threads), str(block), str(p)], capture_output=True, text=True, env=env)
            assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
            assert r.stdout == ser.stdout, (name, threads, block)
