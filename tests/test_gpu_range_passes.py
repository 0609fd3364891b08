"""GPU: inputs above one device pass are split into hash ranges (msd.hip run_all).  The split is
forced on small inputs through BBK_PASS_LIMIT (read once per process, hence the subprocess)."""
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(bool(os.environ.get("BBK_DISABLE_MSD")), reason="tests of the MSD path's modes")]

SCRIPT = r"""
import numpy as np, sys
sys.path.insert(0, %(root)r)
import spades_for_blackbird_amd as B
from oracle import oracle as O
from tests.helpers import synth_reads
ctx = B.Context(0)
reads = synth_reads(3000, read_len=150, genome_len=20000, sub_rate=0.01, seed=3, n_rate=0.001)
for k in (21, 33):
    r = ctx.reads_from_ascii(reads)
    got, gc = ctx.count(r, k, B.BOTH_STRANDS | B.WITH_COUNTS).export(B.ORDER_REFERENCE_BUCKETS16, with_counts=True)
    exp, ec = O.kmercount(reads, k, 16, 2, with_counts=True)
    assert np.array_equal(got, exp) and np.array_equal(gc, ec), k
    x = ctx.extindex(r, k)
    ox = O.ExtIndex(reads, k, 1)
    order = np.lexsort([ox.kmers[:, j] for j in range(ox.kmers.shape[1] - 1, -1, -1)])
    gk, gm = x.export()
    assert np.array_equal(gk, ox.kmers[order]) and np.array_equal(gm, ox.masks[order]), k
print("RANGE-PASSES-OK")
"""


def test_forced_hash_range_passes():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BBK_PASS_LIMIT="40000", BBK_VERBOSE="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": root}], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RANGE-PASSES-OK" in r.stdout
    # the verbose log shows more than one stage-A pass per call
    assert r.stderr.count("msd mode=0") + r.stderr.count("msd slots") >= 8


SLOT_SCRIPT = r"""
import numpy as np, sys
sys.path.insert(0, %(root)r)
import spades_for_blackbird_amd as B
from oracle import oracle as O
from tests.helpers import synth_reads
ctx = B.Context(0)
reads = synth_reads(4000, read_len=150, genome_len=30000, sub_rate=0.01, seed=5, n_rate=0.001)
# heavy repeats: single k-mers with tens of thousands of instances overflow their bucket / segment slots
reads += ["A" * 150] * 400 + ["ACGT" * 37] * 300 + ["AC" * 75] * 200 + ["ACGGTCA" * 21] * 150
for k in (21, 33, 19):
    r = ctx.reads_from_ascii(reads)
    for flags in (B.BOTH_STRANDS | B.WITH_COUNTS, B.BOTH_STRANDS | B.REFERENCE_ORDER):
        s = ctx.count(r, k, flags)
        exp, ec = O.kmercount(reads, k, 16, 2, with_counts=True)
        if flags & B.WITH_COUNTS:
            got, gc = s.export(B.ORDER_REFERENCE_BUCKETS16, with_counts=True)
            assert np.array_equal(gc, ec), k
        else:
            got = s.export(B.ORDER_REFERENCE_BUCKETS16)
        assert np.array_equal(got, exp), k
    u = ctx.count(r, k, B.CANONICAL | B.UNSORTED | B.WITH_COUNTS)
    canon = ctx.count(r, k, B.CANONICAL | B.WITH_COUNTS)
    uk, uc = u.export_by_owner(1, dst_keys=None)[0], None
    ck = canon.export(B.ORDER_SORTED)
    assert len(u) == len(canon)
    a = np.array(sorted(map(tuple, uk.tolist())), dtype=np.uint64).reshape(-1, uk.shape[1])
    assert np.array_equal(a, np.array(sorted(map(tuple, ck.tolist())), dtype=np.uint64).reshape(-1, ck.shape[1])), k
    x = ctx.extindex(r, k)
    ox = O.ExtIndex(reads, k, 1)
    order = np.lexsort([ox.kmers[:, j] for j in range(ox.kmers.shape[1] - 1, -1, -1)])
    gk, gm = x.export()
    assert np.array_equal(gk, ox.kmers[order]) and np.array_equal(gm, ox.masks[order]), k
print("SLOTS-OK")
"""


def test_forced_slot_mode_with_overflow():
    """The histogram-free slot mode (msd.hip) forced on a small input whose repeats overflow bucket and segment
    slots: the spill list and the overflowing slots are reprocessed by the exact path; results must equal the
    oracle bit for bit (both-strand set with counts, reference order, unsorted canonical set, extension index)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BBK_SLOTS_MIN="0", BBK_VERBOSE="1")
    env.pop("BBK_NO_SLOTS", None)  # this test is about the slot mode
    r = subprocess.run([sys.executable, "-c", SLOT_SCRIPT % {"root": root}], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "SLOTS-OK" in r.stdout
    lines = [l for l in r.stderr.splitlines() if "msd slots" in l and " N=" in l]
    assert lines, "the slot mode did not run"
    assert any("spill=0 " not in l for l in lines), "no slot overflowed: the test input is too tame"
