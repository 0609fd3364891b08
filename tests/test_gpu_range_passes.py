"""GPU: inputs above one device pass are split into hash ranges (msd.hip run_all).  The split is
forced on small inputs through BBK_PASS_LIMIT (read once per process, hence the subprocess)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

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
    assert r.stderr.count("msd mode=0") >= 8
