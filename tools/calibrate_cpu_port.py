"""Calibration of bench.py cpu_baseline: the oracle port timed in the build container (8 vCPU) on the
workload SURVEY section 6 timed the reference binaries on (1 M x 150 bp, 3 Mbp genome, 0.5 % substitutions, k=21, 8
threads): reference spades-kmercount 12.3 s wall; this script printed 8.48 s for the port -> ratio 1.45."""
import sys, time, os
sys.path.insert(0, '.')
import numpy as np
from oracle import oracle as O
from tests.helpers import synth_reads
n = 1_000_000
t0 = time.time()
rng = np.random.default_rng(42)
g = rng.integers(0, 4, size=3_000_000, dtype=np.uint8)
starts = rng.integers(0, len(g) - 150 + 1, size=n)
idx = starts[:, None] + np.arange(150)[None, :]
r = g[idx]
sub = rng.random(r.shape) < 0.005
r = np.where(sub, (r + rng.integers(1, 4, size=r.shape, dtype=np.uint8)) & 3, r).astype(np.uint8)
flip = rng.random(n) < 0.5
r[flip] = (3 - r[flip])[:, ::-1]
blob = np.frombuffer(b"ACGT", dtype=np.uint8)[r].tobytes()
offs = np.arange(n + 1, dtype=np.uint64) * 150
print("gen %.1fs" % (time.time() - t0), flush=True)
st = O.mk_reads_blob(blob, offs)
for thr in (8,):
    t0 = time.time()
    out = O.kmercount(None, 21, 16, thr, blob=st)
    dt = time.time() - t0
    print("port kmercount 1M reads k=21 threads=%d: %.2f s, %d distinct, %.2f M distinct/s" % (thr, dt, len(out), len(out) / dt / 1e6), flush=True)
