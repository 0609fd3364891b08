"""Shared test helpers (pure Python, no reference code)."""
import gzip

import numpy as np

_COMP = str.maketrans("ACGTacgt", "TGCAtgca")


def rc(s):
    return s[::-1].translate(_COMP)


def read_fastq_gz(path):
    out = []
    with gzip.open(path, "rt") as f:
        while True:
            h = f.readline()
            if not h:
                break
            out.append(f.readline().strip())
            f.readline()
            f.readline()
    return out


def synth_reads(n_reads, read_len=150, genome_len=None, sub_rate=0.005, seed=42, n_rate=0.0):
    """Synthetic reads in the shape of SURVEY 8(d): uniform genome, uniform starts,
    random strand, substitutions; optional N injection for the LongestValid rule."""
    rng = np.random.default_rng(seed)
    if genome_len is None:
        genome_len = max(read_len + 1, n_reads * read_len // 50)
    g = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
    starts = rng.integers(0, genome_len - read_len + 1, size=n_reads)
    idx = starts[:, None] + np.arange(read_len)[None, :]
    r = g[idx]
    sub = rng.random(r.shape) < sub_rate
    r = np.where(sub, (r + rng.integers(1, 4, size=r.shape, dtype=np.uint8)) & 3, r).astype(np.uint8)
    flip = rng.random(n_reads) < 0.5
    r[flip] = (3 - r[flip])[:, ::-1]
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    asc = lut[r]
    if n_rate > 0:
        asc = np.where(rng.random(asc.shape) < n_rate, np.uint8(ord("N")), asc)
    return [bytes(row).decode() for row in asc]
