import numpy as np, sys
sys.path.insert(0, '.')
import spades_for_blackbird_amd as B
from oracle import oracle as O
from tests.helpers import synth_reads, rc
ctx = B.Context(0)
reads = synth_reads(50, read_len=100, genome_len=1000, seed=1)
for k in (33, 55, 64, 77):
    r = ctx.reads_from_ascii(reads)
    s = ctx.count(r, k, B.CANONICAL)
    keys = s.export(B.ORDER_SORTED)
    # expected canonical set
    exp = set()
    for t in reads:
        for i in range(len(t)-k+1):
            x = t[i:i+k]; c = min(x, rc(x)); exp.add(tuple(O.kmer_words(c)))
    got = [tuple(int(v) for v in row) for row in keys]
    print(k, "n got", len(got), "distinct got", len(set(got)), "exp", len(exp), "sorted", got == sorted(got), "inter", len(set(got) & exp))
    bad = [g for g in set(got) if g not in exp][:3]
    for b in bad:
        km = O._Kmer()
        for i, w in enumerate(b): km.w[i] = w
        print("  bad", O.kmer_str(km, k))
    s2 = ctx.count(r, k, B.BOTH_STRANDS)
    k2 = s2.export(B.ORDER_SORTED)
    exp2 = O.kmercount(reads, k, 16, 1)
    print("   both: got", len(k2), "exp", len(exp2))
