import numpy as np, sys, torch
sys.path.insert(0, '.')
import spades_for_blackbird_amd as B
from oracle import oracle as O
from tests.helpers import synth_reads, rc
ctx = B.Context(0)
rng = np.random.default_rng(0)
# pure sort+unique, W=2, k=33 (word1 has 2 bits) and k=55
for k in (33, 55, 21):
    nw = (k+31)//32
    for n in (5000, 50000, 300000):
        base = rng.integers(0, 2**63, size=(n//4, nw), dtype=np.uint64)
        vb = 2*k - 64*(nw-1)
        if vb < 64: base[:, nw-1] &= np.uint64((1 << vb) - 1)
        keys = base[rng.integers(0, len(base), size=n)]
        t = torch.from_numpy(keys.view(np.int64)).cuda()
        s = ctx.kmerset_from_device(t, n, k)
        got = s.export(B.ORDER_SORTED)
        exp = np.unique(keys, axis=0)
        order = np.lexsort([exp[:, j] for j in range(nw-1, -1, -1)])
        exp = exp[order]
        ok = got.shape == exp.shape and np.array_equal(got, exp)
        print("sort-unique k", k, "n", n, "got", len(got), "exp", len(exp), "OK" if ok else "FAIL")
# extraction check
reads = synth_reads(2000, read_len=100, genome_len=20000, sub_rate=0.005, seed=11)
for k in (33,):
    exp = set()
    for t in reads:
        for i in range(len(t)-k+1):
            x = t[i:i+k]; c = min(x, rc(x)); exp.add(tuple(O.kmer_words(c)))
    for rep in range(3):
        r = ctx.reads_from_ascii(reads)
        s = ctx.count(r, k, B.CANONICAL)
        keys = s.export(B.ORDER_SORTED)
        got = [tuple(int(v) for v in row) for row in keys]
        sg = set(got)
        print("extract k", k, "rep", rep, "n", len(got), "distinct", len(sg), "exp", len(exp), "garbage", len(sg-exp), "missing", len(exp-sg), "sorted", got == sorted(got))
