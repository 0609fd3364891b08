import sys, os, time
sys.path.insert(0, '.')
import spades_for_blackbird_amd as B
R = int(sys.argv[1]); k = int(sys.argv[2])
ctx = B.Context(0)
reads = ctx.reads_synth(R, read_len=150, genome_len=R * 150 // 50)
for i in range(3):
    s = ctx.count(reads, k, B.BOTH_STRANDS | B.REFERENCE_ORDER)
    n = len(s); s.free(); ctx.trim()
    print("rep", i, "count", n, flush=True)
    x = ctx.extindex(reads, k)
    nx = len(x); x.free(); ctx.trim()
    print("rep", i, "ext", nx, flush=True)
print("REPRO-OK")
