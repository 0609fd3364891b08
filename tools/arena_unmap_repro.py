"""Sequence driver for the allocator diagnostics: python tools/arena_unmap_repro.py R k "c t x t c" ...
   c = both-strand count in final_kmers order, x = extension index, t = bbk_ctx_trim, C = canonical count with counts."""
import sys, os, time
sys.path.insert(0, '.')
import spades_for_blackbird_amd as B
R = int(sys.argv[1]); k = int(sys.argv[2])
seq = (sys.argv[3] if len(sys.argv) > 3 else "c t x t c t x t").split()
ctx = B.Context(0)
reads = ctx.reads_synth(R, read_len=150, genome_len=R * 150 // 50)
for i, op in enumerate(seq):
    t0 = time.time()
    if op == "c":
        s = ctx.count(reads, k, B.BOTH_STRANDS | B.REFERENCE_ORDER); n = len(s); s.free()
    elif op == "C":
        s = ctx.count(reads, k, B.CANONICAL | B.UNSORTED | B.WITH_COUNTS); n = len(s); s.free()
    elif op == "x":
        x = ctx.extindex(reads, k); n = len(x); x.free()
    elif op == "t":
        ctx.trim(); n = 0
    print("step", i, op, n, "%.2f s" % (time.time() - t0), flush=True)
print("REPRO-OK")
