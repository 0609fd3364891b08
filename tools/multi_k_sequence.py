import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spades_for_blackbird_amd as B
ctx = B.Context(0)
def show(tag, dt, n):
    m = ctx.memory_stats()
    print("%-28s %7.1f ms  n=%d  mapped_now=%.1f GB mapped_total=%.1f GB map_s=%.3f" % (tag, dt * 1e3, n, m['mapped_now'] / 1e9, m['mapped_total'] / 1e9, m['map_seconds']), flush=True)
def loop(tag, reads, k, reps):
    s = None
    for it in range(reps):
        t0 = time.time()
        s = ctx.count(reads, k, B.BOTH_STRANDS | B.REFERENCE_ORDER)
        ctx.synchronize()
        show("%s k=%d it %d" % (tag, k, it), time.time() - t0, len(s))
    if s is not None: s.free()
main = ctx.reads_synth(10_000_000)
loop("uniform", main, 21, 4)
loop("uniform", main, 55, 3)
r = ctx.reads_synth_meta(10_000_000, read_len=150, seed=44)
for k in (21, 33, 55):
    loop("meta", r, k, 4)
