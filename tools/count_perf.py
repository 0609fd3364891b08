"""One count (both strands, final_kmers order) of synthetic reads in a loop, with the HIP-event time of every kernel family:
   python tools/count_perf.py <k> [reads]      (on the GPU box; BBK_NO_SUPERK=1 etc. select the A/B variants)"""
import os, sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import spades_for_blackbird_amd as B
k = int(sys.argv[1]) if len(sys.argv) > 1 else 55
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
flags = B.BOTH_STRANDS | B.REFERENCE_ORDER
if os.environ.get("WITH_COUNTS"): flags |= B.WITH_COUNTS
EXT = bool(os.environ.get("EXT_INDEX"))  # time the extension index (edge masks) instead of the count
ctx = B.Context(0)
reads = ctx.reads_synth(n)
def one():
    return ctx.extindex(reads, k) if EXT else ctx.count(reads, k, flags)
for it in range(3):
    s = one(); ctx.synchronize(); nn = len(s); del s
ctx.profile(True); ctx.profile_reset()
t0 = time.time()
R = 3
each = []
for it in range(R):
    t1 = time.time()
    s = one(); ctx.synchronize(); del s
    each.append((time.time() - t1) * 1e3)
dt = (time.time() - t0) / R
print(f"k={k} n={n} distinct={nn} ms/step={dt*1e3:.2f}  (each: {' '.join('%.2f' % e for e in each)})")
print("  memory:", ctx.memory_stats())
fams = ["k_sk_part1", "k_sk_part2_hist", "k_sk_part2", "k_sk_dedup", "k_part_reads_narrow", "k_part_narrow2", "k_part_reads", "k_part_l2", "k_bucket_hash32", "k_bucket_hash", "k_bucket_hashidx", "k_part_l1", "k_bucket_dist", "k_bucket", "k_part_hist1", "k_part_hist2", "compact", "stat_superk_records", "stat_superk_declined"]
for f in fams:
    d = ctx.profile_get(f); ms, cnt, by = d["ms"], d["launches"], d["bytes"]
    if cnt: print(f"  {f:24s} {ms/R:8.3f} ms/step  launches/step {cnt/R:5.1f}  bytes/step {by/R/1e9:8.3f} GB")
