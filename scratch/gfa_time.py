import sys, time, os
sys.path.insert(0, '.')
import spades_for_blackbird_amd as B
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 21
cov = len(sys.argv) > 3
ctx = B.Context(0)
ctx.profile(True)
r = ctx.reads_synth(n, read_len=150, genome_len=n*150//50)
ctx.synchronize()
for rep in range(2):
    ctx.profile_reset()
    t0 = time.perf_counter()
    x = ctx.extindex(r, k); ctx.synchronize(); t1 = time.perf_counter()
    u = ctx.unitigs(x); t2 = time.perf_counter()
    if cov: u.add_coverage(r)
    t2b = time.perf_counter()
    p = "/dev/shm/bbk_test.gfa"
    u.write_gfa(p); t3 = time.perf_counter()
    sz = os.path.getsize(p); os.unlink(p)
    print("rep", rep, "reads", n, "k", k, "kmers", len(x), "unitigs", len(u), "loops", u.n_loops, "vertices", u.n_vertices, "links", u.n_links,
          "extindex %.3fs unitigs %.3fs cov %.3fs write %.3fs total %.3fs gfa_bytes %d" % (t1-t0, t2-t1, t2b-t2, t3-t2b, t3-t0, sz), flush=True)
    print({f: round(ctx.profile_get(f)["ms"],2) for f in ("part_hist1_reads","part_scatter1_reads","part_hist2","part_scatter2","lds_dedup","lds_sort","compact","hist","scatter","walk0","walk1","coverage")})
    u.free(); x.free()
