import sys, time, os
sys.path.insert(0, '.')
import spades_for_blackbird_amd as B
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ctx = B.Context(0)
ctx.profile(True)
r = ctx.reads_synth(n, read_len=150, genome_len=n*150//50)
ctx.synchronize()
for rep in range(2):
    ctx.profile_reset()
    t0 = time.perf_counter()
    x = ctx.extindex(r, 21); ctx.synchronize(); t1 = time.perf_counter()
    u = ctx.unitigs(x); t2 = time.perf_counter()
    p = "/dev/shm/bbk_test.gfa"
    u.write_gfa(p); t3 = time.perf_counter()
    sz = os.path.getsize(p); os.unlink(p)
    print("rep", rep, "reads", n, "kmers", len(x), "unitigs", len(u), "loops", u.n_loops, "vertices", u.n_vertices, "links", u.n_links,
          "extindex %.3fs unitigs %.3fs write %.3fs total %.3fs gfa_bytes %d" % (t1-t0, t2-t1, t3-t2, t3-t0, sz), flush=True)
    print({f: ctx.profile_get(f) for f in ("extract","hist","scan","scatter","unique","reduce","walk0","walk1")})
    u.free(); x.free()
