"""CLI wall time at a larger size: R x 150 bp FASTA on tmpfs -> spades-kmercount / spades-gbuilder --gfa"""
import sys, os, time, json, subprocess
sys.path.insert(0, '.')
import numpy as np
import spades_for_blackbird_amd as B
sys.argv = [sys.argv[0]] + sys.argv[1:]
R = int(sys.argv[1]); k = int(sys.argv[2])
import bench
d = "/dev/shm/bbk_e2e_big"
os.makedirs(d, exist_ok=True)
fa = os.path.join(d, "reads.fa")
ctx = B.Context(0)
t0 = time.time()
with open(fa, "wb") as f:
    done = 0
    part = 10_000_000
    i = 0
    while done < R:
        n = min(part, R - done)
        r = ctx.reads_synth(n, read_len=150, genome_len=R * 150 // 50, seed_genome=42, seed_reads=43 + i)
        blob, offs = r.to_ascii()
        r.free()
        arr = np.frombuffer(blob, dtype=np.uint8).reshape(n, 150)
        out = np.empty((n, 154), dtype=np.uint8)
        out[:, 0] = ord(">"); out[:, 1] = ord("r"); out[:, 2] = 10; out[:, 3:153] = arr; out[:, 153] = 10
        out.tofile(f)
        done += n; i += 1
ctx.close()
print("FASTA written: %.1f GB in %.1f s" % (os.path.getsize(fa) / 1e9, time.time() - t0), flush=True)
bins = os.path.join("spades_for_blackbird_amd", "bin")
res = {"reads": R, "k": k, "fasta_bytes": os.path.getsize(fa)}
res["kmercount"] = bench.run_cli([os.path.join(bins, "spades-kmercount"), "-k", str(k), "-t", "16", "-w", d, fa])
fk = os.path.join(d, "final_kmers")
if os.path.exists(fk):
    res["kmercount"]["final_kmers_bytes"] = os.path.getsize(fk); os.unlink(fk)
print(json.dumps(res["kmercount"]), flush=True)
if len(sys.argv) > 3:
    gfa = os.path.join(d, "g.gfa")
    res["gbuilder"] = bench.run_cli([os.path.join(bins, "spades-gbuilder"), fa, gfa, "-k", str(k), "-t", "16", "--gfa"])
    if os.path.exists(gfa):
        res["gbuilder"]["gfa_bytes"] = os.path.getsize(gfa)
        # structure of the whole file: every line is an S or an L line, S lines hold ACGT only (sampled), the counts
        # are the tool's own ("<n> sequences extracted")
        n_s = n_l = other = 0
        bad_seq = 0
        carry = b"\n"
        with open(gfa, "rb") as f:
            while True:
                buf = f.read(1 << 28)
                if not buf:
                    break
                a = np.frombuffer(carry[-1:] + buf, dtype=np.uint8)
                starts = a[1:][a[:-1] == 10]
                n_s += int((starts == ord("S")).sum()); n_l += int((starts == ord("L")).sum())
                other += int(((starts != ord("S")) & (starts != ord("L"))).sum())
                carry = buf
        with open(gfa, "rb") as f:  # the last 64 MB line by line: the tail is where a shortened launch would show
            f.seek(max(0, os.path.getsize(gfa) - (64 << 20)))
            tail = f.read().split(b"\n")[1:-1]
        for ln in tail:
            if ln.startswith(b"S"):
                sq = ln.split(b"\t")[2]
                bad_seq += int(len(sq.strip(b"ACGT")) != 0 or len(sq) == 0)
        res["gbuilder"]["gfa_check"] = {"S_lines": n_s, "L_lines": n_l, "other_lines": other, "bad_S_in_last_64MB": bad_seq,
                                        "ends_with_newline": carry[-1:] == b"\n"}
        os.unlink(gfa)
    print(json.dumps(res["gbuilder"]), flush=True)
os.unlink(fa)
json.dump(res, open("gpurun_out/e2e_big_%d.json" % R, "w"), indent=1)
