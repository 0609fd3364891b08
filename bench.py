#!/usr/bin/env python3
"""bench.py -- headline benchmark of the k-mer counting path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--k 21] [--ext-index]

Metric (BASELINE.json): distinct k-mers/sec, k=21, 150 bp synthetic reads.
N=1 workload = BASELINE.json configs[1]: 10 M x 150 bp uniform reads (50x coverage of a
30 Mbp random genome, 0.5 % substitutions), k-mer count only, reads already packed in HBM.
One "step" = one full pass of the hot path over the batch: 2-bit extraction of canonical
k-mers -> hash-partitioned dedup -> both-strand expansion -> sort into the reference
(final_kmers) order, result left in HBM.  value = |final_kmers records| * N / time.
`--reads 100000000 --k 55 --ext-index` is BASELINE.json configs[2] (count + DeBruijnExtensionIndex build per step).

N>1: one rank per GPU over RCCL.  Launched by torch.distributed.run (RANK/WORLD_SIZE in the env) or, when
`python bench.py --gpus N` is called directly, this process starts the N ranks itself (before touching the GPU) and
returns their exit code.  Every rank holds its own 10 M reads of a common genome that grows with N (weak scaling);
ranks count locally, partition distinct canonical k-mers by owner hash, exchange them with ONE all_to_all_single (the
only collective on the data path), merge-unique their shard and expand it to both strands.

Extra objects on the JSON line (N=1): "roofline" for the dominant kernel, measured with HIP events on the engine's
stream inside the timed region; "k55": the same step at k=55 (16-byte keys) with its own roofline; "gfa_build": wall
seconds of extension index -> unitigs -> GFA on the same reads; "e2e": wall seconds of the two CLI binaries
(spades-kmercount, spades-gbuilder --gfa) on the same reads written as FASTA to tmpfs, with their phases, next to the
CPU oracle run end to end (parse included) on a bounded sample; "cpu_baseline": the CPU oracle -- a port of the
reference's split/sort/unique/merge algorithm -- timed on a bounded sample on this box's host cores.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md), the figure fractions are quoted against
# Timer names are kernel symbols (one name per kernel template, the way rocprofv3 lists them; tools/pmc_summary.py holds
# the symbol -> name table), so that "the dominant kernel" is a kernel and not a group of them.
FAMILIES = ["extract", "hist", "scan", "scatter", "unique", "reduce", "expand", "compact",
            "k_part_reads_narrow", "k_part_narrow2", "k_bucket_hash32",                      # stage A, 4-byte records (k 17..21)
            "k_part_reads", "k_part_reads_hist",                                              # stage A from reads, full keys
            "k_part_hist0", "k_part_l0", "k_part_hist1", "k_part_l1", "k_part_hist2", "k_part_l2",  # key arrays (stage B, merges)
            "k_bucket_hash", "k_bucket_hashidx", "k_bucket_dist", "k_bucket",                 # buckets finished in LDS
            "k_sk_part1", "k_sk_part2_hist", "k_sk_part2", "k_sk_dedup", "k_sk_dedup_B", "k_sk_expand"]  # super-k-mer records
# HBM traffic from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected in separate runs of this very
# command and corrected as MI355X_MICROARCH.md prescribes; tools/pmc_summary.py).  A profiler cannot run
# inside the timed process, so the committed summary is attached when the workload is the one it was
# measured on; otherwise "traffic" stays null.
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")
PMC_WORKLOAD = {"reads": 10_000_000, "read_len": 150, "k": 21, "gpus": 1}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--ext-index", action="store_true",
                    help="a step also builds the extension index (BASELINE configs[2] with --reads 100000000 --k 55)")
    ap.add_argument("--cpu-sample-reads", type=int, default=4_000_000)
    ap.add_argument("--e2e-cpu-sample-reads", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gfa", action="store_true", help="skip the (untimed-region) GFA-build wall-time measurement")
    ap.add_argument("--no-e2e", action="store_true", help="skip the CLI wall-time measurement on FASTA input")
    ap.add_argument("--no-k55", action="store_true", help="skip the k=55 (16-byte keys) object")
    ap.add_argument("--no-meta", action="store_true", help="skip the skewed-metagenome (configs[4] shape) object")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (nothing has touched the GPU in this
    process), hand their output through, return their exit code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_baseline(ctx, args, B):
    """Times the CPU oracle (port of the reference algorithm) on a bounded sample."""
    from oracle import oracle as O
    n = min(args.cpu_sample_reads, args.reads)
    r = ctx.reads_synth(n, read_len=args.read_len, genome_len=max(args.read_len, n * args.read_len // 50))
    blob, offs = r.to_ascii()
    r.free()
    cores = min(os.cpu_count() or 1, 64)
    st = O.mk_reads_blob(blob, offs)
    t0 = time.perf_counter()
    out = O.kmercount(None, args.k, 16, cores, blob=st)
    dt = time.perf_counter() - t0
    inst = 2 * n * (args.read_len - args.k + 1)
    return {
        "value": len(out) / dt, "unit": "distinct k-mers/s", "cores": cores, "kind": "port",
        "instances_per_s": inst / dt, "seconds": dt,
        # kind "port": the reference cannot be built on the GPU box (cmake-generated headers, bzlib.h).  The
        # only place where both ran is the 8-vCPU survey container: reference spades-kmercount -t 8 on 1 M x 150 bp,
        # k=21: 12.3 s wall; this port, 8 threads, same workload: 8.48 s (tools/calibrate_cpu_port.py).
        # That ratio is a statement about 8 threads and is NOT applied to `value`, which is the port at `cores` threads.
        "calibration": {"t_reference_over_t_port_at_8_threads": 1.45,
                        "measured_on": "8 vCPU container, 1 M x 150 bp, k=21, 8 threads (not this box, not applied)"},
        "sample": "%d x %d bp synthetic reads (same generator and 50x coverage as the GPU workload), "
                  "oracle/bbk_oracle.c orc_kmercount, 16 buckets, OpenMP %d threads (the sort phase can use 16: one per "
                  "bucket, as CountAll(16, ...)); parse not included (see e2e.cpu_port for the end-to-end figure)"
                  % (n, args.read_len, cores),
    }


def gfa_build(ctx, reads, k):
    """Second half of the metric: wall seconds of the spades-gbuilder path on the same reads (extension
    index -> unitigs + links -> GFA text written to tmpfs), measured outside the timed region."""
    import tempfile
    ctx.synchronize()
    d = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(d, "bbk_bench_%d.gfa" % os.getpid())
    best = None
    graph_fams = ["walk0", "walk1", "links", "gfa_text", "scatter", "hist", "scan"]

    def snap(fams):
        p = {f: ctx.profile_get(f) for f in fams}
        return {f: v for f, v in p.items() if v["launches"]}
    for it in range(2):  # first pass warms the allocator, second is reported (with HIP-event times per kernel family)
        ctx.profile(it == 1)
        ctx.profile_reset()
        t0 = time.perf_counter()
        x = ctx.extindex(reads, k)
        t1 = time.perf_counter()
        prof_ext = snap(FAMILIES) if it == 1 else {}
        ctx.profile_reset()
        u = ctx.unitigs(x)
        t2 = time.perf_counter()
        prof_uni = snap(graph_fams) if it == 1 else {}
        ctx.profile_reset()
        u.write_gfa(path)
        t3 = time.perf_counter()
        prof_wr = snap(graph_fams) if it == 1 else {}
        best = {"wall_s": t3 - t0, "extindex_s": t1 - t0, "unitigs_s": t2 - t1, "write_s": t3 - t2,
                "kmers": len(x), "unitigs": len(u), "vertices": u.n_vertices, "links": u.n_links,
                "gfa_bytes": os.path.getsize(path), "output": "GFA1 text on tmpfs",
                "extindex_kernel_ms": {f: v["ms"] for f, v in prof_ext.items()},
                # the extension index is the same sort-reduce with a mask payload: roofline of its dominant kernel
                "extindex_roofline": roofline_of(prof_ext, 1),
                "unitigs_kernel_ms": {f: v["ms"] for f, v in prof_uni.items()},
                "unitigs_lookup_gbs": {f: v["bytes"] / (v["ms"] * 1e-3) / 1e9 for f, v in prof_uni.items()
                                       if v["bytes"] and v["ms"]},
                "write_kernel_ms": {f: v["ms"] for f, v in prof_wr.items()}}
        os.unlink(path)
        u.free()
        x.free()
    ctx.profile(False)
    return best


def write_fasta(reads, path, n=None):
    """SURVEY 8(d) input format: `>r\\n<bases>\\n` per read (154 B for 150 bp)."""
    import numpy as np
    blob, offs = reads.to_ascii()
    nr = len(offs) - 1 if n is None else n
    L = int(offs[1] - offs[0]) if nr else 0
    arr = np.frombuffer(blob, dtype=np.uint8)[: nr * L].reshape(nr, L)
    out = np.empty((nr, L + 4), dtype=np.uint8)
    out[:, 0] = ord(">")
    out[:, 1] = ord("r")
    out[:, 2] = 10
    out[:, 3:3 + L] = arr
    out[:, 3 + L] = 10
    out.tofile(path)
    return nr


def run_cli(cmd, timeout=None):
    env = dict(os.environ, BBK_PHASES="1")
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": "timed out after %s s" % timeout, "wall_s": time.perf_counter() - t0}
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        return {"error": (r.stderr or r.stdout)[-400:], "wall_s": wall}
    ph = None
    for line in r.stdout.splitlines():
        if line.startswith("BBK_PHASES "):
            ph = json.loads(line[len("BBK_PHASES "):])
    return {"wall_s": wall, "phases": ph}


def cpp_multi_gpu(ctx, args, world):
    """N > 1 only, rank 0, after the timed region: the C++ one-process host (`spades-kmercount --devices 0,..,N-1`:
    one host thread + context per GPU, owner-hash shards, ONE grouped ncclSend/ncclRecv all-to-all, bucket-wise merge
    into one final_kmers) on the node's GPUs, beside the single-device tool on the same FASTA: wall times and whether
    the two files are the same bytes.  The only place where the RCCL path of the C++ host meets more than one GPU (the
    build box has one); any failure is reported in the object, never raised."""
    import hashlib
    import tempfile
    from spades_for_blackbird_amd import build_host
    out = {"devices": world}
    d = tempfile.mkdtemp(prefix="bbk_cpp_multi_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        n = min(args.reads, 2_000_000) * world
        r = ctx.reads_synth(n, read_len=args.read_len, genome_len=max(args.read_len, n * args.read_len // 50), seed_reads=77)
        fa = os.path.join(d, "reads.fa")
        write_fasta(r, fa)
        r.free()
        ctx.trim()
        out["reads"] = n
        bins = {os.path.basename(p): p for p in build_host.build()}

        def md5(path):
            h = hashlib.md5()
            with open(path, "rb") as f:
                for b in iter(lambda: f.read(1 << 26), b""):
                    h.update(b)
            return h.hexdigest()
        devs = ",".join(str(i) for i in range(world))
        # BBK_BENCH_CPP_MULTI_ARGS="--devices 0,0 --exchange copy": rehearsal on a box with fewer GPUs than ranks
        multi = os.environ.get("BBK_BENCH_CPP_MULTI_ARGS", "--devices " + devs).split()
        out["args"] = " ".join(multi)
        res = {}
        for name, extra in (("single_device", ["--device", "0"]), ("all_devices_rccl", multi)):
            w = os.path.join(d, name)
            os.makedirs(w, exist_ok=True)
            res[name] = run_cli([bins["spades-kmercount"], "-k", str(args.k), "-t", "16", "-w", w, fa] + extra, timeout=300)
            fk = os.path.join(w, "final_kmers")
            if os.path.exists(fk):
                res[name]["final_kmers_bytes"] = os.path.getsize(fk)
                res[name]["md5"] = md5(fk)
                os.unlink(fk)
        out["kmercount"] = res
        out["final_kmers_identical"] = bool(res["single_device"].get("md5")) and \
            res["single_device"].get("md5") == res["all_devices_rccl"].get("md5")
        gfa = os.path.join(d, "g.gfa")
        out["gbuilder_all_devices_rccl"] = run_cli([bins["spades-gbuilder"], fa, gfa, "-k", str(args.k), "-t", "16", "--gfa",
                                                    ] + multi, timeout=300)
        if os.path.exists(gfa):
            out["gbuilder_all_devices_rccl"]["gfa_bytes"] = os.path.getsize(gfa)
    except Exception as ex:
        out["error"] = repr(ex)[:300]
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)
    return out


def e2e(ctx, reads, args):
    """What north_star's target is stated on: wall time of the two CLIs on FASTA input, beside the CPU path.
    The GPU step's reads are written as FASTA to tmpfs; the built binaries are run as a user would run them
    (process start, HIP context creation, parse, upload, device work, download, file write all inside the wall time)."""
    import tempfile
    from spades_for_blackbird_amd import build_host
    k = args.k
    d = tempfile.mkdtemp(prefix="bbk_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    out = {"input": None}
    # the CLIs are separate processes on the same GPU: hand this process's cached device memory back first, or the
    # driver has to evict it to make room for theirs (seen as a 1 s "device" phase in the child)
    ctx.trim()
    try:
        import torch
        torch.cuda.empty_cache()
    except Exception:
        pass
    try:
        fa = os.path.join(d, "reads.fa")
        n = write_fasta(reads, fa)
        out["input"] = {"format": "FASTA (>r\\n<150 bases>\\n)", "reads": n, "bytes": os.path.getsize(fa), "on": "tmpfs"}
        bins = {os.path.basename(p): p for p in build_host.build()}
        threads = min(os.cpu_count() or 1, 16)
        out["kmercount"] = run_cli([bins["spades-kmercount"], "-k", str(k), "-t", str(threads), "-w", d, fa])
        fk = os.path.join(d, "final_kmers")
        if os.path.exists(fk):
            out["kmercount"]["final_kmers_bytes"] = os.path.getsize(fk)
            os.unlink(fk)
        gfa = os.path.join(d, "g.gfa")
        out["gbuilder"] = run_cli([bins["spades-gbuilder"], fa, gfa, "-k", str(k), "-t", str(threads), "--gfa"])
        if os.path.exists(gfa):
            out["gbuilder"]["gfa_bytes"] = os.path.getsize(gfa)
            os.unlink(gfa)
        os.unlink(fa)
        # the CPU port end to end on a bounded sample of the same reads: parse (serial, like the reference's master
        # thread) + count + file write; parse + extension index + unitigs + GFA write
        from oracle import oracle as O
        ns = min(args.e2e_cpu_sample_reads, n)
        sample = ctx.reads_synth(ns, read_len=args.read_len, genome_len=max(args.read_len, ns * args.read_len // 50))
        sfa = os.path.join(d, "sample.fa")
        write_fasta(sample, sfa)
        sample.free()
        cores = min(os.cpu_count() or 1, 64)
        t0 = time.perf_counter()
        blob, offs = O.read_fastx(sfa)
        t1 = time.perf_counter()
        st = O.mk_reads_blob(blob, offs)
        km = O.kmercount(None, k, 16, cores, blob=st)
        km.tofile(os.path.join(d, "cpu_final_kmers"))
        t2 = time.perf_counter()
        tg = max(1, cores // 2 + 1)  # gbuilder's default -t (projects/gbuilder/main.cpp:47-87: omp_max / 2 + 1)
        x = O.ExtIndex(None, k, tg, blob=st)
        u = x.unitigs(threads=tg)  # path extraction over 16 * t chunks in parallel, as the reference
                                   # (debruijn_graph_constructor.hpp:351-375)
        text = u.gfa()[0]
        with open(os.path.join(d, "cpu.gfa"), "w") as f:
            f.write(text)
        t3 = time.perf_counter()
        out["cpu_port"] = {"kind": "port", "cores": cores, "sample_reads": ns, "parse_s": t1 - t0,
                           "kmercount_wall_s": t2 - t0, "gbuilder_wall_s": (t1 - t0) + (t3 - t2),
                           "gbuilder_threads": tg,
                           "note": "oracle/bbk_oracle.c end to end on a %d-read sample of the same generator "
                                   "(serial kseq-style parse as in the reference, OpenMP count / extension index, "
                                   "unbranching paths over 16 x t chunks in parallel like the reference's "
                                   "ExtractUnbranchingPaths, sequential loop collection and GFA text like the "
                                   "reference's); a port, not the reference binary: scale by reads/sample_reads to "
                                   "compare with the CLI walls" % ns}
        if "wall_s" in out["kmercount"] and "error" not in out["kmercount"]:
            out["speedup_vs_cpu_port_scaled"] = {
                "kmercount": out["cpu_port"]["kmercount_wall_s"] * (n / ns) / out["kmercount"]["wall_s"],
                "gbuilder": (out["cpu_port"]["gbuilder_wall_s"] * (n / ns) / out["gbuilder"]["wall_s"])
                if "error" not in out["gbuilder"] else None}
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)
    return out


def metagenome(ctx, args, B, fence):
    """BASELINE configs[4] shape at one-GPU size: the same number of reads drawn from a skewed community (200 genomes,
    lengths log-uniform 0.5-8 Mbp, abundances log-normal sigma=2, seed 44: SURVEY 8d), multi-k {21, 33, 55}; per k the
    step time and what the histogram-free slot mode spilled / had to reprocess under real skew."""
    r = ctx.reads_synth_meta(args.reads, read_len=args.read_len, seed=44)
    out = {"workload": "%d x %d bp reads of a synthetic metagenome: 200 genomes, lengths log-uniform 0.5-8 Mbp, "
                       "abundances log-normal(sigma=2), 0.5 %% substitutions, seed 44; k-mer count only, 1 GPU"
                       % (args.reads, args.read_len), "per_k": {}}
    stats = ("stat_slot_records", "stat_slot_spilled", "stat_slot_overflow_segments", "stat_slot_overflow_buckets",
             "stat_slot_reprocessed")
    for kk in (21, 33, 55):
        def st():
            s = ctx.count(r, kk, B.BOTH_STRANDS | B.REFERENCE_ORDER)
            return len(s), s
        # two untimed steps: the first sizes the result of this k, the second still makes the allocator map memory for it
        # (tens of GB at k = 33; on memory the driver has to clear first -- e.g. what the CLI children of `e2e` just
        # released -- that is ~30 ms/GiB and was the "159.7 ms with 50.4 ms of kernels" of the round-2 driver run).
        # What mapping is left inside the timed steps is reported.
        map0 = ctx.memory_stats()["map_seconds"]
        n, dt, prof = timed_steps(ctx, st, fence, 2, 2)
        map_in_timed = ctx.memory_stats()["map_seconds"] - map0
        sv = {f: ctx.profile_get(f) for f in stats}
        rec = max(1.0, sv["stat_slot_records"]["bytes"])
        out["per_k"][str(kk)] = {
            "ms_per_step": dt / 2 * 1e3, "distinct_kmers": n, "value": n / (dt / 2), "unit": "distinct k-mers/s",
            "memory_map_ms_untimed_and_timed": map_in_timed * 1e3,
            "slot_mode": {"passes": sv["stat_slot_records"]["launches"] / 2,
                          "spilled_frac": sv["stat_slot_spilled"]["bytes"] / rec,
                          "reprocessed_frac": sv["stat_slot_reprocessed"]["bytes"] / rec,
                          "overflow_segments_per_step": sv["stat_slot_overflow_segments"]["bytes"] / 2,
                          "overflow_buckets_per_step": sv["stat_slot_overflow_buckets"]["bytes"] / 2},
            "kernel_ms_per_step": {f: v["ms"] / 2 for f, v in prof.items()}}
    r.free()
    return out


def roofline_of(prof, steps, traffic_lookup=None, step_ms=None):
    """`roofline` object of one timed loop.  Keyed by KERNEL (the timer names are kernel symbols): the dominant kernel is
    the one with the largest summed duration over the timed steps; `kernels` lists every kernel of the step the same
    way (so the object still holds the figures when two kernels are close and the dominant one changes between
    runs), `step` is the whole step: sum of the kernels' algorithmic bytes over the wall time of a step."""
    if not prof:
        return None

    def entry(name):
        p = prof[name]
        ach = p["bytes"] / (p["ms"] * 1e-3) / 1e9 if p["ms"] > 0 else 0.0
        traffic, traffic_src = (traffic_lookup(name) if traffic_lookup else (None, None))
        return {"kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src, "launches": p["launches"],
                "avg_launch_ms": p["ms"] / p["launches"], "ms_per_step": p["ms"] / steps,
                "algorithmic_bytes_per_launch": p["bytes"] / p["launches"]}

    order = sorted(prof, key=lambda f: -prof[f]["ms"])
    out = {"bound": "hbm"}
    out.update(entry(order[0]))
    out["kernels"] = [entry(f) for f in order if prof[f]["bytes"] > 0]
    tot_bytes = sum(v["bytes"] for v in prof.values()) / steps
    if step_ms:
        out["step"] = {"algorithmic_bytes": tot_bytes, "ms": step_ms, "achieved": tot_bytes / (step_ms * 1e-3) / 1e9,
                       "frac": tot_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "kernel_ms": sum(v["ms"] for v in prof.values()) / steps}
    if order[0].startswith("k_sk_dedup"):
        # the in-LDS table walk over super-k-mer records: vector-instruction bound (SQ counters), it
        # reads 1/13 of the bytes the k-mer path's dedup read -- the HBM fraction says how little it moves, not how busy it is
        out["note"] = "VALU/LDS-bound kernel (in-LDS dedup of packed super-k-mer records); quoted against HBM only because " \
                      "the contract has no other axis for this path"
    return out


SETTLE_STEPS = 0


def timed_steps(ctx, step, fence, warmup, steps, settle_allocator=False):
    def drop(x):
        for y in (x if isinstance(x, (list, tuple)) else [x]):
            if hasattr(y, "free"):
                y.free()
    n_rec = 0
    # Untimed, before the W warm-up steps: repeat the step until the engine's allocator has stopped mapping device
    # memory (the first steps of a workload size their buffers; mapping costs 1.4-30 ms per GiB) -- at most
    # four times.  What a long-running caller sees is the settled state; the count is reported as allocator_settle_steps.
    global SETTLE_STEPS
    settle = 0
    if settle_allocator:
        mapped = -1
        fixed = settle_allocator if isinstance(settle_allocator, int) and not isinstance(settle_allocator, bool) else None
        while settle < (fixed if fixed is not None else 4):
            m = ctx.memory_stats()["mapped_total"]
            if fixed is None and m == mapped:  # (several ranks: a fixed count, every rank runs the same collectives)
                break
            mapped = m
            n_rec, keep = step()
            drop(keep)
            del keep
            settle += 1
    SETTLE_STEPS = settle
    for _ in range(warmup):
        n_rec, keep = step()
        drop(keep)
        del keep
    ctx.profile(True)
    ctx.profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        n_rec, keep = step()
        drop(keep)
        del keep
    fence()
    dt = time.perf_counter() - t0
    ctx.profile(False)
    prof = {f: ctx.profile_get(f) for f in FAMILIES}
    prof = {f: v for f, v in prof.items() if v["launches"]}
    return n_rec, dt, prof


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    import torch
    import torch.distributed as dist
    import spades_for_blackbird_amd as B
    from spades_for_blackbird_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        return 2
    # BBK_BENCH_FORCE_SHARDED=1: take the N>1 code path (owner partition, RCCL all_to_all, merge, expand) with a
    # single rank -- the only way to exercise it on a one-GPU box
    sharded = world > 1 or os.environ.get("BBK_BENCH_FORCE_SHARDED") == "1"
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # BBK_BENCH_BACKEND=gloo: rehearsal of the N>1 code path on a box with fewer GPUs than ranks (the ranks share
        # the devices, the exchange is staged through the host); the measured path is always nccl = RCCL
        dist.init_process_group(os.environ.get("BBK_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    ctx = B.Context(dev_index, stream=torch.cuda.current_stream())

    k, L = args.k, args.read_len
    total_reads = args.reads * world
    genome_len = max(L, total_reads * L // 50)
    reads = ctx.reads_synth(args.reads, read_len=L, genome_len=genome_len, seed_genome=42, seed_reads=43 + rank)
    torch.cuda.synchronize()

    def step():
        """Returns (#records of this rank's part of the result, keep-alive)."""
        if not sharded and args.ext_index:
            # BASELINE configs[2]: count + DeBruijnExtensionIndex of the same reads from ONE pass over them
            # (bbk_count_extindex: stage A once with the mask payload; BBK_BENCH_SEPARATE=1: the two calls of round 2)
            if os.environ.get("BBK_BENCH_SEPARATE") == "1":
                s = ctx.count(reads, k, B.BOTH_STRANDS | B.REFERENCE_ORDER)
                n = len(s)
                s.free()
                return n, ctx.extindex(reads, k)
            s, x = ctx.count_extindex(reads, k, B.BOTH_STRANDS | B.REFERENCE_ORDER)
            return len(s), [s, x]
        if not sharded:
            # the set is built in the final_kmers order (what spades-kmercount leaves on disk) and stays in HBM
            s = ctx.count(reads, k, B.BOTH_STRANDS | B.REFERENCE_ORDER)
            ptr, order = s.device_keys()
            assert order == B.ORDER_REFERENCE_BUCKETS16 and (ptr or len(s) == 0)
            return len(s), s
        both = D.sharded_count(ctx, reads, k, both_strands=True, reference_order=True)
        ptr, order = both.device_keys()
        assert order == B.ORDER_REFERENCE_BUCKETS16
        return len(both), both

    def fence():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    n_rec, dt, prof = timed_steps(ctx, step, fence, args.warmup, args.steps, settle_allocator=(2 if sharded else True))
    settle_steps = SETTLE_STEPS

    red_dev = dev if (not sharded or dist.get_backend() == "nccl") else torch.device("cpu")
    tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    nn = torch.tensor([n_rec], dtype=torch.int64, device=red_dev)
    if sharded:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
    dt_max = float(tt.item())
    distinct_total = int(nn.item())

    # N>1: the second half of the metric over all ranks (configs[3] shape): sharded extension index -> gathered on
    # rank 0 -> unitigs + links -> GFA text on tmpfs; one un-timed-region measurement, every rank takes part
    gfa_sharded = None
    if world > 1 and not args.no_gfa:
        import tempfile
        d = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
        path = os.path.join(d, "bbk_bench_sharded_%d.gfa" % os.getpid())
        fence()
        t0 = time.perf_counter()
        u = D.sharded_gfa(ctx, reads, k, path, dst=0)
        fence()
        t1 = time.perf_counter()
        if rank == 0:
            gfa_sharded = {"wall_s": t1 - t0, "unitigs": len(u), "vertices": u.n_vertices, "links": u.n_links,
                           "gfa_bytes": os.path.getsize(path), "ranks": world,
                           "output": "GFA1 text on tmpfs, written by rank 0 from the gathered extension index"}
            os.unlink(path)
            u.free()

    # N > 1: the C++ one-process host over RCCL on the same GPUs, once, outside every timed region (rank 0 runs the
    # tools; all ranks hand their device memory back first and wait)
    cpp_multi = None
    if world > 1 and not args.no_e2e and os.environ.get("BBK_BENCH_CPP_MULTI", "1") != "0":
        try:
            ctx.trim()
            torch.cuda.empty_cache()
            fence()
            # the other ranks wait on the HOST (a gloo barrier): an RCCL barrier would park a spinning kernel on the very
            # GPUs the tools are about to use
            cpu_group = dist.new_group(backend="gloo")
            dist.barrier(group=cpu_group)
            if rank == 0:
                cpp_multi = cpp_multi_gpu(ctx, args, world)
            dist.barrier(group=cpu_group)
        except Exception as ex:
            cpp_multi = {"error": repr(ex)[:300]}

    if rank == 0:
        def pmc(dom):
            same = (args.reads, L, k, world, args.ext_index) == (PMC_WORKLOAD["reads"], PMC_WORKLOAD["read_len"],
                                                                 PMC_WORKLOAD["k"], PMC_WORKLOAD["gpus"], False)
            if same and os.path.exists(PMC_FILE):
                fam = json.load(open(PMC_FILE)).get("families", {}).get(dom)
                if fam:
                    return fam["hbm_bytes_per_launch"], "profiles/pmc_traffic.json (rocprofv3 --pmc)"
            return None, None
        roof = roofline_of(prof, args.steps, pmc, dt_max / args.steps * 1e3)
        inst_per_gpu = 2 * args.reads * (L - k + 1)
        if args.ext_index:
            wl = ("BASELINE.json configs[2]: synthetic %d x %d bp uniform reads, k=%d, k-mer count (both strands, "
                  "final_kmers order) + DeBruijnExtensionIndex build, 1 MI355X" % (args.reads, L, k)) \
                if (args.reads, k) == (100_000_000, 55) else \
                ("synthetic %d x %d bp uniform reads per GPU, k=%d, k-mer count + extension index" % (args.reads, L, k))
        else:
            wl = "BASELINE.json configs[1]: synthetic %d x %d bp uniform reads per GPU, k=%d, k-mer count only " \
                 "(both strands, final_kmers order)" % (args.reads, L, k)
        line = {
            "metric": "distinct k-mers/sec (k=%d, %d bp synthetic reads, %s)" % (
                k, L, "count + extension index" if args.ext_index else "count only"),
            "value": distinct_total * 1.0 / (dt_max / args.steps),
            "unit": "distinct k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "allocator_settle_steps": settle_steps,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": wl, "reads_per_gpu": args.reads, "read_len": L, "k": k, "genome_len": genome_len,
                       "coverage": 50, "parallelism": "owner-hash shards, %d rank(s)" % world,
                       "device": ctx.device_info()},
            "distinct_kmers": distinct_total,
            "kmer_instances_per_s": inst_per_gpu * world / (dt_max / args.steps),
            "kernel_ms_per_step": {f: v["ms"] / args.steps for f, v in prof.items()},
            "roofline": roof,
        }
        big = args.reads > 20_000_000
        # the CLIs first: they are separate processes on the same GPU, and with the tens of GB that the k = 55 and
        # metagenome objects leave mapped in this process their device phase was intermittently 0.3-0.5 s instead of 0.03 s
        if world == 1 and not args.no_e2e and not big:
            try:
                line["e2e"] = e2e(ctx, reads, args)
            except Exception as ex:  # the headline must not die with a side measurement
                line["e2e"] = {"error": repr(ex)[:300]}
        if world == 1 and not args.no_k55 and not big and k != 55:
            # the 16-byte-key path in front of the driver: the same step at k=55 on the same reads
            def step55():
                s = ctx.count(reads, 55, B.BOTH_STRANDS | B.REFERENCE_ORDER)
                return len(s), s
            n55, dt55, prof55 = timed_steps(ctx, step55, fence, 1, 2)
            line["k55"] = {"workload": "the same %d reads, k=55 (16-byte keys), k-mer count only" % args.reads,
                           "steps": 2, "warmup": 1, "ms_per_step": dt55 / 2 * 1e3, "distinct_kmers": n55,
                           "value": n55 / (dt55 / 2), "unit": "distinct k-mers/s", "dtype": "u128",
                           "kernel_ms_per_step": {f: v["ms"] / 2 for f, v in prof55.items()},
                           "roofline": roofline_of(prof55, 2, None, dt55 / 2 * 1e3)}
        if world == 1 and not args.no_meta and not big:
            line["metagenome"] = metagenome(ctx, args, B, fence)
        if world == 1 and not args.no_gfa and not big:
            line["gfa_build"] = gfa_build(ctx, reads, k)
        if gfa_sharded:
            line["gfa_build"] = gfa_sharded
        if cpp_multi:
            line["cpp_one_process_multi_gpu"] = cpp_multi
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ctx, args, B)
        print(json.dumps(line), flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
