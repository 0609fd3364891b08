#!/usr/bin/env python3
"""bench.py -- headline benchmark of the k-mer counting path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--k 21]

Metric (BASELINE.json): distinct k-mers/sec, k=21, 150 bp synthetic reads.
N=1 workload = BASELINE.json configs[1]: 10 M x 150 bp uniform reads (50x coverage of a
30 Mbp random genome, 0.5 % substitutions), k-mer count only, reads already packed in HBM.
One "step" = one full pass of the hot path over the batch: 2-bit extraction of canonical
k-mers -> hash-partitioned dedup -> both-strand expansion -> sort into the reference
(final_kmers) order, result left in HBM.  value = |final_kmers records| * N / time.

N>1 (launched by torch.distributed.run, one rank per GPU, RCCL): every rank holds its own
10 M reads of a common genome that grows with N (weak scaling); ranks count locally, partition
distinct canonical k-mers by owner hash, exchange them with ONE all_to_all_single (the only
collective on the data path), merge-unique their shard and expand it to both strands.

Extra objects on the JSON line: "roofline" for the dominant kernel (the radix scatter pass),
measured with HIP events on the engine's stream inside the timed region, and "cpu_baseline"
(the CPU oracle -- a port of the reference's split/sort/unique/merge algorithm -- timed on a
bounded sample of the same workload on this box's host cores, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md), the figure fractions are quoted against
FAMILIES = ["extract", "hist", "scan", "scatter", "unique", "reduce", "expand", "part_hist1_reads", "part_hist1_keys",
            "part_scatter1_reads", "part_scatter1_keys", "part_hist2", "part_scatter2", "lds_dedup", "lds_sort",
            "compact"]
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
    ap.add_argument("--cpu-sample-reads", type=int, default=4_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gfa", action="store_true", help="skip the (untimed-region) GFA-build wall-time measurement")
    return ap.parse_args()


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
        "sample": "%d x %d bp synthetic reads (same generator and 50x coverage as the GPU workload), "
                  "oracle/bbk_oracle.c orc_kmercount, 16 buckets, OpenMP %d threads" % (n, args.read_len, cores),
    }


def gfa_build(ctx, reads, k):
    """Second half of the metric: wall seconds of the spades-gbuilder path on the same reads (extension
    index -> unitigs + links -> GFA text written to tmpfs), measured once outside the timed region."""
    import tempfile
    ctx.synchronize()
    d = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(d, "bbk_bench_%d.gfa" % os.getpid())
    best = None
    for _ in range(2):  # first pass warms the allocator, second is reported
        t0 = time.perf_counter()
        x = ctx.extindex(reads, k)
        t1 = time.perf_counter()
        u = ctx.unitigs(x)
        t2 = time.perf_counter()
        u.write_gfa(path)
        t3 = time.perf_counter()
        best = {"wall_s": t3 - t0, "extindex_s": t1 - t0, "unitigs_s": t2 - t1, "write_s": t3 - t2,
                "kmers": len(x), "unitigs": len(u), "vertices": u.n_vertices, "links": u.n_links,
                "gfa_bytes": os.path.getsize(path), "output": "GFA1 text on tmpfs"}
        os.unlink(path)
        u.free()
        x.free()
    return best


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import spades_for_blackbird_amd as B
    from spades_for_blackbird_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # BBK_BENCH_FORCE_SHARDED=1: take the N>1 code path (owner partition, RCCL all_to_all, merge, expand) with a
    # single rank -- the only way to exercise it on a one-GPU box
    sharded = world > 1 or os.environ.get("BBK_BENCH_FORCE_SHARDED") == "1"
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctx = B.Context(local_rank, stream=torch.cuda.current_stream())

    k, L = args.k, args.read_len
    nw = B.engine.words(k)
    total_reads = args.reads * world
    genome_len = max(L, total_reads * L // 50)
    reads = ctx.reads_synth(args.reads, read_len=L, genome_len=genome_len, seed_genome=42, seed_reads=43 + rank)
    torch.cuda.synchronize()

    def step():
        """Returns (#records of this rank's part of the result, keep-alive)."""
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

    def drop(x):
        if hasattr(x, "free"):
            x.free()

    for _ in range(args.warmup):
        n_rec, keep = step()
        drop(keep)
        del keep
    ctx.profile(True)
    ctx.profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_rec, keep = step()
        drop(keep)
        del keep
    fence()
    dt = time.perf_counter() - t0
    ctx.profile(False)

    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    nn = torch.tensor([n_rec], dtype=torch.int64, device=dev)
    if sharded:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
    dt_max = float(tt.item())
    distinct_total = int(nn.item())

    if rank == 0:
        prof = {f: ctx.profile_get(f) for f in FAMILIES}
        prof = {f: v for f, v in prof.items() if v["launches"]}
        dom = max(prof, key=lambda f: prof[f]["ms"]) if prof else None
        roof = None
        if dom:
            p = prof[dom]
            ach = p["bytes"] / (p["ms"] * 1e-3) / 1e9
            traffic, traffic_src = None, None
            same = (args.reads, L, k, world) == (PMC_WORKLOAD["reads"], PMC_WORKLOAD["read_len"], PMC_WORKLOAD["k"],
                                                 PMC_WORKLOAD["gpus"])
            if same and os.path.exists(PMC_FILE):
                fam = json.load(open(PMC_FILE)).get("families", {}).get(dom)
                if fam:
                    traffic, traffic_src = fam["hbm_bytes_per_launch"], "profiles/pmc_traffic.json (rocprofv3 --pmc)"
            roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "launches": p["launches"], "avg_launch_ms": p["ms"] / p["launches"],
                    "algorithmic_bytes_per_launch": p["bytes"] / p["launches"]}
        inst_per_gpu = 2 * args.reads * (L - k + 1)
        line = {
            "metric": "distinct k-mers/sec (k=%d, %d bp synthetic reads, count only)" % (k, L),
            "value": distinct_total * 1.0 / (dt_max / args.steps),
            "unit": "distinct k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: synthetic %d x %d bp uniform reads per GPU, k=%d, "
                                   "k-mer count only (both strands, final_kmers order)" % (args.reads, L, k),
                       "reads_per_gpu": args.reads, "read_len": L, "k": k, "genome_len": genome_len,
                       "coverage": 50, "parallelism": "owner-hash shards, %d rank(s)" % world},
            "distinct_kmers": distinct_total,
            "kmer_instances_per_s": inst_per_gpu * world / (dt_max / args.steps),
            "kernel_ms_per_step": {f: v["ms"] / args.steps for f, v in prof.items()},
            "roofline": roof,
        }
        if world == 1 and not args.no_gfa:
            line["gfa_build"] = gfa_build(ctx, reads, k)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ctx, args, B)
        print(json.dumps(line), flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
