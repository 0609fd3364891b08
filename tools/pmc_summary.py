#!/usr/bin/env python3
"""Summarises two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) into per-kernel HBM traffic.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> out.json

Units and gfx950 correction (MI355X_MICROARCH.md, HBM section): both counters are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read on gfx950, so
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  (The guide calibrates the factor for 16 B/lane
streams; the kernels here read 8/16 B per lane.  The doubled figure reproduces the algorithmic read
bytes of the streaming kernels exactly, e.g. 10.43 GB for 1.3 G 8-byte keys.)
"""
import collections
import csv
import json
import re
import sys

FAMILY = [  # rocprof kernel symbol -> the name bbk_ctx_profile_get / bench.py use for the same kernel
    # k_part_reads<W, HAS_VAL, HIST_ONLY>, k_part<W, HAS_VAL, HIST_ONLY, LVL1> (LVL1 also serves the level-0 pass)
    (r"k_sk_part1", "k_sk_part1"),
    (r"k_sk_part2<\d, true>", "k_sk_part2_hist"),
    (r"k_sk_part2<\d, false>", "k_sk_part2"),
    (r"k_sk_dedup<.*SkdB>", "k_sk_dedup_B"),
    (r"k_sk_dedup", "k_sk_dedup"),
    (r"k_part_reads_narrow", "k_part_reads_narrow"),
    (r"k_part_narrow2", "k_part_narrow2"),
    (r"k_bucket_hash32", "k_bucket_hash32"),
    (r"k_part_reads<\d, (true|false), false>", "k_part_reads"),
    (r"k_part_reads<\d, (true|false), true>", "k_part_reads_hist"),
    (r"k_part<\d, (true|false), false, true>", "k_part_l1"),
    (r"k_part<\d, (true|false), false, false>", "k_part_l2"),
    (r"k_part<\d, (true|false), true, true>", "k_part_hist1"),
    (r"k_part<\d, (true|false), true, false>", "k_part_hist2"),
    (r"k_bucket_hashidx", "k_bucket_hashidx"),
    (r"k_bucket_hash", "k_bucket_hash"),
    (r"k_bucket_dist<", "k_bucket_dist"),
    (r"k_bucket<", "k_bucket"),
    (r"k_compact", "compact"),
    (r"k_scatter<", "scatter"),
    (r"k_hist<", "hist"),
    (r"k_expand_rc", "expand"),
    (r"k_extract", "extract"),
]


def family(name):
    for pat, fam in FAMILY:
        if re.search(pat, name):
            return fam
    return None


def load(path, counter):
    out = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            out[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return out


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    fams = collections.defaultdict(lambda: {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0, "kernels": set()})
    for name in set(fetch) | set(write):
        fam = family(name)
        if not fam:
            continue
        f, w = fetch.get(name, []), write.get(name, [])
        d = fams[fam]
        d["launches"] += max(len(f), len(w))
        d["fetch_kib"] += sum(f)
        d["write_kib"] += sum(w)
        d["kernels"].add(name.split("(")[0])
    res = {}
    for fam, d in fams.items():
        hbm = (2.0 * d["fetch_kib"] + d["write_kib"]) * 1024.0
        res[fam] = {"launches": d["launches"], "FETCH_SIZE_KiB": d["fetch_kib"], "WRITE_SIZE_KiB": d["write_kib"],
                    "hbm_bytes_total": hbm, "hbm_bytes_per_launch": hbm / max(1, d["launches"]),
                    "kernels": sorted(d["kernels"])}
    json.dump({"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py "
                       "--steps 1 --warmup 0 --no-cpu-baseline; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
               "families": res}, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for fam, r in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_total"]):
        print("%-22s launches=%2d  %.2f GB/launch" % (fam, r["launches"], r["hbm_bytes_per_launch"] / 1e9))


if __name__ == "__main__":
    main()
