"""Per-kernel SQ counter ratios from a rocprofv3 --pmc counter_collection.csv (waves waiting / issuing / LDS busy):
   rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
       SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d out -o pmc -- python3 tools/count_perf.py 55
   python tools/pmc_sq.py out/.../pmc_counter_collection.csv"""
import csv, sys, re, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::", "", r.get("Kernel_Name", ""))
    name = re.sub(r"\(.*", "", name)[:60]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(name, r["Counter_Name"])] += 1
for name, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
    n = max(1, cnt[(name, "SQ_WAVE_CYCLES")])
    wc = d.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"{name:58s} disp={n:4d} " + " ".join(f"{k[3:]}={v/wc:.3f}" for k, v in sorted(d.items()) if k != "SQ_WAVE_CYCLES") + f" WAVE_CYCLES/disp={wc/n:.3e}")
