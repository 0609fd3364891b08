for k in 21 55; do
for e in "" "BBK_NO_DIST=1"; do
  echo "k=$k $e"
  env $e timeout -k 10 300 python bench.py --k $k --steps 2 --warmup 1 --no-cpu-baseline --no-gfa 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:round(v,2) for k,v in d['kernel_ms_per_step'].items() if k in ('lds_sort','lds_dedup','compact')})"
done; done
