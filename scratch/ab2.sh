for v in "" _t16k _t16k512 _t4k; do
  echo "variant=$v"
  BBK_LIB=$PWD/spades_for_blackbird_amd/libbbk$v.so timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gfa 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms_per_step'].items() if k.startswith('part_')})"
done
