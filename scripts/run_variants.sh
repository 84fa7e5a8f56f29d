# quick A/B of the two arithmetic forms of the line kernel (used during tuning)
for f in 1 0; do
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cols 4 --fast $f --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fast=$f', round(d['value'],2), {k: round(x,2) for k,x in d['kernel_ms_per_step'].items() if 'gas' in k})"
done
