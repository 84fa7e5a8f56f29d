# scratch: ablation variants of the line kernel on the ERA5-like grids (exp_*.so built by hand)
for v in base nomom noprepass noring none; do
  L=grtcode_amd/lib/exp_$v.so; [ $v = base ] && L=grtcode_amd/lib/libgrtcode_hip.so
  GRT_LIB_PATH=$L timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cols 4 --lw-dw 0.1 --sw-dw 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), {k: round(x,2) for k,x in d['kernel_ms_per_step'].items()})"
done
