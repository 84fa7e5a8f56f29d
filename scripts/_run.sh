export PYTHONPATH=.
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
for DW in 0.001 0.0025 0.01 0.1; do
  timeout -k 10 300 python scripts/fine_grid.py --dw $DW --reps 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dw=$DW', d['seconds_per_column'], d['kernel_ms'], d['ran']['tile'], d['ran']['moments'], d['ran']['halo'])"
done
