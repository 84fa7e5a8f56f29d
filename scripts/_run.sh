#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
mkdir -p gpurun_out/r2fine; rm -f gpurun_out/r2fine/fine_grids.jsonl
python bench.py > gpurun_out/r2fine/bench_line.json 2> gpurun_out/r2fine/bench.err || exit 1
python bench.py --lw-dw 0.001 --cols 2 --steps 3 --no-cpu-baseline --no-extras > gpurun_out/r2fine/g3_pipeline.json 2>/dev/null || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2fine/g3trace -- python3 scripts/fine_grid.py --dw 0.001 --reps 3 > gpurun_out/r2fine/g3.json 2> gpurun_out/r2fine/g3.err || exit 1
for DW in 0.1 0.05 0.01 0.005 0.0025 0.001; do
  timeout -k 10 400 python scripts/fine_grid.py --dw $DW --reps 3 --compare 2 2>/dev/null | tail -1 >> gpurun_out/r2fine/fine_grids.jsonl
done
echo done
