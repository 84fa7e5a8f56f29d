#!/bin/bash
# GPU box: the bench at the per-rank batch shapes an 8-GPU run of RFMIP-like column sets has (task 4):
# N columns per step in ONE launch (N <= 64) or in launches of 64 -- 13 = ceil(100/8), 225 = 1 800/8.
#   bash scripts/batch_sweep.sh <out.json> [extra bench arguments]
OUT=${1:-gpurun_out/batch_sweep.json}; shift
cd "$(dirname "$0")/.."
echo "[" > $OUT
first=1
for n in 1 2 4 8 9 13 16 32 64 225; do
    chunk=$n; [ $n -gt 64 ] && chunk=64
    steps=$(( 640 / n )); [ $steps -lt 6 ] && steps=6; [ $steps -gt 48 ] && steps=48
    line=$(python3 bench.py --cols $n --chunk $chunk --steps $steps --warmup 2 --no-extras --no-cpu-baseline "$@" 2>/dev/null | tail -1)
    [ -z "$line" ] && { echo "bench failed at --cols $n" >&2; exit 1; }
    [ $first -eq 0 ] && echo "," >> $OUT
    first=0
    echo "$line" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernel_ms_per_step']
print(json.dumps({'cols': $n, 'chunk': $chunk, 'steps': d['steps'], 'columns_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'ms_per_column': d['ms_per_step']/$n,
                  'kernel_ms_per_step': k, 'parity_ok': (d.get('parity') or {}).get('ok')}))" >> $OUT
    tail -1 $OUT | cut -c1-150
done
echo "]" >> $OUT
