#!/bin/bash
# HBM-side traffic and L2 atomics of the 0.001 cm-1 column's kernels (separate --pmc passes).
#   bash scripts/profile_g3_traffic.sh r3   -> gpurun_out/prof_g3t_<tag>/...
set -o pipefail
TAG=${1:-r3}
OUT=gpurun_out/prof_g3t_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
for c in FETCH_SIZE WRITE_SIZE "TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"; do
    d=$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --output-format csv -d $OUT/$d -- python3 scripts/fine_grid.py --dw 0.001 --reps 1 > /dev/null 2> $OUT/$d.err || echo "pass $d failed"
done
echo done
