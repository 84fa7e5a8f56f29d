#!/bin/bash
# The 0.001 cm-1 longwave column (n = 3 249 001) under rocprofv3: kernel stats, then SQ counters in a pass of their own.
#   bash scripts/profile_g3.sh r3   -> gpurun_out/prof_g3_<tag>/...   (then: python scripts/summarize_g3.py r3)
set -o pipefail
TAG=${1:-r3}
OUT=gpurun_out/prof_g3_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/fine_grid.py --dw 0.001 --reps 3 > $OUT/fine_grid.json 2> $OUT/trace.err || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -- python3 scripts/fine_grid.py --dw 0.001 --reps 2 > /dev/null 2> $OUT/pmc_sq.err || exit 1
# HBM-side traffic and where the atomics are carried out (separate passes: the TCC counters share few slots)
# (reads exactly: the L2's read requests by size -- FETCH_SIZE is half the bytes of vector loads and all the bytes of scalar loads)
for c in FETCH_SIZE WRITE_SIZE "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
    d=pmc_$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --output-format csv -d $OUT/$d -- python3 scripts/fine_grid.py --dw 0.001 --reps 1 > /dev/null 2> $OUT/$d.err || echo "pass $d failed"
done
echo done
