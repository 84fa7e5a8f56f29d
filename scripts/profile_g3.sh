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
echo done
