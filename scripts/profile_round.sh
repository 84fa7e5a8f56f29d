#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + HBM traffic counters of the default bench command.
# Usage: bash scripts/profile_round.sh r2   -> gpurun_out/prof_<tag>/...
set -o pipefail
TAG=${1:-r1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_under_trace.json 2> $OUT/trace.err || exit 1
# HBM traffic: FETCH_SIZE and WRITE_SIZE need separate passes on gfx950 (TCC slot budget)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.err || exit 1
# READ bytes exactly: the L2's memory-side read requests by size (32, 64, 128 bytes).  FETCH_SIZE tallies a 128-byte request
# at 64 bytes on gfx950, so it has to be doubled for vector loads and NOT for scalar loads (scripts/fetch_calibration.hip,
# profiles/r4_fetch_calibration.json); these counters need no such knowledge of the access pattern
rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/pmc_rdreq -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_rdreq.err || exit 1
echo done
