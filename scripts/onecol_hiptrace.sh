#!/bin/bash
# One-column reference ABI under rocprofv3's HIP-API, kernel and copy traces (no counters): where the host waits.
#   bash scripts/onecol_hiptrace.sh   -> gpurun_out/onecol_hip/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
rm -rf gpurun_out/onecol_hip; mkdir -p gpurun_out/onecol_hip
python3 scripts/one_column_abi_timing.py --cols 60 --free > gpurun_out/onecol_hip/free.json 2> gpurun_out/onecol_hip/free.err || exit 1
python3 scripts/one_column_abi_timing.py --cols 60 > gpurun_out/onecol_hip/synced.json 2> /dev/null || exit 1
timeout -k 10 400 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/onecol_hip -- python3 scripts/one_column_abi_timing.py --cols 40 --free > gpurun_out/onecol_hip/out.json 2> gpurun_out/onecol_hip/err.txt
echo rc=$?
