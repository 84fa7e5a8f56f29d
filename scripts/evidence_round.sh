#!/bin/bash
# GPU box (via gpurun): retake EVERY figure tests/test_bench_contract.py and DESIGN.md quote, on ONE build, in one session
# (task 2: round 4 kept lines of two different builds side by side).  Output: gpurun_out/evidence_<tag>/ ;
# then, locally:  python scripts/collect_evidence.py <tag>   -> profiles/<tag>_*
#   bash scripts/evidence_round.sh r5 [part ...]      parts: bench profile pair lines sweep g3 driver tests soak (default: all)
set -o pipefail
TAG=${1:-r5}; shift
PARTS="${*:-bench profile pair lines sweep g3 driver tests soak}"
OUT=gpurun_out/evidence_$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.." || exit 1
export PYTHONPATH=$PWD
sha256sum grtcode_amd/lib/libgrtcode_hip.so grtcode_amd/csrc/hip/*.hip grtcode_amd/csrc/hip/*.h > $OUT/build_sha256.txt
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
stamp() { echo "[evidence] $(date +%H:%M:%S) $*"; }

if has bench; then
    stamp "the headline: python bench.py (defaults)"
    python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err || { echo "bench failed"; tail -5 $OUT/bench_line.err; exit 1; }
fi
if has pair; then
    # RCCL at world size 1 against the plain run, BACK TO BACK on this build, three times over, in one file
    stamp "forced-RCCL / plain pairs"
    echo "[" > $OUT/rccl_world1_pairs.json
    for i in 1 2 3; do
        p=$(python3 bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1)
        f=$(GRT_BENCH_FORCE_DIST=1 python3 bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1)
        [ -z "$p" -o -z "$f" ] && { echo "pair $i failed"; exit 1; }
        [ $i -gt 1 ] && echo "," >> $OUT/rccl_world1_pairs.json
        echo "{\"plain\": $p, \"forced\": $f}" >> $OUT/rccl_world1_pairs.json
    done
    echo "]" >> $OUT/rccl_world1_pairs.json
fi
if has lines; then
    stamp "the other configurations' lines"
    python3 bench.py --columns 100 --no-extras --no-cpu-baseline > $OUT/strong_100_columns_bench_line.json 2>/dev/null || exit 1
    python3 bench.py --lw-dw 0.1 --sw-dw 10 --cols 64 --no-extras --no-cpu-baseline > $OUT/era5_like_bench_line.json 2>/dev/null || exit 1
    python3 bench.py --lanes 2 --cols 128 --chunk 64 --no-extras --no-cpu-baseline > $OUT/two_streams_bench_line.json 2>/dev/null || exit 1
fi
if has g3; then
    stamp "G3 (0.001 cm-1 longwave) through the whole pipeline; 32 columns per step in column groups"
    python3 bench.py --lw-dw 0.001 --cols 2 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $OUT/g3_pipeline_bench_line.json 2>/dev/null || exit 1
    python3 bench.py --lw-dw 0.001 --cols 32 --chunk 32 --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $OUT/g3_pipeline_32_columns_bench_line.json 2> $OUT/g3_32.err || echo "32-column G3 line failed (see g3_32.err)"
fi
if has sweep; then
    stamp "batch shapes 1 .. 225 columns per step"
    bash scripts/batch_sweep.sh $OUT/batch_sweep.json > $OUT/batch_sweep.log 2>&1 || exit 1
fi
if has driver; then
    stamp "the unchanged reference driver binary, the one-column ABI"
    python3 scripts/time_reference_driver.py --api-timing --out $OUT/reference_driver_timing.json > $OUT/driver.log 2>&1 || echo "driver timing failed (see driver.log)"
    python3 scripts/one_column_abi_timing.py --cols 100 --free --out $OUT/one_column_abi_unsynchronised.json > /dev/null 2>&1 || echo "one-column timing failed"
fi
if has tests; then
    stamp "GPU suite, plain and deterministic"
    timeout -k 10 900 python3 -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee -a $OUT/pytest_gpu.log
    GRT_DETERMINISTIC=1 timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $OUT/pytest_gpu_deterministic.log 2>&1; echo "deterministic rc=$?" | tee -a $OUT/pytest_gpu_deterministic.log
fi
if has soak; then
    stamp "1 200 randomised parity cases"
    GRT_STRESS_SEEDS=600 GRT_STRESS_WIDE=1 timeout -k 10 400 python3 -m pytest tests/test_gpu_moment_kernel.py -q -m gpu -k randomised > $OUT/soak.log 2>&1; echo "wide=1 rc=$?" >> $OUT/soak.log
    GRT_STRESS_SEEDS=600 GRT_STRESS_WIDE=2 timeout -k 10 400 python3 -m pytest tests/test_gpu_moment_kernel.py -q -m gpu -k randomised >> $OUT/soak.log 2>&1; echo "wide=2 rc=$?" >> $OUT/soak.log
    tail -3 $OUT/soak.log
fi
if has profile; then
    stamp "rocprofv3: kernel trace + counters of the default bench command; the G3 column"
    bash scripts/profile_round.sh $TAG > $OUT/profile_round.log 2>&1 || { echo "profile_round failed"; tail -5 $OUT/profile_round.log; }
    bash scripts/profile_g3.sh $TAG > $OUT/profile_g3.log 2>&1 || { echo "profile_g3 failed"; tail -5 $OUT/profile_g3.log; }
fi
stamp done
