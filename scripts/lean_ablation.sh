#!/bin/bash
# Timing experiments on the lean line loop (results are WRONG by construction: parts of the loop are compiled out).
#   local:   bash scripts/lean_ablation.sh build        -> grtcode_amd/lib/variants/libgrt_<name>.so
#   GPU box: bash scripts/lean_ablation.sh run          -> kernel times of the default bench per variant
set -e
cd "$(dirname "$0")/.."
V=grtcode_amd/lib/variants
NAMES="base NOEVAL NORAW NOSLOTS NOREDUCE NOLDSADD NOMOM"
if [ "$1" = "build" ]; then
    mkdir -p $V
    python -m grtcode_amd.build > /dev/null
    for n in $NAMES; do
        D=""; [ $n != base ] && D="-DGRT_ABL_$n"
        hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -munsafe-fp-atomics -fno-slp-vectorize -Iinclude $D $EXTRA \
              -c grtcode_amd/csrc/hip/k_gas_optics_mp.hip -o $V/mp_$n.o
        OBJS=$(ls grtcode_amd/lib/obj/*.o | grep -v k_gas_optics_mp.o | grep -v grt_clouds.o)
        hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libgrt_$n.so $OBJS $V/mp_$n.o -L/opt/rocm/lib -lamdhip64 -lm -ldl -Wl,-rpath,/opt/rocm/lib
        rm $V/mp_$n.o
        echo built $n
    done
else
    for n in $NAMES; do
        GRT_LIB_PATH=$PWD/$V/libgrt_$n.so python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --cols 64 --chunk 64 2>/dev/null \
            | python3 -c "import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_ms_per_step'];print('$n', round(d['value'],1), 'lw', round(k['gas_optics_lw'],2), 'sw', round(k['gas_optics_sw'],2))"
    done
fi
