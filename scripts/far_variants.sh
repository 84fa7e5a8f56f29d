#!/bin/bash
# Timing experiments on the far-field gather: variants of k_gas_optics_far.hip under other flags (-D macros put into the
# source for the experiment and taken out again: profiles/r5_sw_tail_steps.txt lists the ones of round 5 and what they showed).
#   local:   bash scripts/far_variants.sh build "name:flags" ...     GPU box: bash scripts/far_variants.sh run name ... [-- bench args]
set -e
cd "$(dirname "$0")/.."
V=grtcode_amd/lib/variants
if [ "$1" = "build" ]; then
    shift
    mkdir -p $V
    python -m grtcode_amd.build > /dev/null
    for spec in "$@"; do
        n=${spec%%:*}; D=${spec#*:}
        hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -munsafe-fp-atomics -fno-slp-vectorize -Iinclude $D \
              -c grtcode_amd/csrc/hip/k_gas_optics_far.hip -o $V/far_$n.o 2> /dev/null
        OBJS=$(ls grtcode_amd/lib/obj/*.o | grep -v k_gas_optics_far.o | grep -v grt_clouds.o)
        hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libgrt_$n.so $OBJS $V/far_$n.o -L/opt/rocm/lib -lamdhip64 -lm -ldl -Wl,-rpath,/opt/rocm/lib
        rm $V/far_$n.o
        echo built $n "($D)"
    done
else
    shift
    names=(); extra=()
    while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; extra=("$@"); break; fi; names+=("$1"); shift; done
    for n in "${names[@]}"; do
        GRT_LIB_PATH=$PWD/$V/libgrt_$n.so python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras "${extra[@]}" 2>/dev/null \
            | python3 -c "import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_ms_per_step'];print('$n ${extra[*]}', round(d['value'],1), 'far lw', round(k['far_field_lw'],3), 'far sw', round(k['far_field_sw'],3), 'solver sw', round(k['sw_solver'],3))" || echo "$n failed"
    done
fi
