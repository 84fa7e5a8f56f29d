#!/bin/bash
# GPU box: time prebuilt variants (grtcode_amd/lib/variants/libgrt_<name>.so); extra bench args after "--"
cd "$(dirname "$0")/.."
V=grtcode_amd/lib/variants
names=(); extra=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; extra=("$@"); break; fi; names+=("$1"); shift; done
for n in "${names[@]}"; do
    GRT_LIB_PATH=$PWD/$V/libgrt_$n.so python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --cols 32 --chunk 32 "${extra[@]}" 2>/dev/null \
        | python3 -c "import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_ms_per_step'];print('$n ${extra[*]}', round(d['value'],1), 'lw', round(k['gas_optics_lw'],2), 'sw', round(k['gas_optics_sw'],2))"
done
