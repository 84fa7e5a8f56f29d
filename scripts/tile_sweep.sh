#!/bin/bash
# GPU box: per-band tilings of the line kernel at the bench's launch size (exploration)
cd "$(dirname "$0")/.."
run() { python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c "import json,sys,os;d=json.loads(sys.stdin.read());k=d['kernel_ms_per_step'];print(os.environ.get('TAG',''), round(d['value'],1), 'lw', round(k['gas_optics_lw'],2), 'sw', round(k['gas_optics_sw'],2))"; }
for t in 64 128 256; do for ns in 1 2; do TAG="lw tile $t nslice $ns" GRT_BENCH_LW_TILE=$t run --lw-nslice $ns; done; done
for t in 128 256; do for ns in 1 2; do TAG="sw tile $t nslice $ns" GRT_BENCH_SW_TILE=$t GRT_BENCH_SW_NSLICE=$ns run; done; done
