#!/bin/bash
# Host layer (C99) under AddressSanitizer + UndefinedBehaviorSanitizer: the host objects are rebuilt with
# -fsanitize=address,undefined and linked with the ordinary gfx950 kernel objects into
# grtcode_amd/lib/asan/libgrtcode_hip_asan.so; tests pick it up through GRT_LIB_PATH.  Device code is not
# sanitized (GPU ASan is not available on this pool).
#   bash scripts/asan_host.sh build            # here or on the GPU box
#   bash scripts/asan_host.sh test [pytest args]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/grtcode_amd/lib/asan
ASAN=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
case "$1" in
build)
  python -m grtcode_amd.build > /dev/null
  mkdir -p $OUT/obj
  for f in grt_error grt_util grt_grid grt_device grt_optics grt_tips grt_gas_optics grt_solvers grt_pipeline grt_multi; do
    gcc -std=gnu99 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fPIC -D__HIP_PLATFORM_AMD__ \
        -I$ROOT/include -I/opt/rocm/include -I$ROOT/grtcode_amd/csrc/host -c $ROOT/grtcode_amd/csrc/host/$f.c -o $OUT/obj/$f.o
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o $OUT/libgrtcode_hip_asan.so $OUT/obj/*.o \
        $ROOT/grtcode_amd/lib/obj/k_*.o -L/opt/rocm/lib -lamdhip64 -lm -ldl -Wl,-rpath,/opt/rocm/lib
  echo $OUT/libgrtcode_hip_asan.so ;;
test)
  shift
  LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:protect_shadow_gap=0 UBSAN_OPTIONS=print_stacktrace=1 \
  GRT_LIB_PATH=$OUT/libgrtcode_hip_asan.so python -m pytest -q -p no:cacheprovider "$@" ;;
*) echo "usage: $0 build | test [pytest args]"; exit 2 ;;
esac
