#!/bin/bash
# Run on the GPU box (via gpurun): the vector-instruction MIX of the line kernels by class (PMC SQ_INSTS_VALU_*),
# and what an instruction of each class costs the vector pipe (scripts/valu_mix.hip).
# Usage: bash scripts/profile_mix.sh r4   -> gpurun_out/mix_<tag>/{valu_mix.txt,mix.json}
set -o pipefail
TAG=${1:-r4}
OUT=gpurun_out/mix_$TAG
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 scripts/valu_mix.hip -o /tmp/valu_mix 2> /dev/null && /tmp/valu_mix > $OUT/valu_mix.txt || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-extras --cols 16 --chunk 16"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 \
    --output-format csv -d $OUT/pmc_a -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_a.err || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM \
    --output-format csv -d $OUT/pmc_b -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_b.err || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d $OUT/pmc_c -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_c.err || exit 1
python3 - <<PY
import csv, glob, collections, json
out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in ("pmc_a", "pmc_b", "pmc_c"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gas_optics" not in k:
                continue
            key = k.split("(")[0][-60:] + "@" + r["Grid_Size"]
            out[key][r["Counter_Name"]][0] += float(r["Counter_Value"])
            out[key][r["Counter_Name"]][1] += 1
res = {k: {c: v[0]/v[1] for c, v in cs.items()} for k, cs in out.items()}
json.dump(res, open("$OUT/mix.json", "w"), indent=1, sort_keys=True)
for k, cs in res.items():
    tot = cs.get("SQ_INSTS_VALU", 0.0)
    if tot <= 0:
        continue
    print(k)
    for c in sorted(cs):
        print("   %-28s %14.4g  %6.3f of VALU" % (c, cs[c], cs[c]/tot))
PY
echo done
