#!/bin/bash
# HBM-side traffic counters only (two passes: FETCH_SIZE and WRITE_SIZE cannot share one on gfx950)
TAG=${1:-t}
OUT=gpurun_out/traffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err || exit 1
python3 - <<PY
import csv, glob, collections
for d,c in (('pmc_fetch','FETCH_SIZE'),('pmc_write','WRITE_SIZE')):
    f=glob.glob('$OUT/%s/*/*counter_collection.csv'%d)[0]
    a=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']==c and 'gas_optics' in r['Kernel_Name']:
            a[r['Grid_Size']][0]+=float(r['Counter_Value']); a[r['Grid_Size']][1]+=1
    print(c, {k:(round(v[0]/v[1]/1e6,3),'M KB/launch',v[1]) for k,v in a.items()})
PY
python3 -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['kernel_ms_per_step'])"
