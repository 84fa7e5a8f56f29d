#!/bin/bash
# GPU box: FETCH_SIZE against known byte counts in the tree gather's access patterns (scripts/fetch_calibration.hip)
set -o pipefail
OUT=gpurun_out/fetch_cal
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 scripts/fetch_calibration.hip -o /tmp/fetch_cal 2> $OUT/build.err || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -- /tmp/fetch_cal > $OUT/run.log 2> $OUT/pmc.err || exit 1
cat $OUT/run.log
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and r["Kernel_Name"].startswith("cal_"):
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
cells = 32*1024*1024
asked = {"cal_stream16": cells*48, "cal_lane48": cells*48, "cal_lane16of48": cells*16, "cal_scalar48": cells*48}
res = {}
for k, v in sorted(acc.items()):
    kib = sum(v)/len(v)
    res[k] = {"FETCH_SIZE_KiB": kib, "bytes_asked_for": asked[k], "bytes_touched_in_128B_lines": cells*48 if k != "cal_lane16of48" else None,
              "FETCH_SIZE_bytes_over_asked": kib*1024/asked[k], "launches": len(v)}
    print(k, res[k])
json.dump(res, open("$OUT/fetch_calibration.json", "w"), indent=1)
PY
