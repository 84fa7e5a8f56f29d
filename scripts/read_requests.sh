#!/bin/bash
# GPU box: HBM-side READ bytes of a command's kernels from the L2's request counters by size (32, 64, 128 bytes) --
# exact, where FETCH_SIZE needs a pattern-dependent correction on gfx950 (scripts/fetch_calibration.hip).
#   bash scripts/read_requests.sh <tag> <program> [args...]     -> gpurun_out/rdreq_<tag>/rdreq.json
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/rdreq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc -- "$@" > $OUT/run.log 2> $OUT/pmc.err || { tail -5 $OUT/pmc.err; exit 1; }
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-70:] + "@" + r["Grid_Size"]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    m = {c: sum(v)/len(v) for c, v in cs.items()}
    n32, n64, n128 = (m.get("TCC_EA0_RDREQ_%s_sum" % s, 0.0) for s in ("32B", "64B", "128B"))
    nall = m.get("TCC_EA0_RDREQ_sum", 0.0)
    res[k] = {"requests_32B": n32, "requests_64B": n64, "requests_128B": n128, "requests_all": nall,
              "read_bytes": 32*n32 + 64*n64 + 128*n128, "launches": len(next(iter(cs.values())))}
json.dump(res, open("$OUT/rdreq.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["read_bytes"])[:12]:
    print("%-75s %8.3f GB  (32B %.3g, 64B %.3g, 128B %.3g, all %.3g)" % (k, v["read_bytes"]/1e9, v["requests_32B"], v["requests_64B"], v["requests_128B"], v["requests_all"]))
PY
