# scratch: instruction-cache counters of the line kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_ic -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_ic.err || { tail -5 gpurun_out/pmc_ic.err; exit 1; }
python3 - <<PY
import csv, glob, collections
f=sorted(glob.glob('gpurun_out/pmc_ic/*/*counter_collection.csv'))[-1]
a=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'gas_optics' in r['Kernel_Name']:
        a[(r['Kernel_Name'][22:50], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for g,c in sorted(a.items()):
    print(g, {k: round(sum(v)/len(v)/1e6,2) for k,v in c.items()})
PY
