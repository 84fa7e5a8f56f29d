# scratch: dynamic VALU instruction counts of ablation variants (exp_*.so built by hand)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in base nomom noprepass nodrain noring nofar none; do
  L=grtcode_amd/lib/exp_$v.so; [ $v = base ] && L=grtcode_amd/lib/libgrtcode_hip.so
  export GRT_LIB_PATH=$GRAFT_REPO_ROOT/$L
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmcv_$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmcv_$v.err || exit 1
  python3 - <<PY
import csv, glob, collections
f=sorted(glob.glob('gpurun_out/pmcv_$v/*/*counter_collection.csv'))[-1]
a=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'gas_optics' in r['Kernel_Name']:
        a[r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for g,c in sorted(a.items(), key=lambda x:int(x[0])):
    print('$v', g, {k: round(sum(v)/len(v)/1e9,3) for k,v in c.items()})
PY
done
