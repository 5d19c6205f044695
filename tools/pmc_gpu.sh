#!/bin/bash
# one PMC pass: tools/pmc_gpu.sh <tag> "<counters>" [bench args]
TAG=$1; CNT=$2; shift 2
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $R/$OUT -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/$OUT/bench.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(float)
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('outfit::','')
        if k.startswith('__amd'): continue
        agg[(k,r['Counter_Name'])]+=float(r['Counter_Value'])
ks=sorted({k for k,_ in agg}); cs=sorted({c for _,c in agg})
print('%-24s'%'kernel'+''.join('%18s'%c[:17] for c in cs))
for k in ks: print('%-24s'%k+''.join('%18.4g'%agg.get((k,c),float('nan')) for c in cs))
PY
