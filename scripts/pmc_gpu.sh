#!/bin/bash
# usage (on the GPU box): scripts/pmc_gpu.sh <tag> "<counters>" [bench args]   -> gpurun_out/<tag>/
set -o pipefail
TAG=$1; CNT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/run.log 2>&1 || { tail -20 $OUT/run.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    agg[row['Kernel_Name'][:50]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in agg.items():
    if 'sbm_sens' in k:
        print(k, row['VGPR_Count'] if False else '')
        for c, v in sorted(d.items()):
            print("   %-28s launches=%d mean=%.4g" % (c, len(v), sum(v)/len(v)))
PY
