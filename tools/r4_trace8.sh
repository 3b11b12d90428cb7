#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-r4trace8}
OUT=$R/gpurun_out/$T; mkdir -p $OUT
B="python $R/bench.py --no-cpu-baseline --no-variants --no-graph --train-iters 0 --shard-of 8"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $B --steps 10 --warmup 3 > $OUT/trace.log 2>&1
find $OUT -name "*_agent_info.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
python - <<PY
import csv,glob
f=glob.glob('$OUT/trace/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:40], r['Calls'], r['AverageNs'], r['MinNs'])
PY
