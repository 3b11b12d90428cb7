#!/bin/bash
# per-kernel durations of the eval frame (eager launches) for rank 0's shard of a W-way split:  bash tools/kt_eval.sh <tag> W...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1; shift; OUT=$R/gpurun_out/$T; mkdir -p $OUT
for W in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$W -o tr -- python $R/bench.py --no-cpu-baseline --no-variants --no-graph --train-iters 0 --steps 20 --warmup 5 --shard-of $W > $OUT/kt$W.log 2>&1
  echo "== shard-of $W"
  python - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/kt$W/tr_kernel_stats.csv')))
for r in rows[:20]: print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), '%9.1f'%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
done
