#!/bin/bash
# per-kernel durations of rank 0's shard of a W-way split (bench.py --shard-of W), eager launches under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1; OUT=$R/gpurun_out/$T; mkdir -p $OUT
for W in 1 2 4 8; do
  python $R/bench.py --steps 20 --warmup 5 --train-iters 0 --no-cpu-baseline --no-variants --shard-of $W > $OUT/shard$W.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('$OUT/shard$W.json')); print('shard-of $W: %.3f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items() if v})
PY
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr8 -o tr -- python $R/bench.py --no-cpu-baseline --no-variants --no-graph --train-iters 0 --steps 20 --warmup 5 --shard-of 8 > $OUT/tr8.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/tr8/tr_kernel_stats.csv')))
for r in rows[:24]: print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), '%9.1f'%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
