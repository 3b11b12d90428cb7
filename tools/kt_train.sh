#!/bin/bash
# rocprofv3 kernel trace + stats of the training iteration (tools/train_bench.py); usage: bash tools/kt_train.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
python $R/tools/train_bench.py --iters 40 > $OUT/train_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -o tr -- python $R/tools/train_bench.py --iters 40 > $OUT/tr.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/tr/tr_kernel_stats.csv')))
print(open('$OUT/train_bench.log').read()[-600:])
for r in rows[:32]: print(r['Name'][:64].ljust(64), r['Calls'].rjust(6), '%9.1f'%(float(r['AverageNs'])/1e3), '%9.1f'%(float(r['MaxNs'])/1e3), r['Percentage'])
PY
find $OUT/tr -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
