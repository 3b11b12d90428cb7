#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r4h}; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 5 --train-iters 0 --no-cpu-baseline --no-variants"
one() { echo "== $*"; $B "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.4f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items() if v})"; }
{
one --shard-of 8; one --shard-of 4; one --shard-of 2; one
for W in 8; do
  rocprofv3 --kernel-trace --output-format csv -d $O/tr$W -o tr -- $B --shard-of $W --min-time 0.2 > $O/tr$W.log 2>&1
  echo "== trace shard-of $W"; python $R/tools/trace_overlap.py $(find $O/tr$W -name '*kernel_trace.csv' | head -1) | grep -A30 timeline
  find $O/tr$W -name '*.csv' -delete
done
if [ -n "$TESTS" ]; then cd $R && timeout 1200 python -m pytest $TESTS -x -q -m gpu 2>&1 | tail -5; fi
} > $O/out.txt 2>&1
cat $O/out.txt
