#!/bin/bash
# frames in flight on a 1/8 shard: does a second stream overlap at all?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4b; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 5 --train-iters 0 --no-cpu-baseline --no-variants --shard-of 8"
one() { echo "== $*"; env "${@:2}" $B --in-flight $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.4f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items() if v})"; }
{
one 1 A=1; one 2 A=1; one 3 A=1; one 2 GPU_MAX_HW_QUEUES=8; one 4 GPU_MAX_HW_QUEUES=8
for F in 1 2; do
  rocprofv3 --kernel-trace --output-format csv -d $O/tr$F -o tr -- $B --in-flight $F --min-time 0.2 > $O/tr$F.log 2>&1
  echo "== trace in-flight $F"; python $R/tools/trace_overlap.py $(find $O/tr$F -name '*kernel_trace.csv' | head -1)
  find $O/tr$F -name '*.csv' -delete
done
} > $O/out.txt 2>&1
cat $O/out.txt
