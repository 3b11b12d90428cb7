#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 5 --train-iters 0 --no-cpu-baseline --no-variants"
for W in 8 1; do
  rocprofv3 --kernel-trace --output-format csv -d $O/tr$W -o tr -- $B --shard-of $W --min-time 0.2 > $O/tr$W.log 2>&1
  echo "== trace shard-of $W"; python $R/tools/trace_overlap.py $(find $O/tr$W -name '*kernel_trace.csv' | head -1)
  find $O/tr$W -name '*.csv' -delete
done > $O/out.txt 2>&1
cat $O/out.txt
