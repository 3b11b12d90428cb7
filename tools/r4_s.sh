#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 5 --train-iters 0 --no-cpu-baseline --no-variants"
{ for K in 1 2 4 5 10; do for W in 1 8; do echo -n "[in-flight $K shard-of $W] "; $B --shard-of $W --in-flight $K 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.4f ms' % d['ms_per_step'], d['config']['frames_in_flight'])"; done; done; } > $O/out.txt 2>&1
cat $O/out.txt
