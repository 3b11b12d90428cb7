#!/bin/bash
# A/B of two builds of the library on the SAME box, alternating:  bash tools/ab.sh <libA> <libB> <rounds> [bench args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; A=$1; B=$2; N=$3; shift 3
for i in $(seq $N); do
  for L in $A $B; do
    INVR_LIB_PATH=$R/$L python $R/bench.py --steps 20 --warmup 5 --train-iters 0 --no-cpu-baseline --no-variants "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L %.4f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items() if v})"
  done
done
