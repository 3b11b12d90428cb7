#!/bin/bash
# quick A/B of bench.py --shard-of W under an environment setting:  bash tools/shard_quick.sh "<env assignments>" W...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; ENVS=$1; shift
for W in "$@"; do
  env $ENVS python $R/bench.py --steps 20 --warmup 5 --train-iters 0 --no-cpu-baseline --no-variants --shard-of $W 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$ENVS] shard-of $W: %.3f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items() if v})"
done
