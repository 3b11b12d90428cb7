#!/bin/bash
# gpurun -- 'bash tools/ab_env.sh <tag> "<ENV1=..>" "<ENV2=..>" ...' : the headline bench under different environment settings, alternating, on one box
# ("" = the default environment)
cd $GRAFT_REPO_ROOT; T=$1; shift; mkdir -p gpurun_out/$T
for i in 1 2; do for E in "$@"; do
  env $E timeout 300 python bench.py --steps 20 --warmup 5 --train-iters 0 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$E] %.4f ms' % d['ms_per_step'], d['replay_bit_exact'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items() if v})"
done; done 2>&1 | tee gpurun_out/$T/ab.log
