#!/bin/bash
# A/B of the survivor order of eval frames on ONE box (INVR_ORDER=0 ray-major / 1 depth-windowed, csrc/k_cull.hip), then the
# reference-golden render test with the windowed order:  bash tools/ab_order.sh [rounds]   -> gpurun_out/ab_order.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; N=${1:-1}; O=$R/gpurun_out/ab_order.txt; mkdir -p $R/gpurun_out; : > $O
for i in $(seq $N); do
  for M in 1 0; do
    INVR_ORDER=$M python $R/bench.py --steps 20 --warmup 5 --train-iters 0 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('INVR_ORDER=$M %.4f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items() if v})" | tee -a $O
  done
done
cd $R && INVR_ORDER=1 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "render_64x64x32 or render_other_scenes or config_variants" 2>&1 | tail -3 | tee -a $O
