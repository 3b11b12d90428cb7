#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
T=${1:-r4mid}
mkdir -p gpurun_out/$T
timeout 400 python -m pytest tests/test_gpu_production_kernels.py tests/test_gpu_parity.py -q -x -m gpu -k "cull_survivor or aggr_mean or fuse or fused_adam or (render_config_variants and not over1)" > gpurun_out/$T/pytest.log 2>&1; tail -4 gpurun_out/$T/pytest.log
for K in 10 5 4; do
  timeout 200 python bench.py --no-cpu-baseline --no-variants --train-iters 0 --in-flight $K --steps 20 --warmup 5 > gpurun_out/$T/inflight_$K.json 2> gpurun_out/$T/inflight_$K.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/$T/inflight_$K.json').read().strip().splitlines()[-1]); print('in-flight $K:', d['ms_per_step'], d['config']['frames_in_flight'], d['stage_ms_per_step']['cull'])" || tail -3 gpurun_out/$T/inflight_$K.err
done
