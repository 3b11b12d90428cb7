#!/bin/bash
# round-4 closing run: the whole -m gpu suite, then tools/prof_all.sh on the same binary, then two side probes of the bench line
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
T=${1:-r4final2}
mkdir -p gpurun_out/$T
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/$T/pytest.log 2>&1; tail -15 gpurun_out/$T/pytest.log
bash tools/prof_all.sh ${T}_prof > gpurun_out/$T/prof_all.log 2>&1; tail -5 gpurun_out/$T/prof_all.log
cd $GRAFT_REPO_ROOT
for K in 2 8; do
  timeout 200 python bench.py --no-cpu-baseline --no-variants --train-iters 0 --in-flight $K --steps 24 --warmup 8 > gpurun_out/$T/inflight_$K.json 2> gpurun_out/$T/inflight_$K.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/$T/inflight_$K.json').read().strip().splitlines()[-1]); print('in-flight $K:', d['ms_per_step'])"
done
