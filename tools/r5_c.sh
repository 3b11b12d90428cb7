#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
T=r5c; mkdir -p gpurun_out/$T
timeout 600 python -m pytest tests/test_gpu_frames.py -q -x > gpurun_out/$T/pytest_frames.log 2>&1; tail -15 gpurun_out/$T/pytest_frames.log
timeout 600 python bench.py --no-cpu-baseline --train-iters 0 > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; tail -3 gpurun_out/$T/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5c/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], json.dumps(d.get('api_frame') or d.get('variants', {}).get('api_frame'), indent=1)[:1500])
PY
