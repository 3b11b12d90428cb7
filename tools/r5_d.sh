#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
T=r5d; mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_parity.py tests/test_gpu_rccl_world1.py -q -x -k "adopted or optimisation_steps or fuse or rccl or grad_scaler" > gpurun_out/$T/pytest.log 2>&1; tail -15 gpurun_out/$T/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-variants > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; tail -3 gpurun_out/$T/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5d/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], json.dumps(d.get('train_step'), indent=0)[:600], json.dumps(d.get('api_train_step'), indent=0)[:900])
PY
