#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
mkdir -p gpurun_out/r4drv
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4drv/smoke.log 2>&1; tail -2 gpurun_out/r4drv/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4drv/bench.json 2> gpurun_out/r4drv/bench.err
python -c "
import json; d=json.loads(open('gpurun_out/r4drv/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['counters_stale'], d['csrc_digest'], d['config']['frames_in_flight'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
