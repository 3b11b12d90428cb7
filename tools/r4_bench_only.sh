#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
mkdir -p gpurun_out/r4final3_prof
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4final3_prof/bench.json 2> gpurun_out/r4final3_prof/bench.err
python -c "
import json; d=json.loads(open('gpurun_out/r4final3_prof/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['counters_stale'], d['roofline']['frac'], d['roofline']['traffic'], d['path_roofline']['measured_bytes_per_step'])"
