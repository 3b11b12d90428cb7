#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4m; mkdir -p $O
python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.err
python - <<PY
import json
d=json.load(open('$O/bench.json'))
print('ms', d['ms_per_step'], 'value', d['value'])
for k in ('shard_projection','api_frame','train_step','api_train_step','mid_density','full_rows','samples_64','variants_error'):
    if k in d: print(k, json.dumps(d[k])[:700])
print('cpu', json.dumps(d['cpu_baseline'])[:300])
PY
