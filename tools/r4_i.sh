#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4i; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 5 --train-iters 0 --no-cpu-baseline"
{ $B --no-variants > $O/b1.json 2> $O/b1.err; tail -c 600 $O/b1.err; python -c "
import json; d=json.load(open('$O/b1.json')); print(d['ms_per_step'], d['value'], d['config']['frames_in_flight'], d['config']['rays_per_frame'], d['config']['active_samples_per_frame_rank0'])"
$B > $O/b2.json 2> $O/b2.err; tail -c 600 $O/b2.err; python -c "
import json; d=json.load(open('$O/b2.json')); print(d['ms_per_step']); [print(k, json.dumps(d[k])[:600]) for k in ('shard_projection','samples_64','mid_density','full_rows','dense_stress','api_frame','variants_error') if k in d]"
} > $O/out.txt 2>&1
cat $O/out.txt
