#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
mkdir -p gpurun_out/r4san
timeout 120 python bench.py --no-cpu-baseline --no-variants --train-iters 0 --steps 20 --warmup 5 > gpurun_out/r4san/n1.json 2> gpurun_out/r4san/n1.err
python -c "
import json; d=json.loads(open('gpurun_out/r4san/n1.json').read().strip().splitlines()[-1]); print('N=1', d['ms_per_step'], d['config']['hip_graph'], d['config']['frames_in_flight'])" || tail -5 gpurun_out/r4san/n1.err
INVR_FORCE_DEVICE=0 INVR_DIST_BACKEND=gloo timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 5 --no-cpu-baseline --no-variants --train-iters 0 --in-flight 5 > gpurun_out/r4san/n2.json 2> gpurun_out/r4san/n2.err
python -c "
import json
for l in open('gpurun_out/r4san/n2.json').read().strip().splitlines():
    if l.startswith('{'):
        d=json.loads(l); print('N=2', d['ms_per_step'], d['n_gpus'], d['config']['hip_graph'], d['config'].get('exchange_captured_in_graph'), d['config']['parallelism'][:60])" || tail -8 gpurun_out/r4san/n2.err
tail -3 gpurun_out/r4san/n2.err
