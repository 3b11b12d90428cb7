#!/bin/bash
# 2-rank smoke of the multi-GPU bench path on ONE GPU (gloo: RCCL refuses two ranks on one device) + the frames / RCCL tests
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4l; mkdir -p $O
{ INVR_DIST_BACKEND=gloo INVR_FORCE_DEVICE=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 $R/bench.py --gpus 2 --steps 8 --warmup 4 --min-time 0.3 --table-log2 14 > $O/b2.json 2> $O/b2.err; tail -c 800 $O/b2.err | grep -v amdgpu.ids
python -c "
import json; d=json.load(open('$O/b2.json')); print(d['n_gpus'], d['ms_per_step'], d['value'], d['config']['parallelism'], d['config']['exchange_only_ms_per_replay'])"
cd $R && timeout 900 python -m pytest tests/test_gpu_frames.py tests/test_gpu_rccl_world1.py -x -q -m gpu 2>&1 | tail -3
} > $O/out.txt 2>&1
cat $O/out.txt
