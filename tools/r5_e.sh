#!/bin/bash
# world-8 rehearsal on ONE GPU: 8 processes (gloo, every rank pinned to device 0) through the whole bench path — eval frames and the
# data-parallel training line — to prove the rank-agreed mode fall-back, the exchange and the JSON line at the node size the driver uses
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
T=r5e; mkdir -p gpurun_out/$T
export INVR_DIST_BACKEND=gloo INVR_FORCE_DEVICE=0
for N in 8; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --steps 20 --warmup 10 > gpurun_out/$T/eval_w$N.json 2> gpurun_out/$T/eval_w$N.err
  echo "eval world $N rc=$?"; tail -c 1500 gpurun_out/$T/eval_w$N.json; tail -3 gpurun_out/$T/eval_w$N.err
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29518 bench.py --train --gpus $N --steps 10 --warmup 3 > gpurun_out/$T/train_w$N.json 2> gpurun_out/$T/train_w$N.err
  echo "train world $N rc=$?"; tail -c 1200 gpurun_out/$T/train_w$N.json; tail -3 gpurun_out/$T/train_w$N.err
done
