#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
T=r5b; mkdir -p gpurun_out/$T
timeout 600 python tools/exp_streams.py 4 8 10 12 16 > gpurun_out/$T/streams.log 2>&1; tail -12 gpurun_out/$T/streams.log
timeout 600 python tools/exp_streams.py --shard-of 8 1 4 10 16 > gpurun_out/$T/streams8.log 2>&1; tail -12 gpurun_out/$T/streams8.log
