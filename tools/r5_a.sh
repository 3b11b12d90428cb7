#!/bin/bash
# round 5, first GPU call: cull patch in the binary (exact survivor set + kernel trace), eager multi-stream frames in flight
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
T=r5a; mkdir -p gpurun_out/$T
timeout 300 python -m pytest tests/test_gpu_production_kernels.py -q -k cull_survivor > gpurun_out/$T/pytest_cull.log 2>&1; tail -3 gpurun_out/$T/pytest_cull.log
bash tools/r4_trace.sh ${T}_trace > gpurun_out/$T/trace.log 2>&1; tail -16 gpurun_out/$T/trace.log
cd $GRAFT_REPO_ROOT
timeout 300 python tools/exp_streams.py 1 2 4 8 > gpurun_out/$T/streams.log 2>&1; tail -6 gpurun_out/$T/streams.log
