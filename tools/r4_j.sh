#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/r4j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_frames.py tests/test_gpu_rccl_world1.py -x -q -m gpu > $O/out.txt 2>&1; tail -15 $O/out.txt
