#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4k; mkdir -p $O
{ echo "== in-process setdefault"; timeout 120 python $R/tools/exp_fault.py nccl_both 2>&1 | grep -v 'amdgpu.ids\|socket.cpp\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl\|Extension modules\|Warning' | tail -4
cd $R && timeout 900 python -m pytest tests/test_gpu_frames.py tests/test_gpu_rccl_world1.py -x -q -m gpu 2>&1 | tail -4
} > $O/out.txt 2>&1
cat $O/out.txt
