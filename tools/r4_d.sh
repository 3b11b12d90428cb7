#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4d; mkdir -p $O
{ for K in 2; do timeout 120 python -X faulthandler $R/tools/exp_pair_graph.py 8 $K 2>&1 | grep -v amdgpu.ids | tail -30; echo "rc=$?"; done; } > $O/out.txt 2>&1
cat $O/out.txt
