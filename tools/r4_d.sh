#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4d; mkdir -p $O
{ for K in 2 3 4 5 8; do timeout 120 python $R/tools/exp_pair_graph.py 8 $K side 2>&1 | grep -v amdgpu.ids | tail -1; done
  for K in 2 4; do timeout 120 python $R/tools/exp_pair_graph.py 1 $K side 2>&1 | grep -v amdgpu.ids | tail -1; done
  for K in 4; do timeout 120 python $R/tools/exp_pair_graph.py 2 $K side 2>&1 | grep -v amdgpu.ids | tail -1; timeout 120 python $R/tools/exp_pair_graph.py 4 $K side 2>&1 | grep -v amdgpu.ids | tail -1; done
  } > $O/out3.txt 2>&1
cat $O/out3.txt
