#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4u; mkdir -p $O
{ bash $R/tools/ab.sh scratch/libinvr_mlp_f32.so instant-nvr_amd/libinvr.so 2 --in-flight 1
bash $R/tools/ab.sh scratch/libinvr_mlp_f32.so instant-nvr_amd/libinvr.so 2
cd $R && timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_production_kernels.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -4
} > $O/out.txt 2>&1
cat $O/out.txt
