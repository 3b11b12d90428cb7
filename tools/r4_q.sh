#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4q; mkdir -p $O
{ python $R/tools/train_bench.py --iters 40 2>&1 | grep -E "iteration|synchronised"
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o tr -- python $R/tools/train_bench.py --iters 30 > $O/tr.log 2>&1
python $R/tools/trace_timeline.py $(find $O/tr -name '*kernel_trace.csv' | head -1) k_adam 20 | awk '$3+0 > 8.0 || /unit/ || /gap *\+[0-9][0-9]/'
find $O/tr -name '*.csv' -delete
cd $R && timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py tests/test_gpu_dist_train.py -x -q -m gpu -k "train and not configs3_real_shape" 2>&1 | tail -4
} > $O/out.txt 2>&1
cat $O/out.txt
