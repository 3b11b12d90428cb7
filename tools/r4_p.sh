#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4p; mkdir -p $O
{ rocprofv3 --kernel-trace --output-format csv -d $O/tr -o tr -- python $R/tools/train_bench.py --iters 30 > $O/tr.log 2>&1
grep -E "iteration|synchronised" $O/tr.log
python $R/tools/trace_timeline.py $(find $O/tr -name '*kernel_trace.csv' | head -1) k_adam 20
find $O/tr -name '*.csv' -delete
cd $R && timeout 900 python -m pytest tests/test_gpu_training.py -x -q -m gpu -k "configs3_real_shape" -s 2>&1 | grep -E "agreement|passed|failed|Error|losses" | head -20
} > $O/out.txt 2>&1
cat $O/out.txt
