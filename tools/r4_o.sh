#!/bin/bash
# training: timeline of one iteration + the full configs[3] schedule run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4o; mkdir -p $O
{ rocprofv3 --kernel-trace --output-format csv -d $O/tr -o tr -- python $R/tools/train_bench.py --iters 30 > $O/tr.log 2>&1
tail -4 $O/tr.log
python $R/tools/trace_timeline.py $(find $O/tr -name '*kernel_trace.csv' | head -1) k_adam 20
find $O/tr -name '*.csv' -delete
cd $R && timeout 900 python tools/train_lan_full.py $O/configs3_full_run.json 2>&1 | grep -v amdgpu.ids | tail -5
} > $O/out.txt 2>&1
tail -150 $O/out.txt
