#!/bin/bash
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
mkdir -p gpurun_out/r4final
timeout 1300 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r4final/pytest.log 2>&1; tail -15 gpurun_out/r4final/pytest.log
bash tools/prof_all.sh r4final_prof > gpurun_out/r4final/prof_all.log 2>&1; tail -5 gpurun_out/r4final/prof_all.log
