#!/bin/bash
# One entry point for the GPU-box runs of a round:   gpurun -- 'bash tools/gpu.sh <task> [tag] [args...]'   (outputs under gpurun_out/<tag>/)
#   tests [tag] [pytest args]   the -m gpu suite (or a selection: ... tests t1 tests/test_gpu_frames.py -k streams)
#   trace [tag]                 rocprofv3 kernel trace of the eager bench frame, top kernels printed (tools/prof_all.sh has the PMC passes)
#   prof  [tag]                 tools/prof_all.sh: bench line + kernel stats + PMC passes of the same binary
#   streams [tag] [K ...]       K frames in flight: branches of one hipGraph against K eager launch chains on K streams (whole frame + 1/8 shard)
#   api [tag]                   Renderer.render with frames in flight across calls: ms per frame, allocator footprint (tools/exp_inflight_diag.py)
#   world8 [tag]                8 gloo ranks on this one GPU through bench.py --gpus 8 (eval) and --train --gpus 8
#   ab [tag] libA libB [n] [bench args]    alternating A/B of two library builds on this box (tools/build_variant.sh); "--train" for the training line
#   abn [tag] n lib1 lib2 ...             the same for any number of builds
#   final [tag]                 tests + prof + a default bench line: the closing run of a round
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
task=$1; T=${2:-$1}; shift 2 2>/dev/null
OUT=gpurun_out/$T; mkdir -p $OUT
case $task in
  tests)
    if [ $# -eq 0 ]; then set -- tests; fi          # ("$@": a -k expression with spaces stays one argument)
    timeout 1700 python -m pytest "$@" -q -m gpu --durations=8 > $OUT/pytest.log 2>&1; tail -25 $OUT/pytest.log ;;
  tol)
    # the headroom of the widened tolerances, five runs: INVR_TOL_REPORT lines of the three tests that carry them
    export INVR_TOL_REPORT=$GRAFT_REPO_ROOT/$OUT/tol_report.txt; rm -f $INVR_TOL_REPORT
    for i in 1 2 3 4 5; do
      timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_production_kernels.py tests/test_gpu_fullsize.py -q -m gpu -k "configs4 or strict_1e4 or spot" > $OUT/pytest_$i.log 2>&1; tail -2 $OUT/pytest_$i.log
    done
    sort $INVR_TOL_REPORT | uniq -c | sort -k2 | head -80 ;;
  trace)
    cd /tmp
    B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --no-graph --train-iters 0"
    rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o trace -- $B --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1
    cd $GRAFT_REPO_ROOT
    find $OUT -name "*_agent_info.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
    python - <<PY
import csv, glob
f = glob.glob('$OUT/trace/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print('%-44s %5s calls  avg %9.1f us  min %9.1f us' % (r['Name'][:44], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
    ;;
  prof)
    bash tools/prof_all.sh $T > $OUT/prof_all.log 2>&1; tail -5 $OUT/prof_all.log ;;
  streams)
    timeout 600 python tools/exp_streams.py ${@:-4 8 10 16} 2>&1 | grep -v amdgpu.ids | tee $OUT/streams.log
    timeout 600 python tools/exp_streams.py --shard-of 8 ${@:-1 4 10 16} 2>&1 | grep -v amdgpu.ids | tee $OUT/streams_shard8.log ;;
  api)
    timeout 600 python tools/exp_inflight_diag.py 2>&1 | grep "to_cpu" | tee $OUT/api.log ;;
  world8)
    export INVR_DIST_BACKEND=gloo INVR_FORCE_DEVICE=0
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --steps 20 --warmup 10 > $OUT/eval_w8.json 2> $OUT/eval_w8.err
    echo "eval world 8 rc=$?"; tail -c 600 $OUT/eval_w8.json
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29518 bench.py --train --gpus 8 --steps 10 --warmup 3 > $OUT/train_w8.json 2> $OUT/train_w8.err
    echo "train world 8 rc=$?"; tail -c 600 $OUT/train_w8.json ;;
  ab)
    A=$1; B=$2; N=${3:-2}; shift 3 2>/dev/null
    for i in $(seq $N); do for L in $A $B; do
      if [ "$1" = "--train" ]; then
        INVR_LIB_PATH=$GRAFT_REPO_ROOT/$L timeout 300 python bench.py --train --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$L %.3f ms/iter  k_adam %.3f ms  %.0f GB/s' % (d['ms_per_step'], r['kernel_ms_per_launch'], r['achieved']))"
      else
        INVR_LIB_PATH=$GRAFT_REPO_ROOT/$L timeout 300 python bench.py --steps 20 --warmup 5 --train-iters 0 --no-cpu-baseline --no-variants "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L %.4f ms' % d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items() if v})"
      fi
    done; done 2>&1 | tee $OUT/ab.log ;;
  abn)
    # abn [tag] n lib1 lib2 ...: alternating runs of any number of library builds on this box
    N=$1; shift
    for i in $(seq $N); do for L in "$@"; do
      INVR_LIB_PATH=$GRAFT_REPO_ROOT/$L timeout 300 python bench.py --steps 20 --warmup 5 --train-iters 0 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L %.4f ms' % d['ms_per_step'], d.get('replay_bit_exact'), {k: round(v,3) for k,v in d['stage_ms_per_step'].items() if v})"
    done; done 2>&1 | tee $OUT/ab.log ;;
  final)
    timeout 1700 python -m pytest tests -q -m gpu --durations=8 > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log
    bash tools/prof_all.sh ${T}_prof > $OUT/prof_all.log 2>&1; tail -5 $OUT/prof_all.log
    cd $GRAFT_REPO_ROOT
    timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/bench_default.json ;;
  *) echo "unknown task $task"; exit 2 ;;
esac
