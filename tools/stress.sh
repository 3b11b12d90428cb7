#!/bin/bash
# gpurun -- 'bash tools/stress.sh <tag> [set]' : the frames-in-flight stress runs of VERDICT r5 #1 (outputs under gpurun_out/<tag>/)
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
T=${1:-stress}; SET=${2:-all}; OUT=gpurun_out/$T; mkdir -p $OUT
run() { name=$1; shift; timeout 900 python tools/stress_replay.py "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? $(tail -c 300 $OUT/$name.json)"; grep "MISMATCH\|VERIFY" $OUT/$name.err | cut -c1-1500 | head -4; }
case $SET in
  all)
    run small_graph        --config small --mode graph   --replays 3000 --snapshot
    run small_graph_rebuild --config small --mode graph  --replays 1500 --rebuild 25 --snapshot
    run small_streams      --config small --mode streams --replays 1500 --snapshot
    run bench_graph        --config bench --mode graph   --replays 300 --snapshot
    run bench_lanes        --config bench --mode lanes   --replays 40
    run shard_graph        --config shard --mode graph   --replays 600 --snapshot ;;
  diag)
    INVR_LIB_PATH=$GRAFT_REPO_ROOT/variants/libinvr_verify.so run verify_bench_graph --config bench --mode graph --replays 250 --snapshot
    run bench_graph_k4   --config bench --mode graph --frames 4 --replays 250 --snapshot
    run bench_eager_k10  --config bench --mode eager --replays 40
    GPU_MAX_HW_QUEUES=1 run bench_streams_q1 --config bench --mode streams --replays 100 ;;
  diag2)
    for v in reload noloop fullwait; do
      INVR_LIB_PATH=$GRAFT_REPO_ROOT/variants/libinvr_$v.so run ${v}_bench_graph --config bench --mode graph --replays 300 --snapshot
    done ;;
  diag3)
    INVR_LIB_PATH=$GRAFT_REPO_ROOT/variants/libinvr_twice.so run twice_bench_graph --config bench --mode graph --replays 500 ;;
  diag4)
    run base_bench_graph --config bench --mode graph --replays 300
    for v in e5 e6 e7 e8; do
      INVR_LIB_PATH=$GRAFT_REPO_ROOT/variants/libinvr_$v.so run ${v}_bench_graph --config bench --mode graph --replays 300
    done ;;
  diag5)
    run base_bench_graph --config bench --mode graph --replays 120 --snapshot ;;
  diag6)
    run base_bench_graph --config bench --mode graph --replays 150
    for v in ${VARS:-e9 e10}; do
      INVR_LIB_PATH=$GRAFT_REPO_ROOT/variants/libinvr_$v.so run ${v}_bench_graph --config bench --mode graph --replays 300
    done ;;
  diag7)
    INVR_LIB_PATH=$GRAFT_REPO_ROOT/variants/libinvr_dump.so run dump_bench_graph --config bench --mode graph --replays 200 --snapshot ;;
  final)
    run small_graph   --config small --mode graph   --replays 2000
    run small_streams --config small --mode streams --replays 1000
    run bench_graph   --config bench --mode graph   --replays 2000 --check-every 1
    run bench_lanes   --config bench --mode lanes   --replays 200
    run shard_graph   --config shard --mode graph   --replays 2000 ;;
esac
