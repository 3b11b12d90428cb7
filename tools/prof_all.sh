#!/bin/bash
# bench line + rocprofv3 kernel stats + PMC passes of the same command (counters in their own runs, kernel-trace only):  bash tools/prof_all.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1
OUT=$R/gpurun_out/$T; mkdir -p $OUT
python $R/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.json
python -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.csrc_digest())" > $OUT/csrc_digest.txt
B="python $R/bench.py --no-cpu-baseline --no-variants --no-graph --train-iters 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $B --steps 10 --warmup 3 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $B --steps 2 --warmup 1 > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/write -o write -- $B --steps 2 --warmup 1 > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $OUT/sq -o sq -- $B --steps 2 --warmup 1 > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o mfma -- $B --steps 2 --warmup 1 > $OUT/mfma.log 2>&1
# the 64-byte-row encoder (training forward / eval_row_sums False): its own trace + byte counters (VERDICT r3 missing #5)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fr_trace -o trace -- $B --full-rows --steps 6 --warmup 2 > $OUT/fr_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fr_fetch -o fetch -- $B --full-rows --steps 2 --warmup 1 > $OUT/fr_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/fr_write -o write -- $B --full-rows --steps 2 --warmup 1 > $OUT/fr_write.log 2>&1
# keep only the small summaries (the raw traces exceed the 64 MiB pull limit)
find $OUT -name "*_agent_info.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
ls -R $OUT | head -40; du -sh $OUT
