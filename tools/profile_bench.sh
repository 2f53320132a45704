#!/bin/bash
# Profiles of the bench command on the GPU box (writes under gpurun_out/$1):
#   1. rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 3` (per-kernel durations)
#   2. separate PMC passes (FETCH_SIZE; WRITE_SIZE; MFMA busy) of a shorter run, kernel trace only
# usage: [BENCH_ARGS="--config 4 --precision fp16"] tools/profile_bench.sh <tag>      (BENCH_ARGS: the bench command's own flags -- configuration, operand format)
set -u
TAG=${1:-prof}
BENCH_ARGS=${BENCH_ARGS:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $ROOT/bench.py $BENCH_ARGS --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-kernel-breakdown --no-extras > $OUT/bench_stats_run.log 2>&1   # steps only: every row is an in-step average
# the same launch shapes (one micro-batch: MICRO_BATCH datasets, default configs[1]'s 64) on ONE stream: kernel durations without co-runners
[ -z "${SKIP_STREAMS1:-}" ] && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o bench -- python $ROOT/bench.py $BENCH_ARGS --steps 10 --warmup 3 --batch ${MICRO_BATCH:-64} --streams 1 --no-cpu-baseline --no-parity --no-kernel-breakdown --no-extras > $OUT/bench_stats1_run.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  name=$(echo $c | tr ' ' '_')
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$name -o bench -- python $ROOT/bench.py $BENCH_ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-kernel-breakdown --no-extras > $OUT/pmc_$name.log 2>&1
done
cd $ROOT
grep '^{"metric"' $OUT/bench_stats_run.log > $OUT/bench.json
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -60
