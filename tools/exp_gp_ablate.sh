#!/bin/bash
# Where the GP sampler's trailing update spends its time: ablation builds (PFN_SYP_ABLATE, wrong results by construction) in _variants/, per-kernel times by rocprofv3.
#   gpurun -- 'bash tools/exp_gp_ablate.sh'   -> gpurun_out/gp_ablate.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $ROOT/gpurun_out; O=$ROOT/gpurun_out/gp_ablate.txt; : > $O
cd /tmp && export TMPDIR=/tmp
for lib in "" $ROOT/transformerscandobayesianinference_amd/_variants/*.so; do
  export PFN_LIB=$lib
  rm -rf /tmp/gpprof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gpprof -o gp -- python $ROOT/tools/bench_gp.py --batch 320 --no-check > /tmp/gpprof.log 2>&1
  f=$(find /tmp/gpprof -name "*kernel_stats.csv" | head -1)
  echo "--- $(basename ${lib:-product}): $(grep 'gp draw' /tmp/gpprof.log)" >> $O
  [ -n "$f" ] && grep "syrk_planes\|trsm_wide" "$f" | cut -d, -f1-5 | cut -c1-150 >> $O || tail -3 /tmp/gpprof.log >> $O
done
