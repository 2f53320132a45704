#!/bin/bash
# GP sampler after a change of its trailing update: tests, accuracy against the f64 oracle, time per dataset and per-kernel times (rocprofv3) for the product
# library and every variant in transformerscandobayesianinference_amd/_variants.
#   gpurun -- 'bash tools/exp_gp_fp16.sh'   -> gpurun_out/gp_fp16.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $ROOT/gpurun_out; O=$ROOT/gpurun_out/gp_fp16.txt; : > $O
cd $ROOT
python -m pytest tests/test_gpu_parity.py -q -k "gp_ or config5_sampler" 2>&1 | tail -1 >> $O
python tools/exp_gp_accuracy.py 2>&1 | grep rel >> $O
cd /tmp && export TMPDIR=/tmp
for lib in "" $ROOT/transformerscandobayesianinference_amd/_variants/*.so; do
  export PFN_LIB=$lib
  python $ROOT/tools/bench_gp.py --batch 640 2>&1 | grep "gp draw" >> $O
  rm -rf /tmp/gpprof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gpprof -o gp -- python $ROOT/tools/bench_gp.py --batch 320 > /tmp/gpprof.log 2>&1
  f=$(find /tmp/gpprof -name "*kernel_stats.csv" | head -1)
  echo "--- kernel stats, $(basename ${lib:-product}) (bench_gp --batch 320, 7 draws): name, calls, total ns, average ns, %" >> $O
  [ -n "$f" ] && python -c "
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]: print(f\"{r['Name'][:44]:46s} calls {r['Calls']:>4s}  total {float(r['TotalDurationNs']) / 1e6:8.3f} ms  avg {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['Percentage']):5.1f} %\")" "$f" >> $O || tail -3 /tmp/gpprof.log >> $O
done
