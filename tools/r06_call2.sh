# round 6, GPU call 2: suites with the key centring; in-step A/B of the attention backward's dataset groups (PFN_TUNE_ATTN_BWD_GROUP) and of the weight-gradient splits
mkdir -p gpurun_out/r06c2
O=gpurun_out/r06c2
PFN_RECORD_BOUNDS=$O/measured_ops.json timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q 2>&1 | tail -25 > $O/pytest_ops.log
tail -6 $O/pytest_ops.log
PFN_BOUNDS_MEASURE_ONLY=1 PFN_RECORD_BOUNDS=$O/measured_parity.json timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -60 > $O/pytest_parity.log
tail -30 $O/pytest_parity.log
# isolated: the backward's launch set at the micro-batch shape, by group size
for G in 0 2 4 8 16; do
  PFN_TUNE=10=$G python tools/bench_attn.py 32 2000 512 4 1604 2>/dev/null | grep "all launches" | sed "s/^/group $G: /" | tee -a $O/attn_groups_isolated.txt
done
B="python bench.py --precision fp16 --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown"
for rep in 1 2 3; do
  for T in "" "10=2" "10=4" "10=8" "10=16" "11=2" "11=3" "11=4" "11=5"; do
    TT=""; [ -n "$T" ] && TT="--tune $T"
    timeout 300 $B $TT 2>/dev/null | tail -1 > $O/step_tune_${T:-default}_$rep.json
    python -c "
import json; d=json.load(open('$O/step_tune_${T:-default}_$rep.json')); print('tune ${T:-default} rep $rep', d['value'], d['ms_per_step'])" | tee -a $O/step_tunes.txt
  done
done
