# round 6, GPU call 4: the four-wave 128 x 128 weight-gradient kernel (PFN_TUNE_WGRAD_WAVES = 4) and the LDS-staged key shift: tests, isolated, in step
mkdir -p gpurun_out/r06c4
O=gpurun_out/r06c4
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "gemm_tn or qkv_projection" 2>&1 | tail -8 | tee $O/pytest_ops.log
python tools/bench_qkv.py 2>/dev/null | tee $O/bench_qkv.txt
for T in "" "14=4" "14=4,11=2" "14=4,11=3" "14=4,11=4" "14=4,11=5" "14=4,11=6" "14=4,11=8"; do
  python tools/bench_wgrad.py --batch 32 --tune "$T" 2>/dev/null | tee -a $O/bench_wgrad.txt
done
B="python bench.py --precision fp16 --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown"
for rep in 1 2 3; do
  for T in "" "14=4" "13=0" "14=4,13=0"; do
    TT=""; [ -n "$T" ] && TT="--tune $T"
    timeout 300 $B $TT 2>/dev/null | tail -1 > $O/step_tune_${T:-default}_$rep.json
    python -c "
import json; d=json.load(open('$O/step_tune_${T:-default}_$rep.json')); print('fp16 tune ${T:-default} rep $rep', d['value'], d['ms_per_step'])" | tee -a $O/step_tunes.txt
  done
done
