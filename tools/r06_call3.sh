# round 6, GPU call 3: what the key centring and the fused Q projection cost / gain in the step (same box, alternating), the new parity tests
mkdir -p gpurun_out/r06c3
O=gpurun_out/r06c3
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "q_projection_inside or self_launches or trained_checkpoint" 2>&1 | tail -15 | tee $O/pytest_new.log
python tools/bench_qkv.py 2>/dev/null | tee $O/bench_qkv.txt
python tools/bench_attn.py 32 2000 512 4 1604 2>/dev/null | tee $O/bench_attn.txt
for prec in fp16 bf16; do
B="python bench.py --precision $prec --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown"
for rep in 1 2 3; do
  for T in "" "13=0" "12=1" "12=1,13=0"; do
    TT=""; [ -n "$T" ] && TT="--tune $T"
    timeout 300 $B $TT 2>/dev/null | tail -1 > $O/step_${prec}_tune_${T:-default}_$rep.json
    python -c "
import json; d=json.load(open('$O/step_${prec}_tune_${T:-default}_$rep.json')); print('$prec tune ${T:-default} rep $rep', d['value'], d['ms_per_step'])" | tee -a $O/step_tunes.txt
  done
done
done
