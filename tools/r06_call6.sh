# round 6, GPU call 6: the whole GPU suite in assert mode (bounds recorded), smoke, cost of the key centring after the key_shift rewrite, batch / stream sweep
mkdir -p gpurun_out/r06c6
O=gpurun_out/r06c6
python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/smoke.log
PFN_RECORD_BOUNDS=$O/measured.json timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest.log
tail -25 $O/pytest.log
python tools/bench_qkv.py 2>/dev/null | tee $O/bench_qkv.txt
B="python bench.py --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown"
for rep in 1 2; do
  for A in "" "--tune 13=0" "--batch 96 --streams 3" "--batch 128 --streams 4" "--batch 128 --streams 2" "--batch 96 --streams 2" "--precision bf16"; do
    timeout 300 $B $A 2>/dev/null | tail -1 > $O/step.json
    python -c "
import json; d=json.load(open('$O/step.json')); print('$A rep $rep:', d['dtype'], d['value'], d['ms_per_step'], d['config']['per_gpu_batch'], d['config']['micro_batch_streams'])" | tee -a $O/steps.txt
  done
done
