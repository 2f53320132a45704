# round 6, GPU call 10: the whole GPU suite with every asserted value recorded (current build: saturating fp16 stores, loss-scale target 2), and 300 optimizer steps of
# configs[3] / configs[4] in fp16 (12 layers, bptt 4000, emsize 1024: the longest gradient sums) -- the loss must stay finite
mkdir -p gpurun_out/r06c10
O=gpurun_out/r06c10
PFN_RECORD_BOUNDS=$O/parity_measured.json timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest.log
tail -6 $O/pytest.log
for cfg in 5 4; do
  timeout 900 python bench.py --config $cfg --steps 300 --warmup 5 --no-cpu-baseline --no-parity --no-kernel-breakdown --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('config $cfg', d['value'], d['ms_per_step'], 'final_loss', d['config'].get('final_loss'))" | tee -a $O/long_runs.txt
done
