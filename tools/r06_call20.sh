# round 6, GPU call 20: per-GPU batch / micro-batch streams of the configs[3] and configs[4] lines (same box, interleaved)
mkdir -p gpurun_out/r06c20
O=gpurun_out/r06c20
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', round(d['value'], 1), round(d['ms_per_step'], 3), d['config']['per_gpu_batch'], d['config']['micro_batch_streams'], 'loss', d['config'].get('final_loss'))"; }
Q="--steps 12 --warmup 3 --no-cpu-baseline --no-parity --no-kernel-breakdown --no-extras"
for rep in 1 2; do
  for cfg in "" "--batch 12 --streams 3" "--batch 24 --streams 3" "--batch 16 --streams 2" "--batch 48 --streams 3"; do
    timeout 900 python bench.py --config 5 $Q $cfg 2>/dev/null | line "config5 [$cfg] rep$rep" | tee -a $O/ab.txt
  done
  for cfg in "" "--batch 96 --streams 3" "--batch 192 --streams 3" "--batch 384 --streams 3"; do
    timeout 900 python bench.py --config 4 $Q $cfg 2>/dev/null | line "config4 [$cfg] rep$rep" | tee -a $O/ab.txt
  done
  for cfg in "--batch 288 --streams 3" "--batch 192 --streams 2" "--batch 384 --streams 3"; do
    timeout 900 python bench.py $Q $cfg 2>/dev/null | line "config2 [$cfg] rep$rep" | tee -a $O/ab.txt
  done
done
