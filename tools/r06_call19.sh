# round 6, GPU call 19: per-GPU batch / micro-batch streams of the configs[1] line with the final kernels (same box, interleaved)
mkdir -p gpurun_out/r06c19
O=gpurun_out/r06c19
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', round(d['value'], 1), round(d['ms_per_step'], 3), d['config']['per_gpu_batch'], d['config']['micro_batch_streams'])"; }
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-kernel-breakdown --no-extras"
for rep in 1 2 3; do
  for cfg in "" "--batch 96 --streams 3" "--batch 128 --streams 4" "--batch 160 --streams 5" "--batch 192 --streams 3"; do
    timeout 600 python bench.py $Q $cfg 2>/dev/null | line "[$cfg] rep$rep" | tee -a $O/ab.txt
  done
done
