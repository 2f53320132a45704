# round 6, GPU call 23: configs[1] at its batch of 64 on 2 / 3 / 4 / 5 micro-batch streams (uneven column groups: 22 + 21 + 21 ...), same box, with timed-path parity
mkdir -p gpurun_out/r06c23
O=gpurun_out/r06c23
timeout 600 python -m pytest tests -m gpu -q -x -k "micro_batch_streams or train_entry or training_loop" 2>&1 | tail -3 | tee $O/pytest.log
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d.get('parity_timed_path') or {}
        print('$1', round(d['value'], 1), round(d['ms_per_step'], 3), d['config']['per_gpu_batch'], d['config']['micro_batch_streams'], {k: p.get(k) for k in ('mean_rel_l2', 'mean_ref_rms')})"; }
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-breakdown --no-extras"
for rep in 1 2 3; do
  for st in 2 3 4 5; do
    timeout 600 python bench.py $Q --streams $st 2>/dev/null | line "streams $st rep$rep" | tee -a $O/ab.txt
  done
done
