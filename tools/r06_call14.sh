# round 6, GPU call 14: fp16 sums where the LayerNorm is its own kernel (emsize 1024 = configs[4]; PFN_TUNE_RESIDUAL16 = 0 keeps f32): fp16 tests, then same-box A/B
mkdir -p gpurun_out/r06c14
O=gpurun_out/r06c14
timeout 1500 python -m pytest tests -m gpu -q -x -k "fp16 or sums or 1024 or config5 or config4" 2>&1 | tail -8 > $O/pytest.log
tail -4 $O/pytest.log
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d.get('parity_timed_path') or {}
        print('$1', d['value'], d['ms_per_step'], 'final_loss', d['config'].get('final_loss'), 'parity', {k: p.get(k) for k in ('nll_rel', 'mean_rel_l2', 'logits_rel_l2')})"; }
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-breakdown --no-extras"
for rep in 1 2 3; do
  timeout 600 python bench.py --config 5 $Q 2>/dev/null | line "config5 fp16 sums rep$rep" | tee -a $O/ab.txt
  timeout 600 python bench.py --config 5 $Q --tune 16=0 2>/dev/null | line "config5 f32 sums rep$rep" | tee -a $O/ab.txt
done
