# round 6, GPU call 11: the pre-LayerNorm sums in operand precision (fp16 models; PFN_SCHED_F32_RESIDUAL / PFN_TUNE_RESIDUAL16 = 0 keeps f32): op and model tests,
# then same-box A/B of the step with 16-bit sums (default) against f32 sums, timed-path parity of both
mkdir -p gpurun_out/r06c11
O=gpurun_out/r06c11
timeout 1500 python -m pytest tests -m gpu -q -x -k "gemm_ln or lnbwd or fp16 or sums" 2>&1 | tail -8 > $O/pytest_sums.log
tail -4 $O/pytest_sums.log
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d.get('parity_timed_path') or {}
        print('$1', d['value'], d['ms_per_step'], 'timed-path parity', {k: p.get(k) for k in ('nll_rel', 'mean_rel_l2', 'logits_rel_l2', 'mean_max_over_range')})"; }
for rep in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-kernel-breakdown --no-extras 2>/dev/null | line "sums16 rep$rep" | tee -a $O/ab.txt
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-kernel-breakdown --no-extras --tune 16=0 2>/dev/null | line "sums32 rep$rep" | tee -a $O/ab.txt
done
timeout 900 python bench.py --no-cpu-baseline --no-extras > $O/bench_sums16.line 2>/dev/null; cp bench_detail.json $O/bench_sums16.json; line "sums16 full" < $O/bench_sums16.line | tee -a $O/ab.txt
timeout 900 python bench.py --no-cpu-baseline --no-extras --tune 16=0 > $O/bench_sums32.line 2>/dev/null; cp bench_detail.json $O/bench_sums32.json; line "sums32 full" < $O/bench_sums32.line | tee -a $O/ab.txt
python - <<'PY' | tee -a gpurun_out/r06c11/ab.txt
import json
for v in ('sums16', 'sums32'):
    d = json.load(open(f'gpurun_out/r06c11/bench_{v}.json'))
    kb = d.get('kernel_breakdown') or d.get('kernels') or []
    for k in kb if isinstance(kb, list) else []:
        n = k.get('name', '')
        if 'gemm_nt_ln' in n:
            print(v, n, {a: k.get(a) for a in ('us', 'isolated_us', 'tflops', 'gbps', 'count')})
PY
