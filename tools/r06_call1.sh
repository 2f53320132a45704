# round 6, GPU call 1: fp16 operand path -- probe, op + parity suites with every tolerance recorded (measure-only), same-box bf16 / fp16 bench lines
mkdir -p gpurun_out/r06c1
hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_fp16 tools/probe_fp16.hip && /tmp/probe_fp16 > gpurun_out/r06c1/probe_fp16.txt 2>&1
cat gpurun_out/r06c1/probe_fp16.txt
PFN_RECORD_BOUNDS=gpurun_out/r06c1/measured_ops.json timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r06c1/pytest_ops.log
tail -12 gpurun_out/r06c1/pytest_ops.log
PFN_BOUNDS_MEASURE_ONLY=1 PFN_RECORD_BOUNDS=gpurun_out/r06c1/measured_parity.json timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -60 > gpurun_out/r06c1/pytest_parity.log
tail -40 gpurun_out/r06c1/pytest_parity.log
for rep in 1 2; do
  for prec in bf16 fp16; do
    timeout 600 python bench.py --precision $prec --no-extras --no-cpu-baseline > gpurun_out/r06c1/bench_${prec}_$rep.line 2> gpurun_out/r06c1/bench_${prec}_$rep.err
    cp bench_detail.json gpurun_out/r06c1/bench_${prec}_$rep.json
    python -c "
import json; d=json.load(open('gpurun_out/r06c1/bench_${prec}_$rep.json'))
print('$prec', $rep, d['value'], d['ms_per_step'], d.get('parity_timed_path',{}).get('nll_rel'), d.get('parity_timed_path',{}).get('mean_rel_l2'), d.get('parity_timed_path',{}).get('logits_rel_l2'), d['config'].get('final_loss'))"
  done
done
