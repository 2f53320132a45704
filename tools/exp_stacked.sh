mkdir -p gpurun_out/c9
python -m pytest tests/test_gpu_parity.py -q -x -k "forward_batches or training_loop_vs_reference or deterministic_schedule_is or data_parallel_train" 2>&1 | tail -30 > gpurun_out/c9/parity.log
python -m pytest tests/test_gpu_ops.py -q -x -k "attention or embed or gather" 2>&1 | tail -5 > gpurun_out/c9/ops.log
python -m pytest tests/test_gpu_parity.py -q -x -k "gp_prior or gp_posterior" 2>&1 | tail -5 > gpurun_out/c9/gp.log
python tools/bench_gp.py --batch 320 > gpurun_out/c9/gp_bench.txt 2>&1
for mode in stacked alt8; do
  F="--aggregate-stacked"; [ $mode = alt8 ] && F="--aggregate-streams 8"
  python bench.py --batch 4 --aggregate-k 25 $F --steps 4 --warmup 2 --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown 2>gpurun_out/c9/b4_$mode.err | tail -1 > gpurun_out/c9/b4x25_$mode.json
done
python bench.py --batch 8 --aggregate-k 8 --aggregate-stacked --steps 4 --warmup 2 --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown 2>/dev/null | tail -1 > gpurun_out/c9/b8x8_stacked.json
python bench.py --batch 16 --aggregate-k 4 --aggregate-stacked --steps 4 --warmup 2 --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown 2>/dev/null | tail -1 > gpurun_out/c9/b16x4_stacked.json
tail -12 gpurun_out/c9/parity.log; cat gpurun_out/c9/ops.log gpurun_out/c9/gp.log gpurun_out/c9/gp_bench.txt | grep -v amdgpu
for f in gpurun_out/c9/b*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done; tail -5 gpurun_out/c9/b4_stacked.err
