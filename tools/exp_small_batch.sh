# round-5: the notebooks' regime (batch_size 4 x aggregate_k_gradients 25) on more alternating streams, and sampled clocks / power during the default step
mkdir -p gpurun_out/c7
for n in 8 12 16 25; do
  python bench.py --batch 4 --aggregate-k 25 --aggregate-streams $n --steps 4 --warmup 2 --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown 2>/dev/null | tail -1 > gpurun_out/c7/b4x25_streams$n.json
done
for n in 4 8; do
  python bench.py --batch 8 --aggregate-k 8 --aggregate-streams $n --steps 4 --warmup 2 --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown 2>/dev/null | tail -1 > gpurun_out/c7/b8x8_streams$n.json
done
# clocks and power while the default step runs
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower --json 2>/dev/null | head -c 1500; echo; sleep 0.5; done > gpurun_out/c7/smi_samples.txt ) &
SMI=$!
python bench.py --steps 300 --warmup 5 --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown 2>/dev/null | tail -1 > gpurun_out/c7/step_long.json
wait $SMI
for f in gpurun_out/c7/b*.json gpurun_out/c7/step_long.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done
head -c 1200 gpurun_out/c7/smi_samples.txt
