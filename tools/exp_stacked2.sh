mkdir -p gpurun_out/c10
for mode in stacked alt8 seq; do
  F="--aggregate-stacked"; [ $mode = alt8 ] && F="--aggregate-streams 8"; [ $mode = seq ] && F=""
  python bench.py --batch 4 --aggregate-k 25 $F --steps 4 --warmup 2 --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown 2>gpurun_out/c10/b4_$mode.err | tail -1 > gpurun_out/c10/b4x25_$mode.json
done
python bench.py --batch 8 --aggregate-k 8 --aggregate-stacked --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown 2>/dev/null | tail -1 > gpurun_out/c10/b8x8_stacked.json
python bench.py --batch 16 --aggregate-k 4 --aggregate-stacked --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown 2>/dev/null | tail -1 > gpurun_out/c10/b16x4_stacked.json
python bench.py --batch 8 --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown 2>/dev/null | tail -1 > gpurun_out/c10/b8_plain.json
python bench.py --batch 100 --streams 2 --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown 2>/dev/null | tail -1 > gpurun_out/c10/b100_uniform.json
python bench.py --batch 100 --streams 2 --steps 5 --warmup 2 --tune 6=0 --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown 2>/dev/null | tail -1 > gpurun_out/c10/b100_uniform_allrows.json
for f in gpurun_out/c10/b*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['config'].get('mean_sep'))"; done; tail -3 gpurun_out/c10/b4_stacked.err
