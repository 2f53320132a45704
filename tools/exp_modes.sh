# round 5: what the optional modes cost -- the deterministic schedule at configs[1], exact-f32 training at configs[1] and at configs[4]'s width (head dim 256: plain attention backward)
mkdir -p gpurun_out/c18
F="--no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown"
python bench.py --steps 10 --warmup 3 $F 2>/dev/null | tail -1 > gpurun_out/c18/default.json
PFN_BENCH_DETERMINISTIC=1 python bench.py --steps 10 --warmup 3 $F 2>gpurun_out/c18/det.err | tail -1 > gpurun_out/c18/deterministic.json
python bench.py --steps 5 --warmup 2 --precision f32 --batch 16 $F 2>gpurun_out/c18/f32.err | tail -1 > gpurun_out/c18/f32_config2.json
timeout 600 python bench.py --config 5 --steps 3 --warmup 1 --precision f32 --batch 4 $F 2>gpurun_out/c18/f32c5.err | tail -1 > gpurun_out/c18/f32_config5.json
for f in gpurun_out/c18/*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['dtype'], d['config'].get('per_gpu_batch'))"; done; tail -3 gpurun_out/c18/*.err
