# round-5 experiment: the 256 x 256 NT GEMM fed by a four-stage ring (PFN_TUNE_GEMM_NT_KERNEL = 4) against the default (2 stages of 64, mode 2)
mkdir -p gpurun_out/c6
python -m pytest tests/test_gpu_ops.py -q -x -k "gemm_nt_big" 2>&1 | tail -4 > gpurun_out/c6/ops.log
for rep in 1 2; do
  python tools/bench_gemm_epi.py --batch 32 --modes 2,4 > gpurun_out/c6/epi_$rep.txt 2>&1
done
for rep in 1 2 3; do
  for mode in 0 4; do
    T=""; [ $mode = 4 ] && T="--tune 0=4"
    python bench.py --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown $T 2>/dev/null | tail -1 > gpurun_out/c6/step_mode${mode}_$rep.json
  done
done
cat gpurun_out/c6/ops.log; grep -v amdgpu gpurun_out/c6/epi_1.txt
for f in gpurun_out/c6/step_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done
