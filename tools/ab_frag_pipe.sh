mkdir -p gpurun_out/c4
V=transformerscandobayesianinference_amd/_variants/libpfn_nopipe.so
python -m pytest tests/test_gpu_ops.py -q -x -k "gemm" 2>&1 | tail -4 > gpurun_out/c4/ops.log
python -m pytest tests/test_gpu_parity.py -q -x -k "golden or config2_full or config4_model or top_layer or deterministic_schedule_grad" 2>&1 | tail -4 > gpurun_out/c4/parity.log
for rep in 1 2; do
  for lib in pipe nopipe; do
    L=""; [ $lib = nopipe ] && L=$V
    PFN_LIB=$L python tools/bench_gemm_epi.py --batch 32 --modes 2 > gpurun_out/c4/epi_${lib}_$rep.txt 2>&1
    PFN_LIB=$L python tools/bench_gemm_ln.py 32 > gpurun_out/c4/ln_${lib}_$rep.txt 2>&1
    PFN_LIB=$L python tools/bench_gemm_lnbwd.py --batch 32 > gpurun_out/c4/lnbwd_${lib}_$rep.txt 2>&1
    PFN_LIB=$L python tools/bench_wgrad.py --batch 32 > gpurun_out/c4/wgrad_${lib}_$rep.txt 2>&1
  done
done
for rep in 1 2 3; do
  for lib in pipe nopipe; do
    L=""; [ $lib = nopipe ] && L=$V
    PFN_LIB=$L python bench.py --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown 2>/dev/null | tail -1 > gpurun_out/c4/step_${lib}_$rep.json
  done
done
cat gpurun_out/c4/ops.log gpurun_out/c4/parity.log
for f in gpurun_out/c4/step_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done
