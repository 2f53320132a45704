# round 5: upper bound of what removing the dS^T round trip could buy in the step (ablation build: no dS^T stores in the key-block pass, no dS^T fetches in the query-block pass; garbage results)
mkdir -p gpurun_out/c20
V=transformerscandobayesianinference_amd/_variants/libpfn_nods.so
for rep in 1 2 3; do
  for lib in base nods; do
    L=""; [ $lib = nods ] && L=$V
    PFN_LIB=$L python bench.py --no-extras --no-cpu-baseline --no-parity --no-kernel-breakdown 2>/dev/null | tail -1 > gpurun_out/c20/step_${lib}_$rep.json
  done
done
PFN_LIB=$V python tools/bench_attn.py > gpurun_out/c20/attn_nods.txt 2>&1
python tools/bench_attn.py > gpurun_out/c20/attn_base.txt 2>&1
for f in gpurun_out/c20/step_*.json; do echo $f; python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'])"; done
grep -v amdgpu gpurun_out/c20/attn_base.txt | tail -8; grep -v amdgpu gpurun_out/c20/attn_nods.txt | tail -8
