# round 6, GPU call 22: the two-rank-on-one-device DP test repeated (it failed once in a full-suite run at the end of round 6): what fails when it fails
mkdir -p gpurun_out/r06c22
O=gpurun_out/r06c22
for i in $(seq 1 14); do
  timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "data_parallel_gradient_equals_global_batch" > $O/run_$i.log 2>&1
  echo "run $i rc $?" | tee -a $O/summary.txt
  grep -E "Error|assert|werr|err" $O/run_$i.log | head -8 >> $O/summary.txt
done
