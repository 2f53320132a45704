# end-of-round measurements (round 5): profiles of the bench command, the whole GPU suite with every asserted tolerance recorded, the three bench lines
TAG=${1:-r05prof}
bash tools/profile_bench.sh $TAG > gpurun_out/${TAG}_run.log 2>&1
PFN_RECORD_BOUNDS=gpurun_out/r05_parity_measured.json python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r05_pytest.log
for cfg in 2 4 5; do
  python bench.py --config $cfg > gpurun_out/r05_bench_config$cfg.line 2> gpurun_out/r05_bench_config$cfg.err
  cp bench_detail.json gpurun_out/r05_bench_config$cfg.json
done
tail -3 gpurun_out/r05_pytest.log; head -c 600 gpurun_out/r05_bench_config2.line; echo; head -40 gpurun_out/$TAG/summary.txt
