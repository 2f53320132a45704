# end-of-round measurements (round 6): rocprofv3 kernel-trace + PMC passes of the bench command for BASELINE configs[1], [3], [4] (tools/profile_bench.sh), installed under
# profiles/ where bench.py looks for them, then the whole GPU suite with every asserted tolerance recorded, then the three bench lines (compact line + bench_detail.json)
# and the bf16 line of configs[1].
TAG=${1:-r06}
O=gpurun_out/$TAG
mkdir -p $O
for cfg in 2 4 5; do
  extra=""; [ $cfg != 2 ] && extra="SKIP_STREAMS1=1"
  env $extra BENCH_ARGS="--config $cfg" bash tools/profile_bench.sh ${TAG}_prof_config$cfg > $O/prof_config${cfg}_run.log 2>&1
  P=gpurun_out/${TAG}_prof_config$cfg
  cp $P/pmc_traffic.json profiles/r06_pmc_traffic_config$cfg.json 2>/dev/null
  cp $P/in_step_attention.json profiles/r06_in_step_kernels_config$cfg.json 2>/dev/null
  cp $P/summary.txt $O/profile_summary_config$cfg.txt 2>/dev/null
  find $P -name "*kernel_stats.csv" -path "*stats/*" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats_config$cfg.csv
  rm -rf $P/stats/*kernel_trace.csv $P/stats1/*kernel_trace.csv $P/pmc_*/*counter_collection.csv $P/pmc_*/*kernel_trace.csv     # (raw traces: tens of MB)
done
PFN_RECORD_BOUNDS=$O/parity_measured.json python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest.log
for cfg in 2 4 5; do
  python bench.py --config $cfg > $O/bench_config$cfg.line 2> $O/bench_config$cfg.err
  cp bench_detail.json $O/bench_config$cfg.json
done
python bench.py --precision bf16 --no-extras > $O/bench_config2_bf16.line 2> $O/bench_config2_bf16.err
cp bench_detail.json $O/bench_config2_bf16.json
cp profiles/r06_pmc_traffic_config*.json profiles/r06_in_step_kernels_config*.json $O/ 2>/dev/null
tail -3 $O/pytest.log; for cfg in 2 4 5; do head -c 700 $O/bench_config$cfg.line; echo; done; head -c 400 $O/bench_config2_bf16.line; echo; head -30 $O/profile_summary_config2.txt
