# round 6, GPU call 5: where the key centring's 25 us go (kernel trace of the projection alone), after the 32-bit row index
mkdir -p gpurun_out/r06c5
O=gpurun_out/r06c5
python tools/bench_qkv.py 2>/dev/null | tee $O/bench_qkv.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o qkv -- python $GRAFT_REPO_ROOT/tools/bench_qkv.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r06c5/trace/**/*kernel_stats.csv', recursive=True)
for r in list(csv.DictReader(open(f[0])))[:12]:
    print(r['Name'][:110], r['Calls'], float(r['AverageNs']) / 1e3)
PY
