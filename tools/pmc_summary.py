"""Condense the outputs of tools/profile_bench.sh: the rocprofv3 kernel stats table plus per-kernel
PMC averages (FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; FETCH_SIZE is doubled for the
gfx950 under-count documented in MI355X_MICROARCH.md 'HBM')."""
import csv
import glob
import os
import sys


def find(root, pattern):
    hits = glob.glob(os.path.join(root, '**', pattern), recursive=True)
    return hits[0] if hits else None


def short(n):
    return n.replace('void ', '').replace('pfn::', '')[:64]


TOP = ' [top layer: queries >= sep]'


def top_layer_dispatches(rows, grid_key):
    """The attention launches of the TOP encoder layer skip the queries below sep (pfn_api.hip): the forward and the query-block pass then run a
    smaller grid, the delta kernel and the key-block pass the usual one.  Returns the Dispatch_Ids of a top-layer launch set: per queue (= HIP
    stream) in dispatch order, a forward / query-block launch with less than the largest grid of its symbol, and the delta and key-block launches
    in front of such a query-block launch."""
    full = {}
    for r in rows:
        n = r['Kernel_Name']
        if 'attn_fwd_kernel' in n or 'attn_bwd_dq_kernel' in n:
            full[n] = max(full.get(n, 0), int(r[grid_key]))
    top, pending = set(), {}
    for r in sorted(rows, key=lambda r: int(r['Dispatch_Id'])):
        n, q = r['Kernel_Name'], r['Queue_Id']
        if 'attn_fwd_kernel' in n and int(r[grid_key]) < full[n]:
            top.add(r['Dispatch_Id'])
        elif 'attn_delta_kernel' in n or 'attn_bwd_kv_kernel' in n:
            pending.setdefault(q, []).append(r['Dispatch_Id'])
        elif 'attn_bwd_dq_kernel' in n:
            if int(r[grid_key]) < full[n]:
                top.add(r['Dispatch_Id'])
                top.update(pending.get(q, []))
            pending[q] = []
    return top


def main():
    import json
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = sys.argv[1]
    traffic = {}
    # what was profiled: the bench line of the kernel-trace run (tools/profile_bench.sh puts it next to the traces)
    meta = {}
    try:
        line = json.load(open(os.path.join(out, 'bench.json')))
        meta = dict(config=line['config']['baseline_config'], batch=line['config']['per_gpu_batch'], streams=line['config']['micro_batch_streams'], precision=line['dtype'],
                    workload=line['config']['workload'], datasets_per_s_under_the_profiler=line['value'])
    except Exception as e:      # (older layouts: the defaults of configs[1])
        meta = dict(config=2, batch=int(sys.argv[2]) if len(sys.argv) > 2 else 64, streams=2, precision='bf16', note_meta=f'bench.json unreadable: {e}')
    try:
        import torch  # noqa: F401
        from transformerscandobayesianinference_amd import _hip
        meta['abi'] = _hip.ABI_VERSION
    except Exception:
        pass
    st = find(os.path.join(out, 'stats'), '*kernel_stats.csv')
    if st:
        print('== rocprofv3 --kernel-trace --stats: bench.py --steps 10 --warmup 3 ==')
        rows = list(csv.DictReader(open(st)))
        print(f'{"kernel":66s} {"calls":>7s} {"total_ms":>10s} {"avg_us":>10s} {"pct":>6s}')
        for r in rows[:28]:
            print(f'{short(r["Name"]):66s} {int(r["Calls"]):7d} {float(r["TotalDurationNs"]) / 1e6:10.3f} {float(r["AverageNs"]) / 1e3:10.2f} {float(r["Percentage"]):6.2f}')
    tr = find(os.path.join(out, 'stats'), '*kernel_trace.csv')
    if tr:
        # the attention kernels' in-step durations, the top layer's short launches (queries >= sep only) apart from the others
        rows = [r for r in csv.DictReader(open(tr)) if 'pfn' in r['Kernel_Name']]      # every kernel of the library (the dominant one is a GEMM in some configurations)
        top = top_layer_dispatches([r for r in rows if 'attn_' in r['Kernel_Name']], 'Grid_Size_X')
        agg = {}
        for r in rows:
            a = agg.setdefault((r['Kernel_Name'], r['Dispatch_Id'] in top), [0, 0.0])
            a[0] += 1
            a[1] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
        print('== kernels inside the step (kernel trace of the same run), the attention\'s top-layer launches apart ==')
        in_step = {}
        for (name, is_top), (n, ns) in sorted(agg.items()):
            if 'attn_' in name or ns / 1e6 > 1.0:
                print(f'{short(name) + (TOP if is_top else ""):96s} n={n:5d} avg_us={ns / n / 1e3:10.2f}')
            in_step[name + (TOP if is_top else '')] = {'calls': n, 'avg_us': ns / n / 1e3}
        json.dump({'note': 'rocprofv3 --kernel-trace of bench.py --steps 10 --warmup 3 (tools/profile_bench.sh): average in-step duration per attention kernel; the top '
                           'encoder layer\'s launches (queries >= sep only, told by the grid of the forward / query-block pass) are listed apart', **meta, 'kernels': in_step},
                  open(os.path.join(out, 'in_step_attention.json'), 'w'), indent=1)
    for d in sorted(glob.glob(os.path.join(out, 'pmc_*'))):
        if not os.path.isdir(d):
            continue
        cc = find(d, '*counter_collection.csv')
        if not cc:
            print('no counter csv under', d)
            continue
        agg = {}
        crow = list(csv.DictReader(open(cc)))
        top = top_layer_dispatches([r for r in crow if 'attn_' in r['Kernel_Name']], 'Grid_Size')
        for r in crow:
            k = (short(r['Kernel_Name']) + (TOP if r['Dispatch_Id'] in top else ''), r['Counter_Name'])
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r['Counter_Value'])
        print(f'== PMC pass {os.path.basename(d)} (average per launch) ==')
        for (k, c), (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
            avg = v / n
            extra = ''
            if c == 'FETCH_SIZE':
                traffic.setdefault(k, {})['read_bytes'] = 2 * avg * 1024
                extra = f'  -> {2 * avg / 1024:10.1f} MB/launch read (x2 gfx950 correction)'
            if c == 'WRITE_SIZE':
                traffic.setdefault(k, {})['write_bytes'] = avg * 1024
                extra = f'  -> {avg / 1024:10.1f} MB/launch written (uncalibrated)'
            print(f'{k:66s} {c:28s} n={n:5d} avg={avg:16.1f}{extra}')
    json.dump({'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB) per launch, bench.py --steps 3 --warmup 1 of the configuration below; FETCH_SIZE doubled per MI355X_MICROARCH.md '
                       '(gfx950 counts 128-B requests at 64 B), WRITE_SIZE as reported', **meta, 'kernels': traffic}, open(os.path.join(out, 'pmc_traffic.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
