"""Per-kernel averages of a rocprofv3 counter_collection.csv:  python tools/pmc_kernels.py <csv> [filter]"""
import csv, sys
agg = {}
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for r in csv.DictReader(open(sys.argv[1])):
    if flt not in r['Kernel_Name']:
        continue
    k = (r['Kernel_Name'].replace('void ', '').replace('pfn::', '')[:48], r['Counter_Name'])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
for (k, c), (n, v) in sorted(agg.items()):
    print(f'{k:50s} {c:28s} n={n:3d} avg={v / n:16.1f}')
