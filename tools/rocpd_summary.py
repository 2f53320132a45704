"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average, restricted
to the last `--steps` benchmark steps when --marker is given.  Usage:
    python tools/rocpd_summary.py gpurun_out/prof1/bench_results.db [--last-fraction 0.7]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*', '', name)
    name = name.replace('void ', '').replace('pfn::', '')
    name = re.sub(r'at::native::', 'torch::', name)
    return name[:70]


def main():
    path = sys.argv[1]
    frac = float(sys.argv[sys.argv.index('--last-fraction') + 1]) if '--last-fraction' in sys.argv else 1.0
    db = sqlite3.connect(path)
    rows = db.execute('select name, start, end from kernels order by start').fetchall()
    if not rows:
        print('no kernels')
        return
    t0, t1 = rows[0][1], rows[-1][2]
    cut = t1 - (t1 - t0) * frac
    agg = {}
    busy = 0
    for name, s, e in rows:
        if s < cut:
            continue
        k = short(name)
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += e - s
        busy += e - s
    span = t1 - max(cut, t0)
    print(f'window {span / 1e6:.3f} ms, kernel busy {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %), {sum(a[0] for a in agg.values())} dispatches')
    print(f'{"kernel":72s} {"calls":>7s} {"total_ms":>10s} {"avg_us":>10s} {"%busy":>7s}')
    for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{k:72s} {n:7d} {d / 1e6:10.3f} {d / n / 1e3:10.2f} {100 * d / busy:7.2f}')


if __name__ == '__main__':
    main()
