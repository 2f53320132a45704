"""Idle time in a rocprofv3 kernel trace of bench.py: per optimizer step, the union of all queues' busy intervals against the
step's wall time, the largest gaps (with the kernels either side), and time per queue.
    python tools/trace_gaps.py gpurun_out/<tag>/stats/bench_kernel_trace.csv [last_n_steps]"""
import csv
import sys


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']))
    rows.sort()
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    adam = [(s, e) for s, e, q, n in rows if 'clip_adam' in n]
    print(f'{len(rows)} kernels, {len(adam)} optimizer steps, queues {sorted({q for _, _, q, _ in rows})}')
    for i in range(max(1, len(adam) - last), len(adam)):
        t0, t1 = adam[i - 1][1], adam[i][1]
        ks = [(s, e, q, n) for s, e, q, n in rows if e > t0 and s < t1]
        busy, cur_s, cur_e, gaps, prev_name = 0, None, None, [], '(step start)'
        for s, e, q, n in ks:
            s, e = max(s, t0), min(e, t1)
            if cur_e is None:
                if s - t0 > 0: gaps.append((s - t0, prev_name, n))
                cur_s, cur_e = s, e
            elif s > cur_e:
                busy += cur_e - cur_s
                gaps.append((s - cur_e, prev_name, n))
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
            if e >= cur_e: prev_name = n
        busy += cur_e - cur_s
        per_q = {}
        for s, e, q, n in ks:
            per_q[q] = per_q.get(q, 0) + min(e, t1) - max(s, t0)
        idle = (t1 - t0) - busy
        print(f'step {i}: {(t1 - t0) / 1e6:.3f} ms wall, some kernel running {busy / 1e6:.3f} ms, idle {idle / 1e6:.3f} ms ({100 * idle / (t1 - t0):.1f} %), '
              f'{len(ks)} kernels, {len(gaps)} gaps; busy per queue ' + ', '.join(f'{q}: {v / 1e6:.2f}' for q, v in sorted(per_q.items())))
        gaps.sort(reverse=True)
        for g, a, b in gaps[:6]:
            print(f'     {g / 1e3:7.1f} us between {a[:50]} -> {b[:50]}')
        small = sum(g for g, _, _ in gaps if g < 20000)
        print(f'     gaps < 20 us: {sum(1 for g, _, _ in gaps if g < 20000)} totalling {small / 1e3:.1f} us')


if __name__ == '__main__':
    main()
