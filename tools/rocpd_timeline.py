"""Two-stream view of a rocprofv3 (rocpd) trace of bench.py: per training step, how long the main
queue is busy / idle and where the GP-prior chain of the side stream sits relative to it.
    python tools/rocpd_timeline.py gpurun_out/prof/bench_results.db"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute('select name, stream_id, start, end from kernels order by start').fetchall()
    gp_q = {q for n, q, s, e in rows if 'gp_' in n}
    adam = [(s, e) for n, q, s, e in rows if 'clip_adam' in n]
    print('queues:', sorted({q for _, q, _, _ in rows}), 'gp queues:', sorted(gp_q), 'steps:', len(adam))
    for i in range(max(1, len(adam) - 6), len(adam)):
        t0, t1 = adam[i - 1][1], adam[i][1]          # one step: end of previous optimizer step -> end of this one
        main = [(s, e) for n, q, s, e in rows if q not in gp_q and s >= t0 and e <= t1]
        gp = [(n, s, e) for n, q, s, e in rows if q in gp_q and 'gp_' in n and e > t0 and s < t1]
        busy = sum(e - s for s, e in main)
        # idle gaps on the main queue
        gaps = []
        prev = t0
        for s, e in main:
            if s - prev > 20000:
                gaps.append(((prev - t0) / 1e3, (s - prev) / 1e3))
            prev = max(prev, e)
        line = f'step {i}: {(t1 - t0) / 1e6:.3f} ms, main busy {busy / 1e6:.3f} ms, {len(main)} kernels'
        if gp:
            line += f'; gp chain {len(gp)} kernels from {(gp[0][1] - t0) / 1e6:+.3f} to {(gp[-1][2] - t0) / 1e6:+.3f} ms, busy {sum(e - s for _, s, e in gp) / 1e6:.3f} ms'
        print(line)
        if gaps:
            print('   main-queue gaps > 20 us (at us, length us):', [(round(a), round(b)) for a, b in gaps[:12]])


if __name__ == '__main__':
    main()
