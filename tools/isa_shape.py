"""Condensed view of a kernel's ISA: one character per instruction (M mfma, E v_exp/log/rcp (quarter rate), v VALU, L LDS, G global/buffer,
s scalar, W s_waitcnt, B s_barrier, X branch), basic blocks on their own lines -- shows at a glance whether vector work is interleaved
with the MFMAs of a block or stranded behind them.

    python tools/isa_shape.py /tmp/attention.s _ZN3pfn15attn_fwd_kernelIDF16bLi128EEEvNS_8AttnArgsE [--min 40]
"""
import re
import sys


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfmac'): return 'M'
    if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)_', op): return 'E'
    if op.startswith('v_accvgpr'): return 'a'
    if op.startswith('v_'): return 'v'
    if op.startswith('ds_'): return 'L'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'G'
    if op == 's_waitcnt': return 'W'
    if op == 's_barrier': return 'B'
    if op.startswith(('s_cbranch', 's_branch')): return 'X'
    if op == 's_nop': return 'n'
    if op.startswith('s_'): return 's'
    return '?'


def main():
    path, sym = sys.argv[1], sys.argv[2]
    min_len = int(sys.argv[sys.argv.index('--min') + 1]) if '--min' in sys.argv else 0
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ':'))
    blocks, cur, label = [], [], 'entry'
    for ln in lines[start + 1:]:
        t = ln.strip()
        if t.startswith('.Lfunc_end') or t.startswith('.section') or t.startswith('s_endpgm'):
            break
        if not t or (t.startswith((';', '.', '//')) and not re.match(r'\.LBB\d+_\d+:', t)):
            continue
        if re.match(r'\.LBB\d+_\d+:', t):
            blocks.append((label, cur)); cur, label = [], t.rstrip(':')
            continue
        op = t.split()[0]
        cur.append(classify(op))
        if op.startswith(('s_cbranch', 's_branch')):
            pass
    blocks.append((label, cur))
    for label, ops in blocks:
        if len(ops) < min_len:
            continue
        s = ''.join(ops)
        counts = {k: s.count(k) for k in 'MEvaLGsWB' if s.count(k)}
        print(f'{label:12s} {len(ops):5d} {counts}')
        for i in range(0, len(s), 160):
            print('    ' + s[i:i + 160])


if __name__ == '__main__':
    main()
