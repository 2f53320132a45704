"""Sets the fp16 argument of every `tol3(precision, f32, bf16)` bound in tests/test_gpu_parity.py to 2 x the largest value a recorded GPU session measured for it
(PFN_RECORD_BOUNDS=... pytest -m gpu  ->  profiles/r06_parity_measured.json), never below the exact-f32 bound and never above the default (a quarter of the bf16 bound).
A bound that already carries an fp16 value is recomputed unless its line has a comment (set by hand).

    python tools/tighten_fp16_bounds.py profiles/r06_parity_measured.json [more records ...] [--write]
"""
import json, math, re, sys

def round_up(x, digits=2):
    e = math.floor(math.log10(x)) - (digits - 1)
    return math.ceil(x / 10 ** e) * 10 ** e

def main():
    rec = {}
    for f in [a for a in sys.argv[1:] if not a.startswith('--')]:      # several sessions: the largest value any of them saw
        for k, v in json.load(open(f)).items():
            if k not in rec or v['max'] > rec[k]['max']:
                rec[k] = v
    write = '--write' in sys.argv
    path = 'tests/test_gpu_parity.py'
    lines = open(path).read().split('\n')
    test = None
    changed = 0
    for i, line in enumerate(lines):
        m = re.match(r'def (test_\w+)\(', line)
        if m:
            test = m.group(1)
        m = re.search(r"within\(f'\{precision\} (.*?)', .*tol3\(precision, ([0-9.e-]+), ([0-9.e-]+)(, [0-9.e-]+)?\)\)", line)
        if not m or test is None or '#' in line:      # (a bound with a comment was set by hand)
            continue
        label, f32, bf16 = m.group(1), float(m.group(2)), float(m.group(3))
        pat = re.compile(re.escape(test) + r'(\[.*\])? :: fp16 ' + re.sub(r'\\\{.*?\\\}', '.*?', re.escape(label)) + '$')
        vals = [v['max'] for k, v in rec.items() if pat.match(k)]
        if not vals:
            print(f'{test}: no fp16 record for "{label}"')
            continue
        default = max(f32, bf16 / 4)
        new = min(default, max(f32, round_up(2 * max(vals)) if max(vals) > 0 else f32))
        if new >= default and not m.group(4):
            continue
        s = ('%.1e' % new).replace('e-0', 'e-')
        lines[i] = line.replace(f'tol3(precision, {m.group(2)}, {m.group(3)}{m.group(4) or ""})', f'tol3(precision, {m.group(2)}, {m.group(3)}, {s})' if new < default else f'tol3(precision, {m.group(2)}, {m.group(3)})')
        print(f'{test}: "{label}"  measured {max(vals):.2e} (n={len(vals)})  default {default:.1e} -> {s}')
        changed += 1
    print(changed, 'bounds tightened')
    if write:
        open(path, 'w').write('\n'.join(lines))

main()
