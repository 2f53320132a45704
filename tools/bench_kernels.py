"""Per-kernel timings at the north-star shape through the single-op C ABI (run on the GPU box):
    python tools/bench_kernels.py [--batch 16] [--only gemm|attn|all]
Prints one line per kernel: average launch duration (HIP events on the launch stream) and TFLOP/s."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from transformerscandobayesianinference_amd import hipops  # noqa: E402
from transformerscandobayesianinference_amd import _hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--sep', type=int, default=1755)
    ap.add_argument('--gemm-modes', default='1,2')
    ap.add_argument('--lib', default=None, help='alternative build of libpfn_hip.so (experiment variants)')
    args = ap.parse_args()
    if args.lib:
        _hip.LIB_PATH = os.path.abspath(args.lib)
    for mode in [int(m) for m in args.gemm_modes.split(',')]:
        _hip.check(_hip.lib().pfn_set_tuning(0, mode), 'tuning')
        for k in bench.kernel_breakdown(args.batch, args.sep):
            print(f"mode{mode} {k['kernel']:48s} {k['seconds'] * 1e6:9.1f} us {k['tflops']:8.1f} TF/s  x{k['launches_per_step']} = {k['step_seconds'] * 1e3:.3f} ms")
    _hip.lib().pfn_set_tuning(0, 0)
    # grouped weight gradients: one layer (4 problems) and the whole stack (24 problems)
    w = bench.WORKLOAD
    E, F, L = w['emsize'], w['nhid'], w['nlayers']
    M = args.batch * w['bptt']
    dev = torch.device('cuda')
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
    def layer_problems():
        return [(r(M, E), r(M, F), torch.zeros(E, F, device=dev), None),
                (r(M, F), r(M, E), torch.zeros(F, E, device=dev), torch.zeros(F, device=dev)),
                (r(M, E), r(M, E), torch.zeros(E, E, device=dev), None),
                (r(M, 3 * E), r(M, E), torch.zeros(3 * E, E, device=dev), torch.zeros(3 * E, device=dev))]
    flops_layer = 2.0 * M * (E * F * 2 + E * E + 3 * E * E)
    one = layer_problems()
    for splits in (0, 4, 8, 16):
        t = bench.time_kernel(lambda: hipops.gemm_tn_group(one, splits))
        print(f'gemm_tn_group[1 layer, splits={splits}] {t * 1e6:9.1f} us {flops_layer / t / 1e12:8.1f} TF/s  x{L} = {t * L * 1e3:.3f} ms')
    allp = [p for _ in range(L) for p in layer_problems()]
    for wrap in (0, 512):
        _hip.lib().pfn_set_tuning(1, wrap)
        for splits in (1, 4):
            t = bench.time_kernel(lambda: hipops.gemm_tn_group(allp, splits), iters=5, warm=2)
            print(f'gemm_tn_group[{L} layers, splits={splits}, wrap={wrap}] {t * 1e6:9.1f} us {flops_layer * L / t / 1e12:8.1f} TF/s  x1 = {t * 1e3:.3f} ms')
    _hip.lib().pfn_set_tuning(1, 0)


if __name__ == '__main__':
    main()
