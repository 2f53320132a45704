"""Yardstick, not product: what the vendor libraries that ship with PyTorch-ROCm (hipBLASLt / rocBLAS behind torch.matmul, the flash kernel behind
F.scaled_dot_product_attention) take for the PLAIN forms of this stack's hot shapes on the same MI355X, next to the hand-written kernels' figures of
bench.py's `kernels` table.  The library calls carry no epilogue (no bias / GELU / residual / LayerNorm / second output) and the attention is dense
(every key for every query, no eval-position mask): they bound what a kernel of that shape can reach here, they are not a replacement.

    python tools/bench_vendor.py [--tokens 64000] > gpurun_out/vendor.json
"""
import argparse, json, sys, time
import torch


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tokens', type=int, default=64000)
    a = ap.parse_args()
    dev = 'cuda:0'
    M = a.tokens
    out = dict(device=torch.cuda.get_device_name(0), torch=torch.__version__, tokens=M, gemms=[], attention={})
    g = torch.Generator(device=dev).manual_seed(0)
    for name, N, K in [('qkv', 1536, 512), ('out_proj', 512, 512), ('linear1', 1024, 512), ('linear2', 512, 1024), ('d(hpre)', 1024, 512),
                       ('dy1 (dh.W1)', 512, 1024), ('dx (dqkv.Win)', 512, 1536), ('d(ctx)', 512, 512)]:
        A = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).bfloat16()
        W = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).bfloat16()
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        us = timeit(lambda: torch.matmul(A, W.t(), out=C))
        out['gemms'].append(dict(gemm=name, M=M, N=N, K=K, us=us, tflops=2.0 * M * N * K / us / 1e6,
                                 hbm_floor_us=(M * K + N * K + M * N) * 2 / 6.3e12 * 1e6))
    # what the same library sustains on large cubes (uniform random operands): the practical bf16 matrix-core ceiling of THIS box under its power limit
    for n in (4096, 8192):
        A = (torch.rand(n, n, device=dev, generator=g) * 2 - 1).bfloat16()
        W = (torch.rand(n, n, device=dev, generator=g) * 2 - 1).bfloat16()
        C = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
        us = timeit(lambda: torch.matmul(A, W.t(), out=C), iters=30 if n == 4096 else 10)
        out['gemms'].append(dict(gemm=f'cube {n}^3 (NT, bf16 out)', M=n, N=n, K=n, us=us, tflops=2.0 * n ** 3 / us / 1e6))
    # weight-gradient form: C[P, Q] = A[M, P]^T B[M, Q]
    for name, P, Q in [('dW_qkv', 1536, 512), ('dW_lin1', 1024, 512), ('dW_lin2', 512, 1024), ('dW_out', 512, 512)]:
        A = (torch.rand(M, P, device=dev, generator=g) * 2 - 1).bfloat16()
        B = (torch.rand(M, Q, device=dev, generator=g) * 2 - 1).bfloat16()
        us = timeit(lambda: torch.matmul(A.t(), B))
        out['gemms'].append(dict(gemm=name + ' (TN, bf16 out)', M=M, N=P, K=Q, us=us, tflops=2.0 * M * P * Q / us / 1e6))
    # dense attention, the micro-batch shape of the bench: B 32, H 4, S 2000, D 128
    import torch.nn.functional as F
    Bn, H, S, D = 32, 4, 2000, 128
    q, k, v = [(torch.randn(Bn, H, S, D, device=dev, generator=g) * 0.5).bfloat16().requires_grad_(True) for _ in range(3)]
    try:
        fwd = timeit(lambda: F.scaled_dot_product_attention(q, k, v), iters=10, warm=3)
        o = F.scaled_dot_product_attention(q, k, v)
        do = torch.randn_like(o)
        both = timeit(lambda: torch.autograd.grad(F.scaled_dot_product_attention(q, k, v), (q, k, v), do), iters=10, warm=3)
        unit = 2.0 * Bn * H * S * S * D        # one S x S x D product
        out['attention'] = dict(shape=[Bn, H, S, D], mask='none (dense)', fwd_us=fwd, fwd_tflops=2 * unit / fwd / 1e6, fwd_plus_bwd_us=both,
                                bwd_us=both - fwd, bwd_tflops_4_units=4 * unit / (both - fwd) / 1e6,
                                note='dense: 2 product units forward, 4 algorithmic units backward (S^2 pairs; this stack\'s masked kernels count S * sep + (S - sep))')
    except Exception as e:      # (no flash kernel for this build / shape)
        out['attention'] = dict(error=repr(e))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
