"""CPU emulation of the GP sampler's blocked Cholesky (256-wide outer blocks, f32 everywhere) with the rank-256 trailing update computed four ways:
exact f32, three bf16 terms / six products (what gp_syrk_planes_kernel does), two fp16 terms / three products (hi.hi + hi.lo + lo.hi on a per-dataset
power-of-two scale), two bf16 terms / three products.  Prints the error of y = L z against the f64 factorisation -- prices a cheaper split BEFORE a kernel
is written for it.        python tools/sim_gp_split.py [n_datasets]"""
import sys
import torch

torch.manual_seed(0)

OB = 256


def split(x, dtype, terms):
    out, r = [], x.clone()
    for _ in range(terms):
        t = r.to(dtype).float()
        out.append(t)
        r = r - t
    return out


def update(X, mode):
    """X [m, 256] f32 -> X X^T the way the update computes it (products exact, f32 accumulation)."""
    if mode == 'f32':
        return X @ X.t()
    if mode == 'bf16x6':
        h, m, l = split(X, torch.bfloat16, 3)
        return h @ l.t() + l @ h.t() + m @ m.t() + m @ h.t() + h @ m.t() + h @ h.t()
    if mode == 'bf16x3':
        h, m = split(X, torch.bfloat16, 2)
        return m @ h.t() + h @ m.t() + h @ h.t()
    if mode in ('fp16x3', 'fp16x4'):
        s = 2.0 ** (12 - torch.ceil(torch.log2(X.abs().max())).item())          # |X s| < 2^12: products < 2^24, lo terms stay normal down to 2^-26 of the maximum
        h, l = split(X * s, torch.float16, 2)
        p = l @ h.t() + h @ l.t() + h @ h.t()
        if mode == 'fp16x4':
            p = p + l @ l.t()
        return p / (s * s)
    raise ValueError(mode)


def prod(A, B, mode, scale):
    """A B^T the way a split product computes it (A, B f32; products exact, f32 accumulation)."""
    if mode == 'f32':
        return A @ B.t()
    ah, al = split(A * scale, torch.float16, 2)
    bh, bl = split(B * scale, torch.float16, 2)
    return (al @ bh.t() + ah @ bl.t() + ah @ bh.t()) / (scale * scale)


def solve_wide(Ld, A, mode, scale):
    """X = A Ld^-T as gp_trsm_wide_kernel does it: 64-wide blocks, V_j -= sum_{i<j} X_i L_ji^T (`mode` products), X_j = V_j L_jj^-T with the explicit f32 inverse."""
    n = Ld.shape[0]
    X = A.clone()
    for j0 in range(0, n, 64):
        j1 = min(n, j0 + 64)
        if j0 > 0:
            X[:, j0:j1] -= prod(X[:, :j0], Ld[j0:j1, :j0], mode, scale)
        inv = torch.linalg.solve_triangular(Ld[j0:j1, j0:j1], torch.eye(j1 - j0), upper=False)
        X[:, j0:j1] = X[:, j0:j1] @ inv.t()
    return X


def chol_blocked(K, z, mode, solve_mode=None):
    K = K.clone()
    n = K.shape[0]
    bad = 0
    for k0 in range(0, n, OB):
        k1 = min(n, k0 + OB)
        try:
            Ld = torch.linalg.cholesky(K[k0:k1, k0:k1])
        except Exception:
            return None, 1
        K[k0:k1, k0:k1] = Ld
        if k1 < n:
            if solve_mode is None:
                X = torch.linalg.solve_triangular(Ld, K[k1:, k0:k1].t(), upper=False).t().contiguous()
            else:
                scale = 2.0 ** (14 - torch.ceil(0.5 * torch.log2(K.diagonal().max())).item())       # sqrt(K_ii) bounds every entry of the factor (the diagonal only shrinks)
                X = solve_wide(Ld, K[k1:, k0:k1].contiguous(), solve_mode, scale)
            K[k1:, k0:k1] = X
            K[k1:, k1:] -= update(X, mode)
    L = torch.tril(K)
    return L @ z, bad


def gram(x, ls, os_, noise, kernel):
    d2 = torch.cdist(x / ls, x / ls).double() ** 2
    if kernel == 'rbf':
        k = torch.exp(-0.5 * d2)
    else:
        r = torch.sqrt(5 * d2)
        k = (1 + r + r * r / 3) * torch.exp(-r)
    return os_ * k + noise * torch.eye(x.shape[0], dtype=torch.float64)


MODES = ('f32', 'bf16x6', 'fp16x3', 'fp16x4', 'bf16x3', 'fp16x3+solve-f32', 'fp16x3+solve-fp16x3')


def errors(T, F, noise, os_, ls, kernel, modes=MODES):
    """One dataset: relative L2 error of y = L z against the f64 factorisation for every mode (nan = the factorisation failed)."""
    x = torch.rand(T, F, dtype=torch.float64)
    z = torch.randn(T, dtype=torch.float64)
    K64 = gram(x, ls, os_, noise, kernel)
    want = torch.linalg.cholesky(K64) @ z
    K32 = K64.float()
    out = {}
    for mode in modes:
        if '+' in mode:
            y, bad = chol_blocked(K32, z.float(), 'fp16x3', 'f32' if mode.endswith('solve-f32') else 'fp16x3')
        else:
            y, bad = chol_blocked(K32, z.float(), mode)
        out[mode] = float('nan') if y is None else ((y.double() - want).norm() / want.norm()).item()
    return out


if __name__ == '__main__':
    nd = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    for (T, F, noise, os_, ls, kernel) in [(2000, 5, 1e-4, 1.0, 0.6, 'rbf'), (2000, 18, 1e-4, 1.0, 0.6, 'rbf'), (2000, 10, 1e-3, 1.0, 0.5, 'matern'), (2000, 5, 1e-4, 30.0, 0.6, 'rbf'), (2000, 5, 1e-4, 0.02, 0.6, 'rbf')]:
        res = {}
        for d in range(nd):
            for m, e in errors(T, F, noise, os_, ls, kernel).items():
                res.setdefault(m, []).append(e)
        print(f'T={T} F={F} {kernel} noise={noise} outputscale={os_}: rel. L2 error of y vs f64   ' +
              '   '.join(f'{m} ' + '/'.join(f'{e:.2e}' for e in v) for m, v in res.items()), flush=True)
