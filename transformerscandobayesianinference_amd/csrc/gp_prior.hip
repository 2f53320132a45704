// Batched Gaussian-process prior sampler (gfx950):  y_b = chol(K_b) z_b,
//   K_b = outputscale_b * k(x_b, x_b; lengthscale_b) + noise_b * I,   k = RBF or Matern-5/2 (ARD).
//
// Replaces the gpytorch ExactGP prior-mode draw of the reference (priors/fast_gp.py:41-58:
// ScaleKernel(RBFKernel) + GaussianLikelihood, sample = MultivariateNormal.sample()) and the
// sampling half of priors/fast_gp_mix.py:85-99 (Matern nu=2.5, ARD lengthscales, per-dataset
// hyper-parameters).  Everything is f32 like the reference.
//
// Schedule per call (all datasets of the batch advance together, grid.y / grid.z = dataset):
//   rng    : Philox4x32-10 -> x ~ U[0,1), z ~ N(0,1) (Box-Muller); y = 0
//   gram   : lower-triangular 64x64 tiles of K                      (HBM-bound, 4 S^2/2 bytes written)
//   two-level right-looking blocked Cholesky, S/256 outer blocks of four 64-wide panels:
//     inside the 256x256 diagonal block (narrow launches, <= 4B workgroups):
//       potrf : factor a 64x64 diagonal block in the registers of one wave, y[blk] += L_kk z[blk]
//       trsm  : one thread per row of the diagonal block below it: row <- row . L_kk^-T; y[row] += row . z[blk]
//       syrk  : rank-64 update of the rest of the diagonal block (exact-f32 MFMA, v_mfma_f32_32x32x2_f32)
//     once per outer block (wide launches):
//       trsm_wide : every row below: X = A L_d^-T against the whole 256-wide factor (MFMA + substitution), y += X z
//       syrk      : rank-256 update of the trailing matrix
// so L.z (the "TRMV") costs no extra pass over the matrix.  Algorithmic work per dataset:
// S^3/3 flop (Cholesky) + S^2 (nf+2) (Gram); the matrix (4 S^2 bytes) lives in K_ws.
#include <algorithm>
#include "pfn_device.h"
#include "pfn_kernels.h"

namespace pfn {

constexpr int NB = 64;  // panel width

// ---- split-fp16 operands (round 5; what they are and why: gp_syrk_planes_kernel) ----
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
constexpr int PL_KC = 16;                           // panel columns per slab / per LDS stage
constexpr int PL_NKC = 256 / PL_KC;                 // slabs per plane of a 256-wide panel
constexpr int PL_ROWB = PL_KC * 2;                  // bytes per slab row
constexpr int PL_NPL = 2;                           // planes: hi, lo
constexpr int PL_NSET = 2;                          // plane sets per dataset: the solved panels of an even and an odd outer block (delayed update, launch_gp_sample)
PFN_DEV long plane_slab_bytes(long rows_alloc) { return rows_alloc * PL_ROWB; }
PFN_DEV long plane_set_bytes(long rows_alloc) { return (long)PL_NPL * PL_NKC * plane_slab_bytes(rows_alloc); }
// byte offset of (plane, chunk, row) inside one dataset's plane scratch of `rows_alloc` rows
PFN_DEV long plane_offset(long rows_alloc, int plane, int kc, long row) { return ((long)(plane * PL_NKC + kc) * rows_alloc + row) * PL_ROWB; }
// exponent e of a dataset's plane scale s = 2^e (see above)
PFN_DEV int plane_scale_exp(const GpArgs& a, int b) {
  const float kii = a.outputscale[b] + a.noise[b];
  const int e = 14 - (int)ceilf(0.5f * __log2f(fmaxf(kii, 1e-30f)));
  return max(-60, min(60, e));
}

PFN_DEV f32x16 mma32_f16(const Frag<bf16>& a, const Frag<bf16>& b, f32x16 c) {      // the fragments carry fp16 bit patterns
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.v), __builtin_bit_cast(f16x8, b.v), c, 0, 0, 0);
}


__global__ __launch_bounds__(256) void gp_rng_kernel(GpArgs a) {
  const long nx = (long)a.B * a.S * a.nf, nz = (long)a.B * a.S;
  const long nx4 = (nx + 3) / 4, nz4 = (nz + 3) / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nx4 + nz4; i += (long)gridDim.x * 256) {
    if (i < nx4) {
      if (!a.gen_x) continue;
      const U4 r = philox4x32_10((unsigned long long)i, a.offset * 2, a.seed);
      const unsigned v[4] = {r.x, r.y, r.z, r.w};
      for (int e = 0; e < 4; ++e) if (4 * i + e < nx) a.x[4 * i + e] = u01(v[e]);
    } else {
      const long j = i - nx4;
      float n[4] = {0, 0, 0, 0};
      if (a.gen_z) {
        const U4 r = philox4x32_10((unsigned long long)j, a.offset * 2 + 1, a.seed);
        const float r0 = sqrtf(-2.f * __logf(u01_open(r.x))), r1 = sqrtf(-2.f * __logf(u01_open(r.z)));
        float s0, c0, s1, c1;
        __sincosf(6.283185307179586f * u01(r.y), &s0, &c0);
        __sincosf(6.283185307179586f * u01(r.w), &s1, &c1);
        n[0] = r0 * c0; n[1] = r0 * s0; n[2] = r1 * c1; n[3] = r1 * s1;
      }
      for (int e = 0; e < 4; ++e)
        if (4 * j + e < nz) {
          if (a.gen_z) const_cast<float*>(a.z)[4 * j + e] = n[e];
          a.y[4 * j + e] = 0.f;
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Gram matrix, lower-triangular 64x64 tiles.  blockIdx.x enumerates (ti >= tj), blockIdx.y = b.
// (Round 5: BAND tiles -- 16 rows x 256 columns, every row one contiguous KiB per store instruction, x_j staged transposed -- were built to lift the 2.1 TB/s
// this kernel writes at, and measured 2x SLOWER: 2528 vs 1280 us per 320 datasets.  profiles/r05_gp_sampler.txt)
// ---------------------------------------------------------------------------------------------
PFN_DEV void tri_decode(int t, int& ti, int& tj) {  // t = ti*(ti+1)/2 + tj, tj <= ti
  ti = (int)((sqrtf(8.f * t + 1.f) - 1.f) * 0.5f);
  while (ti * (ti + 1) / 2 > t) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  tj = t - ti * (ti + 1) / 2;
}

__global__ __launch_bounds__(256) void gp_gram_kernel(GpArgs a) {
  extern __shared__ float gs[];  // xi[64][nf+1], xj[64][nf+1], inv_ls[nf]
  const int nf = a.nf, ldx = nf + 1;
  float* xi = gs; float* xj = gs + 64 * ldx; float* ils = gs + 128 * ldx;
  int ti, tj;
  tri_decode(blockIdx.x, ti, tj);
  const int b = blockIdx.y, S = a.S;
  const float* xb = a.x + (long)b * S * nf;
  for (int i = threadIdx.x; i < 64 * nf; i += 256) {
    const int r = i / nf, f = i % nf;
    xi[r * ldx + f] = (ti * 64 + r < S) ? xb[(long)(ti * 64 + r) * nf + f] : 0.f;
    xj[r * ldx + f] = (tj * 64 + r < S) ? xb[(long)(tj * 64 + r) * nf + f] : 0.f;
  }
  for (int f = threadIdx.x; f < nf; f += 256) ils[f] = 1.f / a.lengthscale[(long)b * nf + f];
  __syncthreads();
  const int r0 = (threadIdx.x >> 4) * 4, c0 = (threadIdx.x & 15) * 4;
  float d2[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) d2[i][j] = 0.f;
  for (int f = 0; f < nf; ++f) {
    const float s = ils[f];
    float xa[4], xc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { xa[i] = xi[(r0 + i) * ldx + f] * s; xc[i] = xj[(c0 + i) * ldx + f] * s; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = xa[i] - xc[j]; d2[i][j] += d * d; }
  }
  const float os = a.outputscale[b], nz = a.noise[b];
  float* Kb = a.K + (long)b * S * S;
  // a thread's four columns are one 16-byte group: whole inside the matrix or whole outside it when S % 4 == 0 (every caller pads to that), and then
  // stored as ONE dwordx4 per row (round 5: the per-element bounds checks had kept hipcc at 4-byte stores -- 8.5 MB per dataset at 1.9 TB/s)
  const int gj0 = tj * 64 + c0;
  const bool vec = (S & 3) == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gi = ti * 64 + r0 + i;
    if (gi >= S) continue;
    f32x4 kv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gj = gj0 + j;
      float k;
      if (a.kernel == 0) k = __expf(-0.5f * d2[i][j]);
      else if (a.kernel == 1) { const float r = sqrtf(5.f * d2[i][j]); k = (1.f + r + r * r * (1.f / 3.f)) * __expf(-r); }   // Matern nu = 5/2
      else if (a.kernel == 2) { const float r = sqrtf(3.f * d2[i][j]); k = (1.f + r) * __expf(-r); }                        // nu = 3/2
      else k = __expf(-sqrtf(d2[i][j]));                                                                                     // nu = 1/2
      kv[j] = os * k + (gi == gj ? nz : 0.f);
    }
    if (vec) {
      if (gj0 < S) *reinterpret_cast<f32x4*>(Kb + (long)gi * S + gj0) = kv;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (gj0 + j < S) Kb[(long)gi * S + gj0 + j] = kv[j];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// potrf: factor the 64x64 diagonal block of panel k.  ONE wave per dataset, lane r keeps row r of
// the block in registers; column j of L is broadcast lane-by-lane with v_readlane, so the 64-step
// dependency chain runs without LDS round trips or barriers (the whole Cholesky is latency-bound on
// this chain: S/64 panels x 64 columns).
// ---------------------------------------------------------------------------------------------
PFN_DEV float lane_bcast(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

__global__ __launch_bounds__(64) void gp_potrf_kernel(GpArgs a, int k0) {
  const int b = blockIdx.x, S = a.S, nb = min(NB, S - k0), r = threadIdx.x;
  float* Kb = a.K + (long)b * S * S + (long)k0 * S + k0;
  float row[NB];
  const bool live = r < nb;
  if (nb == NB) {
#pragma unroll
    for (int c = 0; c < NB; c += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(Kb + (long)r * S + c);
      row[c] = t[0]; row[c + 1] = t[1]; row[c + 2] = t[2]; row[c + 3] = t[3];
    }
  } else {
#pragma unroll
    for (int c = 0; c < NB; ++c) row[c] = (live && c < nb) ? Kb[(long)r * S + c] : 0.f;
  }
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    if (c > r) row[c] = 0.f;                    // only the lower triangle is defined
    if (c == r && !live) row[c] = 1.f;          // padding rows of the last (partial) panel: identity
  }
  const bool solve = a.w != nullptr;
  const float zr = (live && !solve) ? a.z[(long)b * S + k0 + r] : 0.f;
  int bad = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const float piv = lane_bcast(row[j], j);
    if (!(piv > 0.f) && bad == 0) bad = k0 + j + 1;
    const float d = sqrtf(fmaxf(piv, 1e-20f));
    const float l = (r == j) ? d : row[j] / d;
    row[j] = l;
#pragma unroll
    for (int c = j + 1; c < NB; ++c) row[c] -= l * lane_bcast(l, c);
  }
  if (r == 0 && bad && a.info[b] == 0) a.info[b] = bad;
  float acc = 0.f;
  if (!solve) {
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      const float zc = lane_bcast(zr, c);
      if (c <= r) acc += row[c] * zc;
    }
  } else {
    // posterior mode: w[blk] = L_kk^-1 residual[blk] by forward substitution along the lanes (lane j's quotient is
    // the only one read in step j; padding rows of a partial panel carry a zero residual and an identity row)
    float res = live ? a.y[(long)b * S + k0 + r] : 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float wj = lane_bcast(res / row[j], j);
      if (r == j) acc = wj;
      if (r > j) res -= row[j] * wj;
    }
  }
  if (live) {
    if (nb == NB) {
      // Inverse of the block factor for the wide trsm (gp_trsm_wide_kernel multiplies by it on the MFMA instead of
      // substituting): lane c builds column c of L^-1,  inv[r] = L^-1[r][c] = -(sum_{m<r} L[r][m] L^-1[m][c]) / L[r][r]
      // (zero above the diagonal by construction), with L[r][m] broadcast from lane r.  Column c of L^-1 is row c
      // of L^-T, which is exactly the strictly-upper part of the block: the block leaves as [L \ L^-T], diagonal = L.
      float inv[NB];
#pragma unroll
      for (int rr = 0; rr < NB; ++rr) {
        float acc = 0.f;
#pragma unroll
        for (int mm = 0; mm < rr; ++mm) acc += lane_bcast(row[mm], rr) * inv[mm];
        const float d = lane_bcast(row[rr], rr);
        inv[rr] = (rr == r) ? 1.f / d : -acc / d;
      }
#pragma unroll
      for (int c = 0; c < NB; c += 4) {
        *reinterpret_cast<f32x4*>(Kb + (long)r * S + c) = f32x4{c <= r ? row[c] : inv[c], c + 1 <= r ? row[c + 1] : inv[c + 1],
                                                              c + 2 <= r ? row[c + 2] : inv[c + 2], c + 3 <= r ? row[c + 3] : inv[c + 3]};
      }
    } else {
#pragma unroll
      for (int c = 0; c < NB; ++c) if (c <= r) Kb[(long)r * S + c] = row[c];
    }
    if (solve) a.w[(long)b * S + k0 + r] = acc;
    else a.y[(long)b * S + k0 + r] += acc;
  }
}

// ---------------------------------------------------------------------------------------------
// trsm: rows below the diagonal block.  One thread per row; row . L_kk^-T by forward substitution
// with L_kk broadcast from LDS.  Also the panel's contribution to y = L z.
// ---------------------------------------------------------------------------------------------
constexpr int OBW_C = 4 * NB;   // outer block width (OBW below)
__global__ __launch_bounds__(256) void gp_trsm_kernel(GpArgs a, int k0, int r_end, int mirror_planes) {
  __shared__ float L[NB][NB];  // L[j][m], m <= j
  __shared__ float zs[NB];
  const int b = blockIdx.y, S = a.S;
  float* Kb = a.K + (long)b * S * S;
  for (int i = threadIdx.x; i < NB * NB; i += 256) {
    const int r = i / NB, c = i % NB;
    L[r][c] = (c <= r) ? Kb[(long)(k0 + r) * S + k0 + c] : 0.f;
  }
  const float* zsrc = a.w ? a.w : a.z;   // posterior mode: the panel's solution takes the place of the base normals
  if (threadIdx.x < NB) zs[threadIdx.x] = zsrc[(long)b * S + k0 + threadIdx.x];
  __syncthreads();
  const int row = k0 + NB + blockIdx.x * 256 + threadIdx.x;
  if (row >= r_end) return;
  float* p = Kb + (long)row * S + k0;
  float v[NB];
#pragma unroll
  for (int c = 0; c < NB; c += 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p + c);
    v[c] = t[0]; v[c + 1] = t[1]; v[c + 2] = t[2]; v[c + 3] = t[3];
  }
  float ydot = 0.f;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    float s = v[j];
#pragma unroll
    for (int m = 0; m < j; ++m) s -= v[m] * L[j][m];
    s /= L[j][j];
    v[j] = s;
    ydot += s * zs[j];
  }
#pragma unroll
  for (int c = 0; c < NB; c += 4) *reinterpret_cast<f32x4*>(p + c) = f32x4{v[c], v[c + 1], v[c + 2], v[c + 3]};
  a.y[(long)b * S + row] += a.w ? -ydot : ydot;
  if (mirror_planes) {
    // The solved row is row rj of block (j, i) of the outer block's factor (i = this panel).  gp_trsm_wide_kernel multiplies those blocks on the fp16 matrix
    // cores, so the row is left once more as its two scaled fp16 terms (hi | lo, 128 + 128 bytes) -- in the MIRRORED block (i, j) of the matrix, which nothing
    // else uses (the Gram kernel writes the lower triangle only; potrf's inverses live inside the DIAGONAL 64 x 64 blocks): row k0 + rj, columns of block j.
    const int kout = k0 & ~(OBW_C - 1), rel = row - kout, jblk = rel >> 6, rj = rel & 63;
    const float sc = __builtin_ldexpf(1.f, plane_scale_exp(a, b));
    char* dst = reinterpret_cast<char*>(Kb + (long)(k0 + rj) * S + kout + jblk * NB);
#pragma unroll
    for (int c = 0; c < NB; c += 8) {
      f16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = v[c + e] * sc;
        hi[e] = (f16)x;
        lo[e] = (f16)(x - (float)hi[e]);
      }
      *reinterpret_cast<f16x8*>(dst + c * 2) = hi;
      *reinterpret_cast<f16x8*>(dst + 128 + c * 2) = lo;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// syrk: C_ij -= P_i P_j^T on 128x128 tiles of rows >= r0, columns [c0, c1), lower triangle only, where
// P = solved panel columns [kp0, kp0 + K), K a multiple of 64 up to 256.  f32-accurate (split-bf16 MFMA, below).
// Two uses (two-level blocking, see launch_gp_sample): K = 64 restricted to the remaining columns of the
// current 256-wide outer block, and K = 256 over the whole trailing matrix once per outer block -- the
// trailing matrix is then read and written S/256 times instead of S/64 times.
// ---------------------------------------------------------------------------------------------
// Arithmetic: f32 accuracy on the bf16 matrix cores.  Every panel value is split into three bf16 terms when its chunk is written
// to LDS, a = hi + mid + lo exactly to 2^-27 |a| (each residual is exact in f32), and a block product is the six bf16 MFMAs
// hi.hi + (hi.mid + mid.hi) + (mid.mid + hi.lo + lo.hi); the dropped terms are below 2^-25 |a||b|, the products are exact in the f32
// accumulator.  Six 32x32x16 bf16 MFMAs take 192 cycles per 16 panel columns where the eight 32x32x2 f32 MFMAs of the exact-f32 form
// take 512.  (Three products -- 2^-17 per term -- would not do: the Gram matrices have condition numbers of 1e6..1e7 in f32 and a
// noise floor of 1e-4 keeps them positive definite.)
constexpr int SYRK_KC = 32;   // panel columns per LDS chunk: 2 operands x 3 planes x 8 KiB per workgroup, so 3 workgroups share a CU
                              // and one's panel loads / C read-modify-write hide under another's MFMAs
constexpr int SYRK_PLANE = 128 * SYRK_KC * 2;   // one bf16 plane of a 128-row operand chunk
constexpr int SYRK_LDS = 2 * 3 * SYRK_PLANE;

// hi / mid / lo planes of the 4 floats of one 16-byte chunk -> three 8-byte LDS writes (row r, float chunk c of the 32-column chunk)
PFN_DEV void syrk_commit_split(LdsPtr tile, int r, int c, u32x4 raw) {
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const f32x4 v = __builtin_bit_cast(f32x4, raw);
  bf16x4 hi, mid, lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = (bf16)v[e];
    const float r1 = v[e] - (float)hi[e];
    mid[e] = (bf16)r1;
    lo[e] = (bf16)(r1 - (float)mid[e]);
  }
  const int off = lds_off16<SYRK_KC * 2>(r, c >> 1) + (c & 1) * 8;
  *reinterpret_cast<lds_bf16x4*>(tile + off) = hi;
  *reinterpret_cast<lds_bf16x4*>(tile + SYRK_PLANE + off) = mid;
  *reinterpret_cast<lds_bf16x4*>(tile + 2 * SYRK_PLANE + off) = lo;
}


// C -= acc for one 128 x 128 tile of the trailing update (lower triangle, rows < r1, columns < cend): read-modify-write in 16-byte pieces, lane = row,
// accumulator group g = 4 consecutive columns.  EVERY load is issued before the first store: the compiler never moves a load above a store (they may alias),
// so the natural "load, subtract, store" loop is a chain of sixteen dependent HBM round trips per wave -- ~2 us each, i.e. most of a tile's time (rocprofv3,
// round 4: 18 us per tile and CU against 5 us of MFMA work).  Tiles strictly below the diagonal and inside the matrix (three quarters of them) take a
// BRANCH-FREE path: sixteen loads, one wait, sixteen stores back to back -- with a branch around every piece hipcc's wait-count pass puts vmcnt(0) at the head
// of every block, and since gfx9 counts stores in the same counter each store then waits for the previous one's acknowledgement.
// (row0, col0) = first row / column of the wave's 64 x 64 sub-tile; `interior` (wave-uniform) = the whole workgroup tile lies strictly below the diagonal and inside the matrix
PFN_DEV void syrk_rmw(float* Kb, int S, const f32x16 (&acc)[2][2], int row0, int col0, bool interior, int r1, int cend, int lane) {
  const int h = lane >> 5;
  if (interior) {
    f32x4 c[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          c[i][j][gq] = *reinterpret_cast<const f32x4*>(Kb + (long)(row0 + i * 32 + (lane & 31)) * S + col0 + j * 32 + 8 * gq + 4 * h);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          f32x4 v = c[i][j][gq];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] -= acc[i][j][4 * gq + e];
          // (non-temporal loads / stores here were measured and are SLOWER: the rank-256 update of the first outer block 1958 vs 1632 us per 320 datasets)
          *reinterpret_cast<f32x4*>(Kb + (long)(row0 + i * 32 + (lane & 31)) * S + col0 + j * 32 + 8 * gq + 4 * h) = v;
        }
    return;
  }
  // diagonal tiles and the last tile row / column: pieces cut by the diagonal or the matrix edge go element by element
  f32x4 cold[2][2][4];
  unsigned whole = 0;                  // bit (i * 8 + j * 4 + gq): the piece lies inside the lower triangle and the matrix as a whole 16-byte vector
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int gi = row0 + i * 32 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int gj = col0 + j * 32 + 8 * gq + 4 * h;
        if (gi < r1 && gj + 3 <= gi && gj + 3 < cend) {
          cold[i][j][gq] = *reinterpret_cast<const f32x4*>(Kb + (long)gi * S + gj);
          whole |= 1u << (i * 8 + j * 4 + gq);
        }
      }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int gi = row0 + i * 32 + (lane & 31);
    if (gi >= r1) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int gj = col0 + j * 32 + 8 * gq + 4 * h;
        if (gj > gi || gj >= cend) continue;
        float* cp = Kb + (long)gi * S + gj;
        if (whole & (1u << (i * 8 + j * 4 + gq))) {
          f32x4 v = cold[i][j][gq];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] -= acc[i][j][4 * gq + e];
          *reinterpret_cast<f32x4*>(cp) = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gj + e <= gi && gj + e < cend) cp[e] -= acc[i][j][4 * gq + e];
        }
      }
  }
}

__global__ __launch_bounds__(256, 3) void gp_syrk_kernel(GpArgs a, int r0, int r1, int c0, int c1, int kp0, int K) {
  constexpr int RB = SYRK_KC * 4;  // 128-byte chunk rows in global memory (f32)
  constexpr int RBT = SYRK_KC * 2; // 64-byte rows of a bf16 plane in LDS
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr tA = lds_cast(smem_raw);
  LdsPtr tB = tA + 3 * SYRK_PLANE;
  // Hardware places workgroup n on XCD n % 8.  All tiles of one dataset share its panel rows, so datasets are
  // dealt to XCDs (dataset b -> XCD b % 8) when the batch allows it: the panel then stays in that XCD's L2
  // instead of being re-fetched by every tile.
  const int S = a.S, T = gridDim.x * gridDim.y;
  int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  int b, t;
  if (a.B % 8 == 0) { const int xcd = lin & 7, j = lin >> 3; b = xcd + 8 * (j / T); t = j % T; }
  else { b = lin / T; t = lin % T; }
  const int i0 = r0 + (t / gridDim.x) * 128, j0 = c0 + (t % gridDim.x) * 128;
  if (j0 > i0 + 127) return;          // tile entirely above the diagonal
  float* Kb = a.K + (long)b * S * S;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wm = wave >> 1, wn = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  using Stage = TileStage<float, 128, RB, 256>;
  Stage sa, sb;
  auto commit = [&](const Stage& st, LdsPtr tile) {
#pragma unroll
    for (int i = 0; i < Stage::PER; ++i) {
      const int id = threadIdx.x + i * 256;
      syrk_commit_split(tile, id / Stage::NCH, id % Stage::NCH, st.regs[i]);
    }
  };
  sa.issue(Kb + (long)i0 * S + kp0, S, r1 - i0, SYRK_KC);
  sb.issue(Kb + (long)j0 * S + kp0, S, r1 - j0, SYRK_KC);
  for (int kc = 0; kc < K; kc += SYRK_KC) {
    if (kc > 0) __syncthreads();      // everyone is done reading the previous chunk
    commit(sa, tA);
    commit(sb, tB);
    __syncthreads();
    if (kc + SYRK_KC < K) {           // next chunk's loads fly under this chunk's MFMAs
      sa.issue(Kb + (long)i0 * S + kp0 + kc + SYRK_KC, S, r1 - i0, SYRK_KC);
      sb.issue(Kb + (long)j0 * S + kp0 + kc + SYRK_KC, S, r1 - j0, SYRK_KC);
    }
#pragma unroll
    for (int ks = 0; ks < SYRK_KC; ks += 16) {
      Frag<bf16> fa[2][3], fb[2][3];   // [block][hi, mid, lo]
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fa[i][pl] = load_frag_row<bf16, RBT>(tA + pl * SYRK_PLANE, wm * 64 + i * 32 + (lane & 31), ks);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fb[j][pl] = load_frag_row<bf16, RBT>(tB + pl * SYRK_PLANE, wn * 64 + j * 32 + (lane & 31), ks);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {   // swapped operands: a lane owns one ROW of the tile; smallest terms first
          f32x16 c = acc[i][j];
          c = mma32(fb[j][2], fa[i][0], c);
          c = mma32(fb[j][0], fa[i][2], c);
          c = mma32(fb[j][1], fa[i][1], c);
          c = mma32(fb[j][1], fa[i][0], c);
          c = mma32(fb[j][0], fa[i][1], c);
          acc[i][j] = mma32(fb[j][0], fa[i][0], c);
        }
    }
  }
  syrk_rmw(Kb, S, acc, i0 + wm * 64, j0 + wn * 64, __builtin_amdgcn_readfirstlane((j0 + 127 < i0 && i0 + 127 < r1 && j0 + 127 < c1) ? 1 : 0) != 0, r1, c1, lane);
}

// ---------------------------------------------------------------------------------------------
// The rank-256 trailing update from PRE-SPLIT panels (round 4; fp16 planes round 5).  gp_syrk_kernel above re-reads its two f32 panels for every
// tile and splits every value again as the chunk goes to LDS -- for the trailing update of an outer block that is ~10x redundant vector work (one
// split per tile that touches the row) staged through 64 registers per lane.  Here gp_trsm_wide_kernel, which produces the solved panel anyway, ALSO
// leaves it split in a scratch behind K_ws (GpArgs::planes), laid out for the consumer:
//     slab (plane p, k-chunk kc of PL_KC columns)  =  [rows below the outer block][PL_KC] 2-byte values, contiguous (32 bytes per row),
//     the two 16-byte halves of a row swapped when (row >> 3) & 1 -- the bank swizzle of load_frag_row<.., 32>.
// A 128-row operand chunk of one plane is then 4 contiguous KiB: the update's tiles fetch them by LDS-DMA (dma16, no staging registers, no vector work).
//
// Round 5: TWO fp16 planes on a per-dataset power-of-two scale instead of three bf16 planes.  x s = hi + lo with hi = fp16(x s), lo = fp16(x s - hi):
// 11 + 11 mantissa bits, |x s - hi - lo| <= 2^-22 |x s|, and a block product is the THREE fp16 MFMAs  lo.hi + hi.lo + hi.hi  (the dropped lo.lo term is
// below 2^-22 of the product; products are exact in the f32 accumulator) -- half the matrix-pipe cycles of the six bf16 products and 4 instead of 6 bytes
// per panel value through L2 / the infinity cache, which is what the kernel was bound by.  fp16's narrow exponent is what the scale is for: every entry of
// a Cholesky factor obeys |L_ij| <= sqrt(K_ii) = sqrt(outputscale + noise), so s = 2^(14 - ceil(log2 sqrt(K_ii))) puts the largest entry below 2^14 (no
// overflow: fp16 reaches 65504; the products' sum stays far inside f32) and an entry keeps a NORMAL lo term down to 2^-16 of the largest; below that lo
// goes subnormal (absolute error 2^-25, i.e. 2^-39 of the largest entry -- and 2^-28 of it if the hardware flushed subnormals).  The scale is a power of
// two, so applying it and taking s^-2 out of the accumulator are exact.  Accuracy was priced BEFORE the kernel was written (tools/sim_gp_split.py, CPU
// emulation of the blocked factorisation): error of y against the f64 factorisation 7.0e-4 / 5.3e-7 (nf 5 / nf 18) where exact f32 gives 6.6e-4 / 5.3e-7
// and the six bf16 products 7.0e-4 / 4.9e-7; two bf16 terms / three products FAIL outright (non-positive pivots) on the nf-5 matrices.
// ---------------------------------------------------------------------------------------------
constexpr int SYP_SLAB = 128 * PL_ROWB;             // one plane of one 128-row operand chunk in LDS: 4 KiB
constexpr int SYP_STAGE = 2 * PL_NPL * SYP_SLAB;    // A and B, two planes each: 16 KiB
#ifndef PFN_SYP_NST
#define PFN_SYP_NST 3
#endif
#ifndef PFN_SYP_WGS
#define PFN_SYP_WGS 3
#endif
#ifndef PFN_SYP_ABLATE      // timing experiments only (results are wrong): 1 = no C read-modify-write, 2 = no MFMAs, 4 = no plane DMA
#define PFN_SYP_ABLATE 0
#endif
constexpr int SYP_NST = PFN_SYP_NST;                // stages in the ring: two in flight under the one being multiplied
constexpr int SYP_LDS = SYP_NST * SYP_STAGE;        // 48 KiB, three workgroups per CU
constexpr int SYP_PPW = 2 * PL_NPL * 4 / 4;         // 1-KiB DMA pieces per wave and stage


// One launch applies `nsets` solved panels (rank 256 each) to its region, so the region's C tiles are read and written once for all of them:
// set s in {set0, set1} holds the panel whose plane row 0 is global row r0 - off_s.
__global__ __launch_bounds__(256, PFN_SYP_WGS) void gp_syrk_planes_kernel(GpArgs a, int r0, int r1, int nsets, int set0, int off0, int set1, int off1) {
  // C[i][j] -= sum_s sum_k X_s[i][k] X_s[j][k] for rows >= r0 and the grid's tile columns from r0 on (lower triangle, 128 x 128 tiles), X_s = a 256-wide solved
  // panel read from its planes
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr smem = lds_cast(smem_raw);
  const int S = a.S, T = gridDim.x * gridDim.y;
  int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  int b, t;
  if (a.B % 8 == 0) { const int xcd = lin & 7, j = lin >> 3; b = xcd + 8 * (j / T); t = j % T; }     // a dataset's tiles on one XCD: its planes stay in that L2
  else { b = lin / T; t = lin % T; }
  const int i0 = r0 + (t / gridDim.x) * 128, j0 = r0 + (t % gridDim.x) * 128;
  if (j0 > i0 + 127) return;          // tile entirely above the diagonal
  float* Kb = a.K + (long)b * S * S;
  const char* pl = reinterpret_cast<const char*>(a.planes) + (long)b * PL_NSET * plane_set_bytes(a.plane_rows);
  const DmaRsrc rp = make_dma_rsrc(pl, (long)PL_NSET * plane_set_bytes(a.plane_rows));
  const int nk = nsets * PL_NKC;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, wm = wave >> 1, wn = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // a stage = 16 one-KiB pieces: piece q = (operand o = q / 8, plane p = (q / 4) % 2, quarter qq = q % 4) -> LDS (o * 2 + p) * SYP_SLAB + qq * 1024; wave w
  // moves pieces w, w + 4, ...  Source: 32 consecutive slab rows = one contiguous KiB (the swizzle is already in the slab).
  auto stage = [&](int buf, int ks) {
    const int second = ks >= PL_NKC, kc = ks & (PL_NKC - 1);
    const long sbase = (long)(second ? set1 : set0) * plane_set_bytes(a.plane_rows);
    const int roff = second ? off1 : off0;
#pragma unroll
    for (int i = 0; i < SYP_PPW; ++i) {
      const int q = wave + 4 * i;
      const int o = q / (4 * PL_NPL), p = (q / 4) % PL_NPL, qq = q % 4;
      const long row = (long)((o ? j0 : i0) - r0) + roff + qq * 32;
      const long off = sbase + plane_offset(a.plane_rows, p, kc, row) + lane * 16;
      if (!(PFN_SYP_ABLATE & 4)) dma16(rp, smem + buf * SYP_STAGE + (o * PL_NPL + p) * SYP_SLAB + qq * 1024, (int)off);
    }
  };
  stage(0, 0);
  if (SYP_NST > 2) stage(1, 1);
  int cur = 0;
  for (int kc = 0; kc < nk; ++kc) {
    // stage kc has landed for everyone (a wave's pieces retire in order: all but the SYP_PPW of stage kc + 1), and everyone is done reading stage kc - 1
    if (SYP_NST > 2 && kc + 1 < nk) wait_vm_barrier<SYP_PPW>(); else wait_vm_barrier<0>();
    if (kc + SYP_NST - 1 < nk) stage(cur == 0 ? SYP_NST - 1 : cur - 1, kc + SYP_NST - 1);      // into the buffer stage kc - 1 was read from
    const lds_char* tA = smem + cur * SYP_STAGE;
    const lds_char* tB = tA + PL_NPL * SYP_SLAB;
    Frag<bf16> fa[2][2], fb[2][2];     // [block][hi, lo]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p) fa[i][p] = load_frag_row<bf16, PL_ROWB>(tA + p * SYP_SLAB, wm * 64 + i * 32 + (lane & 31), 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p) fb[j][p] = load_frag_row<bf16, PL_ROWB>(tB + p * SYP_SLAB, wn * 64 + j * 32 + (lane & 31), 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {   // swapped operands: a lane owns one ROW of the tile; smallest terms first (as gp_syrk_kernel)
        f32x16 c = acc[i][j];
        if (PFN_SYP_ABLATE & 2) { c[0] += (float)fb[j][1].v[0] + (float)fa[i][0].v[0] + (float)fb[j][0].v[0] + (float)fa[i][1].v[0]; acc[i][j] = c; continue; }
        c = mma32_f16(fb[j][1], fa[i][0], c);
        c = mma32_f16(fb[j][0], fa[i][1], c);
        acc[i][j] = mma32_f16(fb[j][0], fa[i][0], c);
      }
    cur = cur == SYP_NST - 1 ? 0 : cur + 1;
  }
  const float inv_s2 = __builtin_ldexpf(1.f, -2 * plane_scale_exp(a, b));      // exact: the scale is a power of two
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv_s2;
  if (PFN_SYP_ABLATE & 1) {      // keep the accumulators alive without the 128 KiB of traffic
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    if (sum == 123.456f) Kb[0] = sum;
    return;
  }
  syrk_rmw(Kb, S, acc, i0 + wm * 64, j0 + wn * 64, __builtin_amdgcn_readfirstlane((j0 + 127 < i0 && i0 + 127 < r1) ? 1 : 0) != 0, r1, r1, lane);
}

// What bounds it now (round 5, profiles/r05_gp_sampler.txt): ablation builds (PFN_SYP_ABLATE) of the rank-256 form gave, per 7 x 320 datasets, 28.3 ms whole,
// 19.2 ms with the C read-modify-write alone (4 TB/s: the HBM's mixed read / write rate), 11.1 ms with the products alone and 14.3 without the read-modify-write
// -- the two phases ADD instead of overlapping, although three workgroups share a CU.  Requesting the C tile two stages before the subtraction (registers, peeled
// tail; built and measured: 25.79 vs 25.86 ms) changes nothing, so it is not the C round trip of the workgroup itself: a CU's vector memory path returns in
// order, and while ANY of its workgroups has C loads out at HBM latency the other workgroups' plane pieces (L2 hits) queue behind them -- the ring's two
// stages in flight then cover a fraction of that latency.  Hence the delayed update (a third less C traffic, launch_gp_sample) rather than more overlap.
// (A 256 x 256-tile form of this kernel -- 8 waves, three 48-KiB stages in a ring, one workgroup per CU -- was built and measured in round 4: the update of the first
// outer block 2013 us per 320 datasets against 1632 here.  The C read-modify-write of a tile costs the HBM as long as its products cost the matrix pipe (512 KiB against
// 20 us per 256 x 256 tile), and with one workgroup per CU the two no longer overlap; three co-resident 128 x 128 workgroups do overlap them.  profiles/r04_gp_sampler_experiments.txt)

// ---------------------------------------------------------------------------------------------
// wide trsm: all rows below a finished 256-wide outer block.  X = A[rows, kout:kout+256] . L_d^-T with L_d the
// factored 256x256 diagonal block, as a blocked forward substitution over its four 64-column blocks:
//   V_j -= sum_{i<j} X_i L_ji^T   exact-f32 MFMA, one 32x32 tile per wave
//   X_j  = V_j L_jj^-T            MFMA as well, with the explicit inverse of the 64x64 block factor that gp_potrf_kernel
//                                 leaves in the block's strictly-upper triangle
// plus the rows' contribution to y = L z.  One workgroup owns 64 rows (their 256 panel columns live in LDS).
// This is the only pass over the rows below per OUTER block; the 64-wide panel kernels above only touch the
// 256 rows of the diagonal block, so the chain of small launches no longer spreads over the whole chip.
// ---------------------------------------------------------------------------------------------
constexpr int OBW = 4 * NB;               // outer block width
constexpr int TW_STRIDE = OBW * 4 + 16;   // padded LDS row (pfn_device.h PadStride)
// Round 5: the factor's blocks are NOT staged in LDS any more.  With row block j of L_d beside the workgroup's rows the kernel held 133 KiB of LDS -- one
// workgroup (four waves) per CU, so nothing ran under its panel load (64 KiB from HBM), its store + plane writes (128 KiB) or its barriers: 4.05 ms per 320
// datasets for 1.0 ms of MFMA chain and ~1.3 ms of HBM traffic.  The L operand of both products is now read straight from global memory into the B fragments
// (a lane's 8 contraction values are 32 contiguous bytes of a row of L_d; the 256-KiB factor of a dataset stays in the L2 of the XCD all its workgroups run
// on -- the dataset -> XCD deal below), one 64-column block ahead of the MFMAs that consume it.  LDS = the rows alone, 67.6 KiB: two workgroups per CU.
// Round 5 (later): the block products V_j -= sum_{i<j} X_i L_ji^T -- two thirds of this kernel's MFMA chain, which the ablation builds showed to be its largest
// single part (8.5 of 20 ms per 7 x 320 datasets) -- run on the fp16 matrix cores like the trailing update: three products of two scaled fp16 terms per
// operand (accuracy: gp_syrk_planes_kernel; emulated for this use as well, tools/sim_gp_split.py `fp16x3+solve-fp16x3`).  Both operands arrive pre-split:
//   L_ji : left by gp_trsm_kernel in the mirrored block (i, j) of the matrix (hi | lo, 128 + 128 bytes per row), read straight into the B fragments;
//   X_i  : when a solved block X_j leaves the accumulators it is written to LDS AS its two terms, over the f32 values of V_j it replaces (same 256 bytes per row);
//          the plane scratch of the trailing update is then a copy of these LDS bytes, and y += X z is taken from the (exact f32) accumulators.
// The solve X_j = V_j L_jj^-T stays on the exact-f32 MFMA (the inverse block's entries have no a-priori bound, hence no safe fp16 scale).
__global__ __launch_bounds__(256, 2) void gp_trsm_wide_kernel(GpArgs a, int kout, int plane_set) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr V = lds_cast(smem_raw);                     // [64][TW_STRIDE]  the workgroup's rows: block j as f32 until it is solved, then as hi | lo fp16 terms
  float* zs = reinterpret_cast<float*>(smem_raw + 64 * TW_STRIDE);  // [256]
  float* yacc = zs + OBW;                            // [2][64] partial y per column half
  const int S = a.S;
  int b = blockIdx.y, rb = blockIdx.x;
  if (a.B % 8 == 0) {      // hardware places workgroup n on XCD n % 8: all row blocks of a dataset on one XCD, whose L2 then holds that dataset's factor
    const int lin = blockIdx.x + gridDim.x * blockIdx.y, xcd = lin & 7, j = lin >> 3;
    b = xcd + 8 * (j / (int)gridDim.x); rb = j % (int)gridDim.x;
  }
  const int row0 = kout + OBW + rb * 64;
  const int rows_valid = min(64, S - row0);
  float* Kb = a.K + (long)b * S * S;
  const float* Ld = Kb + (long)kout * S + kout;      // the factored 256 x 256 diagonal block [L \ L^-T of its 64 x 64 diagonal blocks; planes of L_ji in block (i, j)]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
  const int rt = wave >> 1, ct = wave & 1, c = ct * 32 + li;
  const int sexp = plane_scale_exp(a, b);
  const float sc = __builtin_ldexpf(1.f, sexp), inv_s2 = __builtin_ldexpf(1.f, -2 * sexp);
  {
    TileStage<float, 64, OBW * 4, 256> sv;
    sv.issue(Kb + (long)row0 * S + kout, S, rows_valid, OBW);
    const float zv = (a.w ? a.w : a.z)[(long)b * S + kout + threadIdx.x];
    sv.template commit_p<TW_STRIDE>(V);
    zs[threadIdx.x] = zv;
  }
  // B fragments of the block products: the two fp16 terms of L_ji[row c of block j][k0 + 8h .. + 7], 16 bytes each (mirrored block (i, j): row kout + 64 i + c)
  struct LFrag { Frag<bf16> hi, lo; };
  auto l_frag = [&](int jb, int ib, int k0) {
    const char* p = reinterpret_cast<const char*>(Ld + (long)(ib * NB + c) * S + jb * NB) + (k0 + 8 * h) * 2;
    LFrag f;
    f.hi.v = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
    f.lo.v = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + 128));
    return f;
  };
  // X_j = V_j L_jj^-T needs the inverse block: L^-1[c][k] (k < c) sits at (row k, column c) of the diagonal block, the diagonal holds L[c][c], everything with
  // k > c is masked to zero.
  const int kmax = ct == 0 ? 32 : NB;      // columns c < 32 only see k < 32
  auto inv_frags = [&](int jb, Frag<float> (&finv)[4], float& ldiag) {
    const float* dcol = Ld + (long)(jb * NB) * S + jb * NB + c;
    ldiag = dcol[(long)c * S];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = q * 16 + 8 * h + e;
        finv[q].v[e] = (q * 16 < kmax) ? dcol[(long)k * S] : 0.f;
      }
  };
  // The barriers of the block loop order LDS traffic only (the factor is read-only here), so they wait for the LDS counter and leave the vector-memory
  // counter alone (wait_vm_barrier<63>): __syncthreads() carries a fence that drains every load in flight.  What a block needs from the factor is requested
  // one phase ahead: the inverse block and the first 64 columns of row block j + 1 travel under the X-phase products of block j.
  Frag<float> finv[4];
  LFrag fcur[4];
  float ldiag;
  float ysum[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) ysum[r] = 0.f;
  inv_frags(0, finv, ldiag);
#pragma unroll
  for (int jb = 0; jb < 4; ++jb) {
    wait_vm_barrier<63>();   // V current (initial load / previous block's solution)
    if (jb > 0) {
      // V_j -= sum_{i<j} X_i L_ji^T   (three fp16 products per 16 contraction values; smallest terms first)
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      LFrag fnext[4];
#pragma unroll
      for (int ib = 0; ib < jb; ++ib) {
        if (ib + 1 < jb) {
#pragma unroll
          for (int q = 0; q < 4; ++q) fnext[q] = l_frag(jb, ib + 1, q * 16);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const lds_char* xp = V + (rt * 32 + li) * TW_STRIDE + ib * (NB * 4) + (q * 16 + 8 * h) * 2;
          Frag<bf16> xh, xl;
          xh.v = __builtin_bit_cast(bf16x8, lds_read16(xp));
          xl.v = __builtin_bit_cast(bf16x8, lds_read16(xp + 128));
          acc = mma32_f16(xl, fcur[q].hi, acc);
          acc = mma32_f16(xh, fcur[q].lo, acc);
          acc = mma32_f16(xh, fcur[q].hi, acc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) fcur[q] = fnext[q];
      }
      // (columns jb * 64 .. are read by nobody in this phase: no barrier in front of the update)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        LdsPtr p = V + (rt * 32 + acc_row(r, lane)) * TW_STRIDE + (jb * NB + c) * 4;
        lds_write_f32(p, lds_read_f32(p) - acc[r] * inv_s2);
      }
      wait_vm_barrier<63>();
    }
    {
      const float rdiag = 1.f / ldiag;
      Frag<float> fb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = q * 16 + 8 * h + e;
          fb[q].v[e] = k < c ? finv[q].v[e] : (k == c ? rdiag : 0.f);
        }
      if (jb + 1 < 4) {      // the next block's share of the factor, requested under this block's products
        inv_frags(jb + 1, finv, ldiag);
#pragma unroll
        for (int q = 0; q < 4; ++q) fcur[q] = l_frag(jb + 1, 0, q * 16);
      }
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q * 16 >= kmax) break;
        acc = mma32(load_frag_row_p<float, TW_STRIDE>(V, rt * 32 + li, jb * NB + q * 16), fb[q], acc);
      }
      // this block's part of y = X z (or of the forward solve), from the exact values: the lane's column times z, summed over the 32 lanes of its half
      const float zc = zs[jb * NB + c];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float t = acc[r] * zc;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        ysum[r] += t;
      }
      if (!a.planes) {      // no plane scratch: the update will split the f32 rows itself (gp_syrk_kernel), so they go back to the matrix -- exact, from the accumulators
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = rt * 32 + acc_row(r, lane);
          if (rr < rows_valid) Kb[(long)(row0 + rr) * S + kout + jb * NB + c] = acc[r];
        }
      }
      wait_vm_barrier<63>();   // all of V_j has been read
      typedef __attribute__((address_space(3))) f16 lds_f16;
#pragma unroll
      for (int r = 0; r < 16; ++r) {      // X_j as its two scaled fp16 terms, over V_j
        const float x = acc[r] * sc;
        const f16 xh = (f16)x;
        LdsPtr p = V + (rt * 32 + acc_row(r, lane)) * TW_STRIDE + jb * (NB * 4) + c * 2;
        *reinterpret_cast<lds_f16*>(p) = xh;
        *reinterpret_cast<lds_f16*>(p + 128) = (f16)(x - (float)xh);
      }
    }
  }
  if (li == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) yacc[ct * 64 + rt * 32 + acc_row(r, lane)] = ysum[r];
  }
  __syncthreads();
  if (threadIdx.x < 64 && (int)threadIdx.x < rows_valid) {
    const float d = yacc[threadIdx.x] + yacc[64 + threadIdx.x];
    a.y[(long)b * S + row0 + threadIdx.x] += a.w ? -d : d;
  }
  if (a.planes) {
    // the solved rows for the trailing update (gp_syrk_planes_kernel: layout, scale and why): a copy of the LDS terms.  A task = 8 consecutive columns of a
    // row -> one 16-byte half row in each plane.  (The rows themselves are NOT stored: with the planes nobody reads these columns of the matrix again.)
    char* pl = reinterpret_cast<char*>(a.planes) + ((long)b * PL_NSET + plane_set) * plane_set_bytes(a.plane_rows);
    const long prow0 = (long)row0 - (kout + OBW);                      // row index inside the planes: rows below the outer block
    for (int id = threadIdx.x; id < 64 * (OBW / 8); id += 256) {
      const int r = id / (OBW / 8), c8 = id % (OBW / 8);                 // columns 8 c8 .. 8 c8 + 7: block c8 / 8, 16-byte group c8 % 8 of its hi / lo halves
      if (r >= rows_valid) continue;
      const lds_char* src = V + r * TW_STRIDE + (c8 >> 3) * (NB * 4) + (c8 & 7) * 16;
      const u32x4 hi = lds_read16(src), lo = lds_read16(src + 128);
      const int kc = c8 / 2;
      const long prow = prow0 + r;
      const int half = (c8 & 1) ^ (int)((prow >> 3) & 1);                // the consumer's bank swizzle (load_frag_row<.., 32>), applied at the source
      *reinterpret_cast<u32x4*>(pl + plane_offset(a.plane_rows, 0, kc, prow) + half * 16) = hi;
      *reinterpret_cast<u32x4*>(pl + plane_offset(a.plane_rows, 1, kc, prow) + half * 16) = lo;
    }
  }
}

// K_ws of the C ABI: K [B, S, S] f32, then (256-byte aligned) the plane scratch of gp_syrk_planes_kernel
static long gp_plane_rows(int S) { return S > OBW ? ((long)(S - OBW + 127) / 128) * 128 : 0; }      // whole 128-row tiles of the trailing update
static int64_t gp_k_bytes(int B, int S) { return ((int64_t)B * S * S * 4 + 255) / 256 * 256; }
int64_t gp_workspace_bytes(int B, int S) { return gp_k_bytes(B, S) + (int64_t)B * PL_NSET * PL_NPL * PL_NKC * gp_plane_rows(S) * PL_ROWB; }
void gp_attach_planes(GpArgs& a) {      // a.K = the caller's K_ws of gp_workspace_bytes(B, S) bytes
  a.plane_rows = gp_plane_rows(a.S);
  a.planes = a.plane_rows > 0 ? reinterpret_cast<char*>(a.K) + gp_k_bytes(a.B, a.S) : nullptr;
}

int launch_gp_sample(const GpArgs& a, hipStream_t s) {
  const int S = a.S, B = a.B;
  if (S % 4) return PFN_ERR_UNSUPPORTED;  // 16-byte aligned matrix rows
  if (!a.w) {
    const long work = ((long)B * S * a.nf + 3) / 4 + ((long)B * S + 3) / 4;
    const int grid = (int)std::max<long>(1, std::min<long>((work + 255) / 256, 4096));
    hipLaunchKernelGGL(gp_rng_kernel, dim3(grid), dim3(256), 0, s, a);
  }
  {
    const int t = (S + 63) / 64;
    const size_t lds = (128 * (a.nf + 1) + a.nf) * sizeof(float);
    if (lds > 64 * 1024) return PFN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gp_gram_kernel, dim3(t * (t + 1) / 2, B), dim3(256), lds, s, a);
  }
  // Two-level right-looking blocked Cholesky.  Per 256-wide outer block:
  //   narrow chain (touches only the block's own <= 256 rows, at most B..4B workgroups, harmless to whatever
  //   shares the GPU): 64-wide panels  potrf -> trsm -> rank-64 update, all restricted to rows < kend;
  //   wide, once per outer block: trsm of every row below against the whole 256-wide factor, then one
  //   rank-256 update of the trailing matrix.
  auto syrk = [&](int r0, int r1, int c0, int c1, int kp0, int K) {
    const int ti = (r1 - r0 + 127) / 128, tj = (c1 - c0 + 127) / 128;
    if (ti > 0 && tj > 0) hipLaunchKernelGGL(gp_syrk_kernel, dim3(tj, ti, B), dim3(256), SYRK_LDS, s, a, r0, r1, c0, c1, kp0, K);
  };
  const size_t tw_lds = 64 * TW_STRIDE + (OBW + 128) * sizeof(float);
  static LdsAllowance tw_allowance;   // (constant size; per device; the call costs tens of microseconds of host time)
  tw_allowance.ensure(gp_trsm_wide_kernel, tw_lds);
  for (int kout = 0; kout < S; kout += OBW) {
    const int kend = std::min(kout + OBW, S);
    for (int k0 = kout; k0 < kend; k0 += NB) {
      hipLaunchKernelGGL(gp_potrf_kernel, dim3(B), dim3(64), 0, s, a, k0);
      const int next = k0 + NB;
      if (next >= kend) break;
      hipLaunchKernelGGL(gp_trsm_kernel, dim3((kend - next + 255) / 256, B), dim3(256), 0, s, a, k0, kend, kend < S ? 1 : 0);
      syrk(next, kend, next, kend, k0, NB);
    }
    if (kend < S) {
      const int odd = (kout / OBW) & 1;
      hipLaunchKernelGGL(gp_trsm_wide_kernel, dim3((S - kend + 63) / 64, B), dim3(256), tw_lds, s, a, kout, odd);
      if (a.planes) {
        // DELAYED trailing update from the pre-split planes (round 5): the read-modify-write of the trailing matrix is what the update is bound by (HBM: alone it
        // takes 19.2 of the kernel's 28.3 ms per 7 x 320 datasets, profiles/r05_gp_sampler.txt), so the matrix is touched once per PAIR of outer blocks.  After an
        // even block only the next block's 256 columns are brought up to date (a strip, rank 256, plane set 0); after the odd block the rest takes both panels in
        // one pass (rank 512: set 0 = the even block's panel, whose plane row 0 is global row kend - 256; set 1 = this block's).  C traffic per dataset at
        // S = 2000: 21.5 MB instead of 34.
        const int tn = (S - kend + 127) / 128;
        if (!odd) hipLaunchKernelGGL(gp_syrk_planes_kernel, dim3(std::min(tn, OBW / 128), tn, B), dim3(256), SYP_LDS, s, a, kend, S, 1, 0, 0, 0, 0);
        else hipLaunchKernelGGL(gp_syrk_planes_kernel, dim3(tn, tn, B), dim3(256), SYP_LDS, s, a, kend, S, 2, 0, OBW, 1, 0);
      } else {
        syrk(kend, S, kend, S, kout, OBW);
      }
    }
  }
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------
// Sequential exact-GP predictions (reference priors/fast_gp.py:88-120 `evaluate`: for every t it refits
// an ExactGP on points 0..t-1 and predicts point t -- one Cholesky per t).  All of them fall out of ONE
// factorisation of the full covariance C = outputscale k(x,x) + noise I = L L^T: the leading t x t block of L
// is the factor of the first t points, row t of L is L_t^-1 k_t, and the forward solve w = L^-1 y is causal, so
//   mean_t = sum_{j<t} L[t,j] w[j] = y_t - L[t,t] w[t],   var_t (with noise) = L[t,t]^2,
//   -log N(y_t; mean_t, var_t) = log L[t,t] + w[t]^2 / 2 + log(2 pi) / 2.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gp_posterior_kernel(GpArgs a, const float* y_data, float* nll, float* mean, float* var) {
  const long n = (long)a.B * a.S;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / a.S, t = i % a.S;
    const float d = a.K[(b * a.S + t) * a.S + t], w = a.w[i];
    if (nll) nll[i] = __logf(d) + 0.5f * w * w + 0.9189385332046727f;
    if (mean) mean[i] = y_data[i] - d * w;
    if (var) var[i] = d * d;
  }
}

int launch_gp_posterior(const GpArgs& a, const float* y_data, float* nll, float* mean, float* var, hipStream_t s) {
  if (!a.w || !a.y || !y_data) return PFN_ERR_ARGUMENT;
  if (hipMemcpyAsync(a.y, y_data, sizeof(float) * a.B * a.S, hipMemcpyDeviceToDevice, s) != hipSuccess) return PFN_ERR_LAUNCH;
  const int rc = launch_gp_sample(a, s);
  if (rc != PFN_OK) return rc;
  const long n = (long)a.B * a.S;
  hipLaunchKernelGGL(gp_posterior_kernel, dim3((int)std::min<long>((n + 255) / 256, 2048)), dim3(256), 0, s, a, y_data, nll, mean, var);
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}

}  // namespace pfn
