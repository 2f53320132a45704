// MFMA GEMMs for the PFN encoder stack (gfx950).
//
//   gemm_nt : C[M,N] = epilogue( A[M,K] . B[N,K]^T )      forward linears and dgrads
//             (dgrad uses the pre-transposed weight copy, so it is "NT" as well)
//   gemm_tn : C[P,Q] += A[M,P]^T . B[M,Q]                  weight gradients, split over M
//
// Both are templated on the operand type: bf16 (v_mfma_f32_32x32x16_bf16) is the product path,
// float (v_mfma_f32_32x32x2_f32, exact f32) is the high-precision mode used to separate
// algorithmic from rounding error in the parity tests.  Replaces the torch.nn.Linear calls
// inside nn.TransformerEncoderLayer (reference transformer.py:17-18,84; torch
// nn/modules/transformer.py:952-982, nn/functional.py:6435,6637).
#include "pfn_device.h"
#include "pfn_kernels.h"

namespace pfn {

// XCD-aware, bijective remap of the linear workgroup id: hardware places block b on XCD b%8;
// give every XCD a contiguous range of tile ids so neighbouring tiles (same A row panel) share
// one L2.  (guide §5 "XCD swizzle must be bijective")
PFN_DEV int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_RB = 128;  // RB: bytes of contraction per tile row

template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNT g) {
  constexpr int BK = GEMM_RB / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr smem = lds_cast(smem_raw);
  // [buf][A|B] each 128 rows x 128 B = 16 KiB
  auto tileA = [&](int buf) { return smem + buf * 32768; };
  auto tileB = [&](int buf) { return smem + buf * 32768 + 16384; };

  const int tiles_n = (g.N + GEMM_BN - 1) / GEMM_BN;
  const int tiles_m = (g.M + GEMM_BM - 1) / GEMM_BM;
  const int tid_lin = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = tid_lin / tiles_n, tn = tid_lin % tiles_n;
  const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;
  const int rowsA = min(GEMM_BM, g.M - m0), rowsB = min(GEMM_BN, g.N - n0);

  const T* A = reinterpret_cast<const T*>(g.A) + (long)m0 * g.lda;
  const T* B = reinterpret_cast<const T*>(g.B) + (long)n0 * g.ldb;

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TileStage<T, GEMM_BM, GEMM_RB, 256> sa, sb;
  const int nk = (g.K + BK - 1) / BK;
  sa.issue(A, g.lda, rowsA, g.K);
  sb.issue(B, g.ldb, rowsB, g.K);
  sa.template commit<false>(tileA(0));
  sb.template commit<false>(tileB(0));
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      const int k1 = (kt + 1) * BK;
      sa.issue(A + k1, g.lda, rowsA, g.K - k1);
      sb.issue(B + k1, g.ldb, rowsB, g.K - k1);
    }
    const lds_char* ta = tileA(cur);
    const lds_char* tb = tileB(cur);
#pragma unroll
    for (int ks = 0; ks < BK; ks += 16) {
      Frag<T> fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = load_frag_row<T, GEMM_RB>(ta, wm * 64 + i * 32 + (lane & 31), ks);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = load_frag_row<T, GEMM_RB>(tb, wn * 64 + j * 32 + (lane & 31), ks);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mma32(fa[i], fb[j], acc[i][j]);
    }
    if (kt + 1 < nk) {
      sa.template commit<false>(tileA(cur ^ 1));
      sb.template commit<false>(tileB(cur ^ 1));
    }
    __syncthreads();
  }

  // ---- epilogue: accumulators -> per-wave 64x64 f32 LDS patch -> vectorised row-wise pass ----
  LdsPtr patch = smem + wave * 16384;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        lds_write_f32(patch + ((i * 32 + acc_row(r, lane)) * 64 + j * 32 + (lane & 31)) * 4, acc[i][j][r]);
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): own-wave LDS writes done (patch is wave private)
  __builtin_amdgcn_wave_barrier();

  const int flags = g.flags;
  const float* bias = g.bias;
  const T* aux = reinterpret_cast<const T*>(g.aux);
  T* out_t = reinterpret_cast<T*>(g.out_t);
  T* out2_t = reinterpret_cast<T*>(g.out2_t);
  const int c4 = (lane & 15) * 4;
  const int ncol = n0 + wn * 64 + c4;
#pragma unroll 4
  for (int p = 0; p < 16; ++p) {
    const int rloc = p * 4 + (lane >> 4);
    const long m = (long)m0 + wm * 64 + rloc;
    if (m >= g.M || ncol >= g.N) continue;
    f32x4 v = __builtin_bit_cast(f32x4, lds_read16(patch + (rloc * 64 + c4) * 4));
    const bool full = (ncol + 3 < g.N) && g.vec_ok;
    if (full) {
      if (flags & EPI_BIAS) { f32x4 b = *reinterpret_cast<const f32x4*>(bias + ncol); v += b; }
      if (flags & EPI_GELU_BWD) {
        bf16x4 dummy; (void)dummy;
        float a[4];
        if constexpr (sizeof(T) == 2) { bf16x4 t = *reinterpret_cast<const bf16x4*>(aux + m * g.ld_aux + ncol); for (int e = 0; e < 4; ++e) a[e] = (float)t[e]; }
        else { f32x4 t = *reinterpret_cast<const f32x4*>(aux + m * g.ld_aux + ncol); for (int e = 0; e < 4; ++e) a[e] = t[e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_f(a[e]);
      }
      if (flags & EPI_RESID) { f32x4 r = *reinterpret_cast<const f32x4*>(g.resid + m * g.ld_resid + ncol); v += r; }
      if (flags & EPI_OUT2_T) {
        if constexpr (sizeof(T) == 2) { bf16x4 t; for (int e = 0; e < 4; ++e) t[e] = (bf16)v[e]; *reinterpret_cast<bf16x4*>(out2_t + m * g.ld_out2 + ncol) = t; }
        else *reinterpret_cast<f32x4*>(out2_t + m * g.ld_out2 + ncol) = v;
      }
      if (flags & EPI_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
      }
      if (flags & EPI_OUT_F32) {
        float* o = g.out_f32 + m * g.ld_out_f32 + ncol;
        if (flags & EPI_ACCUM) { f32x4 old = *reinterpret_cast<f32x4*>(o); v += old; }
        *reinterpret_cast<f32x4*>(o) = v;
      }
      if (flags & EPI_OUT_T) {
        if constexpr (sizeof(T) == 2) { bf16x4 t; for (int e = 0; e < 4; ++e) t[e] = (bf16)v[e]; *reinterpret_cast<bf16x4*>(out_t + m * g.ld_out_t + ncol) = t; }
        else *reinterpret_cast<f32x4*>(out_t + m * g.ld_out_t + ncol) = v;
      }
    } else {
      for (int e = 0; e < 4 && ncol + e < g.N; ++e) {
        float x = v[e];
        const int n = ncol + e;
        if (flags & EPI_BIAS) x += bias[n];
        if (flags & EPI_GELU_BWD) x *= gelu_grad_f((float)aux[m * g.ld_aux + n]);
        if (flags & EPI_RESID) x += g.resid[m * g.ld_resid + n];
        if (flags & EPI_OUT2_T) out2_t[m * g.ld_out2 + n] = (T)x;
        if (flags & EPI_GELU) x = gelu_f(x);
        if (flags & EPI_OUT_F32) { float* o = g.out_f32 + m * g.ld_out_f32 + n; *o = (flags & EPI_ACCUM) ? (*o + x) : x; }
        if (flags & EPI_OUT_T) out_t[m * g.ld_out_t + n] = (T)x;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// C[P,Q] (+)= A[M,P]^T . B[M,Q]     contraction over the (long) token axis, split across
// workgroups in z; partial tiles are added with hardware f32 atomics.
// ---------------------------------------------------------------------------------------------
constexpr int TN_BMK = 32;  // token rows per LDS tile

template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmTN g) {
  constexpr int RB = 128 * sizeof(T);  // 128 columns per tile row
  constexpr int TILE = TN_BMK * RB;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr smem = lds_cast(smem_raw);
  auto tileA = [&](int buf) { return smem + buf * 2 * TILE; };
  auto tileB = [&](int buf) { return smem + buf * 2 * TILE + TILE; };

  const int p0 = blockIdx.y * 128, q0 = blockIdx.x * 128;
  const int colsA = min(128, g.P - p0), colsB = min(128, g.Q - q0);
  const long mbeg = (long)blockIdx.z * g.m_chunk;
  const long mend = min((long)g.M, mbeg + g.m_chunk);
  if (mbeg >= mend) return;
  const T* A = reinterpret_cast<const T*>(g.A) + mbeg * g.lda + p0;
  const T* B = reinterpret_cast<const T*>(g.B) + mbeg * g.ldb + q0;

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wp = wave >> 1, wq = wave & 1;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TileStage<T, TN_BMK, RB, 256> sa, sb;
  const int rows_total = (int)(mend - mbeg);
  const int nt = (rows_total + TN_BMK - 1) / TN_BMK;
  // fused bias gradient: column sums of A (= dY) ride along in the workgroups of the first Q tile.
  // Every thread always stages the same 16-byte column chunk (256 % chunks-per-row == 0).
  const bool do_colsum = g.colsum != nullptr && blockIdx.x == 0;
  constexpr int EPC = 16 / sizeof(T);
  float csum[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) csum[e] = 0.f;
  auto add_colsum = [&]() {
    if (!do_colsum) return;
#pragma unroll
    for (int i = 0; i < sa.PER; ++i) {
      if constexpr (sizeof(T) == 2) {
        const bf16x8 v = __builtin_bit_cast(bf16x8, sa.regs[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] += (float)v[e];
      } else {
        const f32x4 v = __builtin_bit_cast(f32x4, sa.regs[i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[e] += v[e];
      }
    }
  };
  sa.issue(A, g.lda, rows_total, colsA);
  sb.issue(B, g.ldb, rows_total, colsB);
  add_colsum();
  sa.template commit<true>(tileA(0));
  sb.template commit<true>(tileB(0));
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) {
      const long r1 = (long)(t + 1) * TN_BMK;
      sa.issue(A + r1 * g.lda, g.lda, rows_total - (int)r1, colsA);
      sb.issue(B + r1 * g.ldb, g.ldb, rows_total - (int)r1, colsB);
      add_colsum();
    }
    const lds_char* ta = tileA(cur);
    const lds_char* tb = tileB(cur);
#pragma unroll
    for (int ks = 0; ks < TN_BMK; ks += 16) {
      Frag<T> fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = load_frag_tr<T, RB, 1>(ta, ks, wp * 64 + i * 32);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = load_frag_tr<T, RB, 1>(tb, ks, wq * 64 + j * 32);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mma32(fa[i], fb[j], acc[i][j]);
    }
    if (t + 1 < nt) {
      sa.template commit<true>(tileA(cur ^ 1));
      sb.template commit<true>(tileB(cur ^ 1));
    }
    __syncthreads();
  }

  if (do_colsum) {
    // threads tid, tid+NCH, ... share a column chunk: reduce over the 256/NCH row groups through LDS
    constexpr int NCH = RB / 16;
    float* red = reinterpret_cast<float*>(smem_raw);  // all tiles are dead after the last barrier
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[threadIdx.x * EPC + e] = csum[e];
    __syncthreads();
    if (threadIdx.x < NCH) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        float t = 0.f;
        for (int j = 0; j < 256 / NCH; ++j) t += red[(j * NCH + threadIdx.x) * EPC + e];
        const int col = p0 + threadIdx.x * EPC + e;
        if (col < g.P) unsafeAtomicAdd(g.colsum + col, t);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = p0 + wp * 64 + i * 32 + acc_row(r, lane);
        const int q = q0 + wq * 64 + j * 32 + (lane & 31);
        if (p < g.P && q < g.Q) {
          float* c = g.C + (long)p * g.ldc + q;
          if (g.atomic) unsafeAtomicAdd(c, acc[i][j][r]);
          else *c = acc[i][j][r];
        }
      }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int launch_gemm_nt(const GemmNT& g_in, int precision, hipStream_t stream) {
  GemmNT g = g_in;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return PFN_OK;
  const size_t es = precision == PFN_PREC_BF16 ? 2 : 4;
  if ((g.lda * es) % 16 || (g.ldb * es) % 16 || !aligned16(g.A) || !aligned16(g.B)) return PFN_ERR_ALIGNMENT;
  // the epilogue moves 4 columns per lane when every stream it touches allows it
  bool vec = true;
  if ((g.flags & EPI_OUT_F32) && (g.ld_out_f32 % 4 || !aligned16(g.out_f32))) vec = false;
  if ((g.flags & EPI_OUT_T) && ((g.ld_out_t * es) % (4 * es) || !aligned16(g.out_t))) vec = false;
  if ((g.flags & EPI_OUT2_T) && ((g.ld_out2 * es) % (4 * es) || !aligned16(g.out2_t))) vec = false;
  if ((g.flags & EPI_RESID) && (g.ld_resid % 4 || !aligned16(g.resid))) vec = false;
  if ((g.flags & EPI_GELU_BWD) && ((g.ld_aux * es) % (4 * es) || !aligned16(g.aux))) vec = false;
  if ((g.flags & EPI_BIAS) && !aligned16(g.bias)) vec = false;
  g.vec_ok = vec ? 1 : 0;
  const int tiles = ((g.M + GEMM_BM - 1) / GEMM_BM) * ((g.N + GEMM_BN - 1) / GEMM_BN);
  if (precision == PFN_PREC_BF16) hipLaunchKernelGGL(gemm_nt_kernel<bf16>, dim3(tiles), dim3(256), 65536, stream, g);
  else hipLaunchKernelGGL(gemm_nt_kernel<float>, dim3(tiles), dim3(256), 65536, stream, g);
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}

int launch_gemm_tn(GemmTN g, int precision, hipStream_t stream) {
  if (g.M <= 0 || g.P <= 0 || g.Q <= 0) return PFN_OK;
  const size_t es = precision == PFN_PREC_BF16 ? 2 : 4;
  if ((g.lda * es) % 16 || (g.ldb * es) % 16 || !aligned16(g.A) || !aligned16(g.B)) return PFN_ERR_ALIGNMENT;
  const int tp = (g.P + 127) / 128, tq = (g.Q + 127) / 128;
  int splits = (1024 + tp * tq - 1) / (tp * tq);
  const int max_splits = (g.M + 4 * TN_BMK - 1) / (4 * TN_BMK);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (!g.atomic) splits = 1;
  int chunk = (g.M + splits - 1) / splits;
  chunk = (chunk + TN_BMK - 1) / TN_BMK * TN_BMK;
  splits = (g.M + chunk - 1) / chunk;
  g.m_chunk = chunk;
  const size_t lds = 4 * TN_BMK * 128 * es;
  if (precision == PFN_PREC_BF16) hipLaunchKernelGGL(gemm_tn_kernel<bf16>, dim3(tq, tp, splits), dim3(256), lds, stream, g);
  else hipLaunchKernelGGL(gemm_tn_kernel<float>, dim3(tq, tp, splits), dim3(256), lds, stream, g);
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}

}  // namespace pfn
