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
#include <type_traits>
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
  operand_store_mode<T>();
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
      if ((flags & EPI_ROWSHIFT) && ncol >= g.rs_n0 && ncol < g.rs_n1) v -= *reinterpret_cast<const f32x4*>(g.rowshift + (m / g.rs_S) * g.rs_ld + (ncol - g.rs_n0));
      if (flags & EPI_GELU_BWD) {
        float a[4];
        if constexpr (sizeof(T) == 2) { X4<T> t = *reinterpret_cast<const X4<T>*>(aux + m * g.ld_aux + ncol); for (int e = 0; e < 4; ++e) a[e] = (float)t[e]; }
        else { f32x4 t = *reinterpret_cast<const f32x4*>(aux + m * g.ld_aux + ncol); for (int e = 0; e < 4; ++e) a[e] = t[e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= a[e];
      }
      if (flags & EPI_RESID) { f32x4 r = *reinterpret_cast<const f32x4*>(g.resid + m * g.ld_resid + ncol); v += r; }
      if (flags & EPI_RESID_T) {
        if constexpr (sizeof(T) == 2) { X4<T> t = *reinterpret_cast<const X4<T>*>(aux + m * g.ld_aux + ncol); for (int e = 0; e < 4; ++e) v[e] += (float)t[e]; }
        else { f32x4 t = *reinterpret_cast<const f32x4*>(aux + m * g.ld_aux + ncol); v += t; }
      }
      f32x4 second = v;          // EPI_OUT2_T: the value before the activation, or with EPI_GELU the activation's derivative there
      if (flags & EPI_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { float y, dy; gelu_and_grad(v[e], y, dy); v[e] = y; second[e] = dy; }
      }
      if (flags & EPI_OUT2_T) {
        if constexpr (sizeof(T) == 2) { X4<T> t; for (int e = 0; e < 4; ++e) t[e] = (T)second[e]; *reinterpret_cast<X4<T>*>(out2_t + m * g.ld_out2 + ncol) = t; }
        else *reinterpret_cast<f32x4*>(out2_t + m * g.ld_out2 + ncol) = second;
      }
      if (flags & EPI_OUT_F32) {
        float* o = g.out_f32 + m * g.ld_out_f32 + ncol;
        if (flags & EPI_ACCUM) { f32x4 old = *reinterpret_cast<f32x4*>(o); v += old; }
        *reinterpret_cast<f32x4*>(o) = v;
      }
      if (flags & EPI_OUT_T) {
        if constexpr (sizeof(T) == 2) { X4<T> t; for (int e = 0; e < 4; ++e) t[e] = (T)v[e]; *reinterpret_cast<X4<T>*>(out_t + m * g.ld_out_t + ncol) = t; }
        else *reinterpret_cast<f32x4*>(out_t + m * g.ld_out_t + ncol) = v;
      }
    } else {
      for (int e = 0; e < 4 && ncol + e < g.N; ++e) {
        float x = v[e];
        const int n = ncol + e;
        if (flags & EPI_BIAS) x += bias[n];
        if ((flags & EPI_ROWSHIFT) && n >= g.rs_n0 && n < g.rs_n1) x -= g.rowshift[(m / g.rs_S) * g.rs_ld + (n - g.rs_n0)];
        if (flags & EPI_GELU_BWD) x *= (float)aux[m * g.ld_aux + n];
        if (flags & EPI_RESID) x += g.resid[m * g.ld_resid + n];
        if (flags & EPI_RESID_T) x += (float)aux[m * g.ld_aux + n];
        float second = x;
        if (flags & EPI_GELU) gelu_and_grad(x, x, second);
        if (flags & EPI_OUT2_T) out2_t[m * g.ld_out2 + n] = (T)second;
        if (flags & EPI_OUT_F32) { float* o = g.out_f32 + m * g.ld_out_f32 + n; *o = (flags & EPI_ACCUM) ? (*o + x) : x; }
        if (flags & EPI_OUT_T) out_t[m * g.ld_out_t + n] = (T)x;
      }
    }
  }
}

// The epilogue of the big NT kernels, straight from the accumulators (a wave's 128 x 64 sub-tile of the tile at (m0, n0); ep_base:
// BigEpi<NWM>::BYTES of LDS free for staging).

constexpr int BIG_BN = 256;      // tile width of the LDS-DMA NT kernels (BigCfg below)
template <int NWM> struct BigEpi {
  static constexpr int EP_T = 32 * 144, EP_F = 32 * 272, EP_WAVE = (2 * EP_T > EP_F ? 2 * EP_T : EP_F), BYTES = NWM * 4 * EP_WAVE;
};
template <typename T, int FLAGS, int NWM>
PFN_DEV void nt_big_epilogue(const GemmNT& g, f32x16 (&acc)[4][2], int m0, int n0, int wave, int lane, LdsPtr ep_base, const lds_char* shift_lds = nullptr) {
  const int wm = wave >> 2, wn = wave & 3;
  const int h = lane >> 5, li = lane & 31;
  // ---- epilogue straight from the accumulators: lane = output row, 4 consecutive columns per group ----
  // Every load of the epilogue is issued BEFORE the stores that would otherwise precede it in program order.  The compiler
  // cannot prove that the outputs do not alias bias / aux / resid, so it never moves a load above a store, and gfx9 counts loads
  // and stores in one counter: written block by block (load, combine, store, next block) each of the tile's 8 blocks waited out
  // a full memory round trip PLUS the drain of the previous block's stores -- 8 serial round trips per tile, a third of a
  // K = 512 GEMM's time.  Now: the bias once, then the loads of NB blocks, then their stores (NB below).
  constexpr int flags = FLAGS;  // compile-time: one straight-line epilogue per flag combination in use
  const T* aux = reinterpret_cast<const T*>(g.aux);
  T* out_t = reinterpret_cast<T*>(g.out_t);
  T* out2_t = reinterpret_cast<T*>(g.out2_t);
  constexpr bool HAS_AUX = (flags & (EPI_GELU_BWD | EPI_RESID_T | EPI_ROWDOT)) != 0;
  constexpr bool HAS_RES = (flags & EPI_RESID) != 0;
  float rowdot = 0.f;     // EPI_ROWDOT: the lane's share of sum_n out[m, n] * aux[m, n] over the two column blocks of its row
  // blocks (32 rows x 32 columns each; 8 per wave) per pass: 4 = half the tile, 2 when an f32 residual is read (twice the
  // registers per block); the 4-wave shape (three workgroups per CU on 168 registers, latency hidden by occupancy) keeps 1
  constexpr bool EARLY = NWM == 2;     // the 4-wave shape loads each 16-byte group where it is used, as before
  constexpr int NB = !EARLY ? 1 : HAS_RES ? 2 : 4;
  f32x4 bias_v[2][4];
  if ((flags & EPI_BIAS) && EARLY) {   // once per tile
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) bias_v[j][gq] = *reinterpret_cast<const f32x4*>(g.bias + min(n0 + wn * 64 + j * 32 + 8 * gq + 4 * h, g.N - 4));
  }
  // Outputs leave through a wave-private LDS image of one 32-row x 64-column strip (both column blocks of a row block), so
  // that every global store instruction writes whole 128-byte lines (8 rows x 128 B in operand precision, 4 rows x 256 B in
  // f32) instead of 32-byte pieces of 32 different rows.  Padded rows (+16 B) keep both the lane-per-row writes and the
  // row-contiguous reads free of bank conflicts.  No workgroup barrier: a wave only re-reads what it wrote itself.
  constexpr int EP_T = 32 * 144, EP_F = 32 * 272, EP_WAVE = (2 * EP_T > EP_F ? 2 * EP_T : EP_F);
  static_assert(!(flags & EPI_ACCUM), "accumulating outputs take the generic kernel");
  LdsPtr ep = ep_base + wave * EP_WAVE;
  const bool lines = n0 + wn * 64 + 64 <= g.N && g.wide_t;   // wave-uniform: the strip is inside N and rows are 16-byte aligned
#pragma unroll
  for (int pass = 0; pass < 8 / NB; ++pass) {
    X4<T> aux_v[HAS_AUX ? NB : 1][4];
    f32x4 res_v[HAS_RES ? NB : 1][4];
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
      const int blk = pass * NB + bb, i = blk >> 1, j = blk & 1;
      if (EARLY && (HAS_AUX || HAS_RES)) {
        const long m = min((long)m0 + wm * 128 + i * 32 + li, (long)g.M - 1);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int n = min(n0 + wn * 64 + j * 32 + 8 * gq + 4 * h, g.N - 4);
          if constexpr (HAS_AUX) aux_v[bb][gq] = *reinterpret_cast<const X4<T>*>(aux + m * g.ld_aux + n);
          if constexpr (HAS_RES) res_v[bb][gq] = *reinterpret_cast<const f32x4*>(g.resid + m * g.ld_resid + n);
        }
      }
    }
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
      const int blk = pass * NB + bb, i = blk >> 1, j = blk & 1;
      const long m_row = (long)m0 + wm * 128 + i * 32 + li;
      const bool mvalid = m_row < g.M;
      const long m = mvalid ? m_row : (long)g.M - 1;     // clamped for the loads; stores are guarded
      const int nb = n0 + wn * 64 + j * 32;              // first column of this 32-column block (wave-uniform)
      float pre[16], post[16];
      // EPI_ROWSHIFT (key centring of the q|k|v projection): the lane's row selects its dataset's shift vector; rs_n0 / rs_n1 are multiples of 64, so a block is
      // shifted as a whole or not at all (wave-uniform branch)
      f32x4 shift_v[4];
      const bool shifted = (flags & EPI_ROWSHIFT) && nb >= g.rs_n0 && nb < g.rs_n1;
      if (shifted) {
        if (shift_lds) {      // the tile's shift rows wait in LDS (gemm_nt_big_kernel put them there before its main loop): no global load between this epilogue's stores
          const int which = min((int)((unsigned)m / (unsigned)g.rs_S) - (int)((unsigned)m0 / (unsigned)g.rs_S), 1);      // (32-bit: a 64-bit division is ~150 instructions, eight times per wave)
          const lds_char* sp = shift_lds + (which * BIG_BN + (nb - n0) + 4 * h) * 4;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) shift_v[gq] = __builtin_bit_cast(f32x4, lds_read16(sp + 32 * gq));
        } else {
          const float* sp = g.rowshift + (long)((unsigned)m / (unsigned)g.rs_S) * g.rs_ld + (nb - g.rs_n0) + 4 * h;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) shift_v[gq] = *reinterpret_cast<const f32x4*>(sp + 8 * gq);
        }
      }
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = min(nb + 8 * gq + 4 * h, g.N - 4);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * gq + e];
        if constexpr (!EARLY) {
          if (flags & EPI_BIAS) bias_v[j][gq] = *reinterpret_cast<const f32x4*>(g.bias + n);
          if constexpr (HAS_AUX) aux_v[bb][gq] = *reinterpret_cast<const X4<T>*>(aux + m * g.ld_aux + n);
          if constexpr (HAS_RES) res_v[bb][gq] = *reinterpret_cast<const f32x4*>(g.resid + m * g.ld_resid + n);
        }
        if (flags & EPI_BIAS) v += bias_v[j][gq];
        if (shifted) v -= shift_v[gq];
        if (flags & EPI_GELU_BWD) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= (float)aux_v[HAS_AUX ? bb : 0][gq][e];
        }
        if (flags & EPI_RESID) v += res_v[HAS_RES ? bb : 0][gq];
        if (flags & EPI_RESID_T) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)aux_v[HAS_AUX ? bb : 0][gq][e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) pre[4 * gq + e] = v[e];   // second output: value before the activation ...
        if (flags & EPI_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { float y; gelu_and_grad(v[e], y, pre[4 * gq + e]); v[e] = y; }   // ... or the GELU derivative there
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) post[4 * gq + e] = v[e];
        if (flags & EPI_ROWDOT) {      // with the values as they are stored (operand precision), like the kernel this replaces read them back
#pragma unroll
          for (int e = 0; e < 4; ++e) rowdot += (float)(T)v[e] * (float)aux_v[HAS_AUX ? bb : 0][gq][e];
        }
        if (flags & EPI_OUT_F32) {
          if (lines) lds_write16(ep + li * 272 + (j * 8 + gq * 2 + h) * 16, __builtin_bit_cast(u32x4, v));
          else if (mvalid && nb + 8 * gq + 4 * h < g.N) *reinterpret_cast<f32x4*>(g.out_f32 + m * g.ld_out_f32 + n) = v;
        }
        if (lines && (flags & (EPI_OUT_T | EPI_OUT2_T))) {
          typedef __attribute__((address_space(3))) X4<T> lds_tx4;
          if (flags & EPI_OUT_T) {
            X4<T> t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = (T)v[e];
            *reinterpret_cast<lds_tx4*>(ep + li * 144 + (j * 4 + gq) * 16 + h * 8) = t;
          }
          if (flags & EPI_OUT2_T) {
            X4<T> t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = (T)pre[4 * gq + e];
            *reinterpret_cast<lds_tx4*>(ep + EP_T + li * 144 + (j * 4 + gq) * 16 + h * 8) = t;
          }
        }
      }
      if ((flags & EPI_ROWDOT) && j == 1) {      // both column blocks of row block i are in: the other half-wave holds the other 32 of the 64 columns
        const float sum = rowdot + __shfl_xor(rowdot, 32, 64);
        rowdot = 0.f;
        if (h == 0 && mvalid) {
          const int nw = n0 + wn * 64;
          const long bq = m_row / g.rd_S, idx = (bq * g.rd_H + nw / g.rd_D) * g.rd_S + (m_row - bq * g.rd_S);
          unsafeAtomicAdd(g.rowdot + idx, sum);
          if (nw % g.rd_D == 0) g.rowdot[g.rd_lse2_off + idx] = g.rd_lse[idx] * 1.4426950408889634f;
        }
      }
      if (lines) {
        if (j == 1) {   // the strip of row block i is complete: write it out row-contiguously
          __builtin_amdgcn_wave_barrier();
          const long mb = (long)m0 + wm * 128 + i * 32;
          const int nw = n0 + wn * 64;
          if (flags & EPI_OUT_F32) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
              const int rr = it * 4 + (lane >> 4), c = lane & 15;
              const u32x4 q = lds_read16(ep + rr * 272 + c * 16);
              if (mb + rr < g.M) *reinterpret_cast<u32x4*>(g.out_f32 + (mb + rr) * g.ld_out_f32 + nw + c * 4) = q;
            }
          }
#pragma unroll
          for (int o = 0; o < 2; ++o) {
            if (!(flags & (o ? EPI_OUT2_T : EPI_OUT_T))) continue;
            T* dst = o ? out2_t : out_t;
            const long ld = o ? g.ld_out2 : g.ld_out_t;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const int rr = it * 8 + (lane >> 3), c = lane & 7;
              const u32x4 q = lds_read16(ep + o * EP_T + rr * 144 + c * 16);
              if (mb + rr < g.M) *reinterpret_cast<u32x4*>(dst + (mb + rr) * ld + nw + c * 8) = q;
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
        continue;
      }
      // operand-precision outputs: 16-byte stores after a half-wave exchange when the whole block is inside N
      // (wave-uniform test) and the row pitch keeps them aligned, else 8-byte stores per group
      if (flags & (EPI_OUT_T | EPI_OUT2_T)) {
        const bool wide = nb + 32 <= g.N && g.wide_t;
        if (wide) {
          if (flags & EPI_OUT2_T) store_row_block<T>(out2_t + m * g.ld_out2 + nb, pre, h, mvalid);
          if (flags & EPI_OUT_T) store_row_block<T>(out_t + m * g.ld_out_t + nb, post, h, mvalid);
        } else {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int n = nb + 8 * gq + 4 * h;
            if (!mvalid || n >= g.N) continue;
            if (flags & EPI_OUT2_T) {
              X4<T> t;
#pragma unroll
              for (int e = 0; e < 4; ++e) t[e] = (T)pre[4 * gq + e];
              *reinterpret_cast<X4<T>*>(out2_t + m * g.ld_out2 + n) = t;
            }
            if (flags & EPI_OUT_T) {
              X4<T> t;
#pragma unroll
              for (int e = 0; e < 4; ++e) t[e] = (T)post[4 * gq + e];
              *reinterpret_cast<X4<T>*>(out_t + m * g.ld_out_t + n) = t;
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// gemm_nt_big_kernel: the product-path NT GEMM for large token counts (bf16 only).
//   256 x 256 output tile, 64-deep contraction stages (128-byte rows), 8 waves as 2 (M) x 4 (N),
//   each wave a 128 x 64 sub-tile = 4 x 2 MFMA tiles of 32 x 32.
//   * operands go HBM -> LDS by direct LDS-DMA (global_load_lds_dwordx4): the LDS image is
//     lane-linear per wave instruction (8 rows x 128 B), so the bank-conflict swizzle of
//     load_frag_row (16-byte chunk ^ ((row >> 1) & 7)) is applied to the per-lane SOURCE address;
//   * two LDS stages (128 KiB): the DMA of stage t+1 is in flight under the MFMAs of stage t;
//   * MFMAs are issued "swapped" (weights as the A operand) so a lane owns one output row and
//     4 consecutive columns per accumulator group: the epilogue reads bias / residual / gelu'
//     inputs and writes its outputs straight from registers in 8- and 16-byte pieces.
// Requirements (checked by the launcher, otherwise gemm_nt_kernel runs): K % 64 == 0, N % 4 == 0,
// 16-byte aligned rows on every stream.
// ---------------------------------------------------------------------------------------------
// Two shapes of the same kernel (template NWM = waves along M, BK = contraction depth per stage):
//   NWM 2, BK 64 : 256 x 256 tile, 8 waves, 128 KiB LDS, one workgroup per CU  -- best per-tile rate, long K
//   NWM 1, BK 32 : 128 x 256 tile, 4 waves,  48 KiB LDS, three workgroups per CU -- the encoder's K = 512..1536
//                  GEMMs spend as long in their prologue + epilogue (first DMA, bias / GELU / residual, 64..256 MB of
//                  output) as in the MFMA loop; with three independent workgroups per CU one's stores and first
//                  loads run under another's MFMAs instead of the whole chip alternating between the two phases.
template <int NWM, int BK> struct BigCfg {
  static constexpr int BM = NWM * 128, BN = BIG_BN, NW = NWM * 4, NT = NW * 64;
  static constexpr int RB = BK * 2;                 // bytes per tile row
  static constexpr int CPR = RB / 16;               // 16-byte chunks per row
  static constexpr int RPP = 1024 / RB;             // rows per 1-KiB DMA piece
  static constexpr int PA = BM / RPP / NW, PB = BN / RPP / NW;   // pieces per wave per operand
  static constexpr int TILE_A = BM * RB, TILE_B = BN * RB, STAGE = TILE_A + TILE_B;
  static constexpr int MIN_BLOCKS = NWM == 1 ? 3 : 1;
};
constexpr int BIG_BM = 256, BIG_BK = 64;            // the large shape (launcher heuristics, tests)

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

template <typename T, int FLAGS, int NWM, int BK>
__global__ __launch_bounds__((BigCfg<NWM, BK>::NT), (BigCfg<NWM, BK>::MIN_BLOCKS)) void gemm_nt_big_kernel(GemmNT g) {
  operand_store_mode<T>();
  using C = BigCfg<NWM, BK>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr smem = lds_cast(smem_raw);

  const int tiles_n = (g.N + C::BN - 1) / C::BN;
  const int tiles_m = (g.M + C::BM - 1) / C::BM;
  const int tid_lin = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = tid_lin / tiles_n, tn = tid_lin % tiles_n;
  const int m0 = tm * C::BM, n0 = tn * C::BN;

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wm = wave >> 2, wn = wave & 3;
  const int h = lane >> 5, li = lane & 31;

  // DMA sources: wave w moves the 1-KiB pieces w, w + NW, ... of each operand tile; the bank-conflict swizzle
  // of load_frag_row goes on the source chunk (the LDS image of a piece is lane-linear)
  const T* pa[C::PA];
  const T* pb[C::PB];
#pragma unroll
  for (int i = 0; i < C::PA; ++i) {
    const int row = (wave + C::NW * i) * C::RPP + lane / C::CPR;
    const int chunk = swz16<C::RB>(row, lane % C::CPR);
    pa[i] = reinterpret_cast<const T*>(g.A) + (long)min(m0 + row, g.M - 1) * g.lda + chunk * 8;
  }
#pragma unroll
  for (int i = 0; i < C::PB; ++i) {
    const int row = (wave + C::NW * i) * C::RPP + lane / C::CPR;
    const int chunk = swz16<C::RB>(row, lane % C::CPR);
    pb[i] = reinterpret_cast<const T*>(g.B) + (long)min(n0 + row, g.N - 1) * g.ldb + chunk * 8;
  }
  auto stage = [&](int buf, int k0) {
    LdsPtr ta = smem + buf * C::STAGE + wave * 1024;
    LdsPtr tb = smem + buf * C::STAGE + C::TILE_A + wave * 1024;
#pragma unroll
    for (int i = 0; i < C::PA; ++i) __builtin_amdgcn_global_load_lds((gvoid_t*)(pa[i] + k0), (lvoid_t*)(ta + i * C::NW * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < C::PB; ++i) __builtin_amdgcn_global_load_lds((gvoid_t*)(pb[i] + k0), (lvoid_t*)(tb + i * C::NW * 1024), 16, 0, 0);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / BK;
  stage(0, 0);
  // EPI_ROWSHIFT: the (at most two) datasets' shift rows of this tile's 256 columns go to LDS behind the stage buffers, zeros outside the shifted column range -- the
  // epilogue then reads them with LDS loads instead of global ones between its stores (which serialise on the one memory counter).  Tiles of rows that span more
  // than two datasets (rs_S < the tile height) keep the global loads.
  const lds_char* shift_lds = nullptr;
  if constexpr ((FLAGS & EPI_ROWSHIFT) != 0) {
    if (g.rs_S >= C::BM && n0 < g.rs_n1 && n0 + C::BN > g.rs_n0) {
      LdsPtr img = smem + 2 * C::STAGE;
      const int b_lo = m0 / g.rs_S, b_last = (g.M - 1) / g.rs_S;
      for (int idx = threadIdx.x; idx < 2 * C::BN; idx += C::NT) {
        const int which = idx / C::BN, n = n0 + idx % C::BN, bq = min(b_lo + which, b_last);
        lds_write_f32(img + idx * 4, (n >= g.rs_n0 && n < g.rs_n1) ? g.rowshift[(long)bq * g.rs_ld + (n - g.rs_n0)] : 0.f);
      }
      shift_lds = img;
    }
  }
  __syncthreads();  // (carries the vmcnt(0) of the DMA)
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);
    const lds_char* ta = smem + cur * C::STAGE;
    const lds_char* tb = ta + C::TILE_A;
#pragma unroll
    for (int ks = 0; ks < BK; ks += 16) {
      Frag<T> fa[4], fb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = load_frag_row<T, C::RB>(tb, wn * 64 + j * 32 + li, ks);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = load_frag_row<T, C::RB>(ta, wm * 128 + i * 32 + li, ks);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mma32(fb[j], fa[i], acc[i][j]);
    }
    __syncthreads();
  }

  static_assert(BigEpi<NWM>::BYTES <= 2 * C::STAGE, "epilogue staging must fit the (dead) stage buffers");
  nt_big_epilogue<T, FLAGS, NWM>(g, acc, m0, n0, wave, lane, smem, shift_lds);
}

// ---------------------------------------------------------------------------------------------
// gemm_nt_ln_kernel: out_proj / linear2 with the residual add AND the following LayerNorm fused (GemmLN in
// pfn_kernels.h).  Tile = 128 rows x ALL N = 64 NWN columns (NWN = 2, 4 or 8 waves side by side, each 128 x 64 like
// the big kernel's waves), 32-deep LDS-DMA stages.  After the MFMA loop a lane holds 32 values of each of its 4
// rows; row sums go lane -> half-wave partner -> the NWN waves through LDS (two-pass mean / variance, like the
// standalone kernel), then the pre-LN sum (f32), the normalised operand copy (bf16) and the statistics are written
// once.  Saves, per LayerNorm, the f32 round trip of the pre-LN sum and the f32 LayerNorm output entirely.
// ---------------------------------------------------------------------------------------------
// (A note on these full-row kernels' main loop: 16-17 us per 512 of K against 10-12 us in the 256 x 256 kernel.  A is streamed from
// HBM once and only one stage of it is in flight; touching the A lines of the stage after next to pull them into L2 ahead of the
// LDS-DMA was measured and changed nothing -- 62.6 vs 63.3 us at K = 1024 -- so the stream is not latency-bound.)
template <int NWN> struct LnEpi {   // LDS of gemm_nt_ln_kernel's epilogue: row partial sums, column vectors, the output strips
  static constexpr int STRIPS = 128 * NWN * 4 + 5 * NWN * 64 * 4;
  static constexpr int WAVE = 32 * 272 + 32 * 144;
  static constexpr int BYTES = STRIPS + NWN * WAVE;
};
template <typename T, int NWN, bool RESID_LN, int BK, bool Y16>
__global__ __launch_bounds__(NWN * 64) void gemm_nt_ln_kernel(GemmLN g) {
  operand_store_mode<T>();
  constexpr int BM = 128, RB = BK * 2, CPR = RB / 16, RPP = 1024 / RB, BN = NWN * 64;
  constexpr int PA = BM / RPP / NWN > 0 ? BM / RPP / NWN : 1, PB = BN / RPP / NWN;
  constexpr int TILE_A = BM * RB, STAGE = TILE_A + BN * RB;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr smem = lds_cast(smem_raw);
  const int m0 = blockIdx.x * BM;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int h = lane >> 5, li = lane & 31;

  const T* pa[PA];
  const T* pb[PB];
  constexpr int APIECES = BM / RPP;
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int piece = wave + NWN * i;
    const int row = piece * RPP + lane / CPR;
    const int chunk = swz16<RB>(row, lane % CPR);
    pa[i] = reinterpret_cast<const T*>(g.A) + (long)min(m0 + row, g.M - 1) * g.lda + chunk * 8;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int row = (wave + NWN * i) * RPP + lane / CPR;
    const int chunk = swz16<RB>(row, lane % CPR);
    pb[i] = reinterpret_cast<const T*>(g.B) + (long)row * g.ldb + chunk * 8;
  }
  auto stage = [&](int buf, int k0) {
    LdsPtr ta = smem + buf * STAGE + wave * 1024;
    LdsPtr tb = smem + buf * STAGE + TILE_A + wave * 1024;
#pragma unroll
    for (int i = 0; i < PA; ++i)
      if (wave + NWN * i < APIECES) __builtin_amdgcn_global_load_lds((gvoid_t*)(pa[i] + k0), (lvoid_t*)(ta + i * NWN * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < PB; ++i) __builtin_amdgcn_global_load_lds((gvoid_t*)(pb[i] + k0), (lvoid_t*)(tb + i * NWN * 1024), 16, 0, 0);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / BK;
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);
    const lds_char* ta = smem + cur * STAGE;
    const lds_char* tb = ta + TILE_A;
#pragma unroll
    for (int ks = 0; ks < BK; ks += 16) {
      Frag<T> fa[4], fb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = load_frag_row<T, RB>(tb, wave * 64 + j * 32 + li, ks);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = load_frag_row<T, RB>(ta, i * 32 + li, ks);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mma32(fb[j], fa[i], acc[i][j]);
    }
    __syncthreads();
  }

  // ---- epilogue: v = acc + bias + residual (kept in the accumulator registers), row statistics, outputs ----
  // The compiler never moves a global load above a global store and waits for each group of loads where it is used, so an
  // epilogue written "load, combine, store" per 16-byte group is a chain of serialized memory round trips (measured: ~half
  // of this kernel's time at K = 512).  So: the per-column vectors go to LDS once (their reads count on lgkmcnt, not
  // vmcnt), the residual rows are fetched one 32-row block ahead of their use, and the store pass issues no global load.
  float* red = reinterpret_cast<float*>(smem_raw);          // [128 rows][NWN] partial sums; the stage buffers are dead
  float* cvec = red + 128 * NWN;                            // [5][BN]: rgamma, rbeta, bias, gamma, beta
  {
    const int n = threadIdx.x;                              // blockDim.x == BN
    float c0 = 0.f, c1 = 0.f;
    if constexpr (RESID_LN) { c0 = g.rgamma[n]; c1 = g.rbeta[n]; }
    const float c2 = g.bias[n], c3 = g.gamma[n], c4 = g.beta[n];
    cvec[n] = c0; cvec[BN + n] = c1; cvec[2 * BN + n] = c2; cvec[3 * BN + n] = c3; cvec[4 * BN + n] = c4;
  }
  // Y16 (GemmLN::y16, fp16 operands): the pre-LN sums are stored -- and the previous LayerNorm's read back -- in operand precision, 2 instead of 4 bytes per
  // element on the two f32 streams of this HBM-bound epilogue; statistics, outputs and the LayerNorm itself still come from the f32 values in the registers.
  typedef typename std::conditional<Y16 && RESID_LN, X4<T>, f32x4>::type RV;      // a residual group as it is fetched
  const float* rsrc = RESID_LN ? reinterpret_cast<const float*>(g.ry) : g.resid;
  // (32-bit row indices, the 64-bit element offsets formed where they are used; the residual rows' LayerNorm statistics travel with the rows, one block
  // ahead, instead of all four blocks' up front: this epilogue sits at the 256-register limit and a handful of long-lived values is a spill)
  float rm[2], rr[2];
  int mrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) mrow[i] = min(m0 + i * 32 + li, g.M - 1);
  RV rv[2][8];
  auto fetch_rows = [&](int i, RV (&dst)[8]) {
    rm[i & 1] = 0.f; rr[i & 1] = 1.f;
    if constexpr (RESID_LN) { rm[i & 1] = g.rmean[mrow[i]]; rr[i & 1] = g.rrstd[mrow[i]]; }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const long off = (long)mrow[i] * BN + wave * 64 + j * 32 + 8 * gq + 4 * h;
        if constexpr (Y16 && RESID_LN) dst[j * 4 + gq] = *reinterpret_cast<const X4<T>*>(reinterpret_cast<const T*>(g.ry) + off);
        else dst[j * 4 + gq] = *reinterpret_cast<const f32x4*>(rsrc + off);
      }
  };
  fetch_rows(0, rv[0]);
  __syncthreads();                                          // cvec visible
  float mean[4], rstd[4];
  const float invN = 1.f / (float)BN;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i + 1 < 4) fetch_rows(i + 1, rv[(i + 1) & 1]);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = wave * 64 + j * 32 + 8 * gq + 4 * h;
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (float)rv[i & 1][j * 4 + gq][e];
        if constexpr (RESID_LN) {
          const f32x4 ga = *reinterpret_cast<const f32x4*>(cvec + n), be = *reinterpret_cast<const f32x4*>(cvec + BN + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] = (r[e] - rm[i & 1]) * rr[i & 1] * ga[e] + be[e];
        }
        const f32x4 bi = *reinterpret_cast<const f32x4*>(cvec + 2 * BN + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[i][j][4 * gq + e] + bi[e] + r[e];
          acc[i][j][4 * gq + e] = v;
          s += v;
        }
      }
    s += __shfl_xor(s, 32, 64);
    if (h == 0) red[(i * 32 + li) * NWN + wave] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWN; ++w) t += red[(i * 32 + li) * NWN + w];
    mean[i] = t * invN;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = acc[i][j][r] - mean[i]; s += d * d; }
    s += __shfl_xor(s, 32, 64);
    if (h == 0) red[(i * 32 + li) * NWN + wave] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWN; ++w) t += red[(i * 32 + li) * NWN + w];
    rstd[i] = rsqrtf(t * invN + g.eps);
  }
  // Outputs leave through wave-private LDS strips (one 32-row block at a time: f32 rows of 256 B + operand-precision rows of
  // 128 B, padded by 16 B) so that every global store instruction writes whole lines -- 4 rows x 256 B of y, 8 rows x 128 B of
  // x_t -- instead of 32 bytes of 32 different rows (as in nt_big_epilogue).
  typedef __attribute__((address_space(3))) X4<T> lds_tx4;
  T* x_t = reinterpret_cast<T*>(g.x_t);
  LdsPtr sf = smem + LnEpi<NWN>::STRIPS + wave * LnEpi<NWN>::WAVE, st = sf + 32 * 272;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool mvalid = m0 + i * 32 + li < g.M;
    const long m = mrow[i];
    if (mvalid && wave == 0 && h == 0) { g.mean[m] = mean[i]; g.rstd[m] = rstd[i]; }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nb = wave * 64 + j * 32;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = nb + 8 * gq + 4 * h;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(cvec + 3 * BN + n), be = *reinterpret_cast<const f32x4*>(cvec + 4 * BN + n);
        f32x4 v, xo;
        X4<T> xt;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[i][j][4 * gq + e];
          xo[e] = (v[e] - mean[i]) * rstd[i] * ga[e] + be[e];
          xt[e] = (T)xo[e];
        }
        if constexpr (Y16) {
          X4<T> vt;
#pragma unroll
          for (int e = 0; e < 4; ++e) vt[e] = (T)v[e];
          *reinterpret_cast<lds_tx4*>(sf + li * 144 + (j * 4 + gq) * 16 + h * 8) = vt;
        } else {
          lds_write16(sf + li * 272 + (j * 8 + gq * 2 + h) * 16, __builtin_bit_cast(u32x4, v));
        }
        *reinterpret_cast<lds_tx4*>(st + li * 144 + (j * 4 + gq) * 16 + h * 8) = xt;
        if (g.x_f32 && mvalid) *reinterpret_cast<f32x4*>(g.x_f32 + m * BN + n) = xo;   // (last layer only)
      }
    }
    __builtin_amdgcn_wave_barrier();
    const long mb = (long)m0 + i * 32;
    if constexpr (Y16) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rr = it * 8 + (lane >> 3), c = lane & 7;
        const u32x4 q = lds_read16(sf + rr * 144 + c * 16);
        if (mb + rr < g.M) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(g.y) + (mb + rr) * BN + wave * 64 + c * 8) = q;
      }
    } else {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int rr = it * 4 + (lane >> 4), c = lane & 15;
        const u32x4 q = lds_read16(sf + rr * 272 + c * 16);
        if (mb + rr < g.M) *reinterpret_cast<u32x4*>(reinterpret_cast<float*>(g.y) + (mb + rr) * BN + wave * 64 + c * 4) = q;
      }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rr = it * 8 + (lane >> 3), c = lane & 7;
      const u32x4 q = lds_read16(st + rr * 144 + c * 16);
      if (mb + rr < g.M) *reinterpret_cast<u32x4*>(x_t + (mb + rr) * BN + wave * 64 + c * 8) = q;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// gemm_nt_ln_wide_kernel (round 3): the same kernel for emsize 1024 (BASELINE configs[4]).  The tile is MI x 32 rows by NWN waves x
// (NJ x 32) columns; (MI, NJ) = (2, 4) gives 64 rows x 1024 columns with the same 8 accumulator blocks per wave as the 128 x 512 tile
// above, 32-deep stages (two of them: 136 KiB).  The epilogue walks "units" of 32 rows x 64 columns (one row block, two column
// blocks; four per wave, like the four row blocks above).  Kept as its own kernel: the generalised source costs the N <= 512
// instantiation three spilled registers.
template <int NWN, int MI, int NJ> struct LnEpiW {   // LDS of gemm_nt_ln_kernel's epilogue: row partial sums, column vectors, the output strips
  static constexpr int STRIPS = MI * 32 * NWN * 4 + 5 * NWN * NJ * 32 * 4;
  static constexpr int WAVE = 32 * 272 + 32 * 144;
  static constexpr int BYTES = STRIPS + NWN * WAVE;
};
template <typename T, int NWN, bool RESID_LN, int BK, int MI, int NJ, bool Y16>
__global__ __launch_bounds__(NWN * 64, (NWN <= 4 ? 2 : 1)) void gemm_nt_ln_wide_kernel(GemmLN g) {      // (4-wave tiles: two workgroups per CU, 256 registers per wave)
  operand_store_mode<T>();
  constexpr int BM = MI * 32, WN = NJ * 32, RB = BK * 2, CPR = RB / 16, RPP = 1024 / RB, BN = NWN * WN;
  constexpr int PA = BM / RPP / NWN > 0 ? BM / RPP / NWN : 1, PB = BN / RPP / NWN;
  constexpr int TILE_A = BM * RB, STAGE = TILE_A + BN * RB;
  constexpr int JH = NJ / 2, NU = MI * JH;                  // units of 32 rows x 64 columns per wave
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr smem = lds_cast(smem_raw);
  const int m0 = blockIdx.x * BM;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int h = lane >> 5, li = lane & 31;

  const T* pa[PA];
  const T* pb[PB];
  constexpr int APIECES = BM / RPP;
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int piece = wave + NWN * i;
    const int row = piece * RPP + lane / CPR;
    const int chunk = swz16<RB>(row, lane % CPR);
    pa[i] = reinterpret_cast<const T*>(g.A) + (long)min(m0 + row, g.M - 1) * g.lda + chunk * 8;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int row = (wave + NWN * i) * RPP + lane / CPR;
    const int chunk = swz16<RB>(row, lane % CPR);
    pb[i] = reinterpret_cast<const T*>(g.B) + (long)row * g.ldb + chunk * 8;
  }
  auto stage = [&](int buf, int k0) {
    LdsPtr ta = smem + buf * STAGE + wave * 1024;
    LdsPtr tb = smem + buf * STAGE + TILE_A + wave * 1024;
#pragma unroll
    for (int i = 0; i < PA; ++i)
      if (wave + NWN * i < APIECES) __builtin_amdgcn_global_load_lds((gvoid_t*)(pa[i] + k0), (lvoid_t*)(ta + i * NWN * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < PB; ++i) __builtin_amdgcn_global_load_lds((gvoid_t*)(pb[i] + k0), (lvoid_t*)(tb + i * NWN * 1024), 16, 0, 0);
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / BK;
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);
    const lds_char* ta = smem + cur * STAGE;
    const lds_char* tb = ta + TILE_A;
#pragma unroll
    for (int ks = 0; ks < BK; ks += 16) {
      Frag<T> fa[MI], fb[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) fb[j] = load_frag_row<T, RB>(tb, wave * WN + j * 32 + li, ks);
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = load_frag_row<T, RB>(ta, i * 32 + li, ks);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mma32(fb[j], fa[i], acc[i][j]);
    }
    __syncthreads();
  }

  // ---- epilogue: v = acc + bias + residual (kept in the accumulator registers), row statistics, outputs ----
  // The compiler never moves a global load above a global store and waits for each group of loads where it is used, so an
  // epilogue written "load, combine, store" per 16-byte group is a chain of serialized memory round trips (measured: ~half
  // of this kernel's time at K = 512).  So: the per-column vectors go to LDS once (their reads count on lgkmcnt, not
  // vmcnt), the residual rows are fetched one unit ahead of their use, and the store pass issues no global load.
  float* red = reinterpret_cast<float*>(smem_raw);          // [BM rows][NWN] partial sums; the stage buffers are dead
  float* cvec = red + BM * NWN;                             // [5][BN]: rgamma, rbeta, bias, gamma, beta
  for (int n = threadIdx.x; n < BN; n += NWN * 64) {
    float c0 = 0.f, c1 = 0.f;
    if constexpr (RESID_LN) { c0 = g.rgamma[n]; c1 = g.rbeta[n]; }
    const float c2 = g.bias[n], c3 = g.gamma[n], c4 = g.beta[n];
    cvec[n] = c0; cvec[BN + n] = c1; cvec[2 * BN + n] = c2; cvec[3 * BN + n] = c3; cvec[4 * BN + n] = c4;
  }
  typedef typename std::conditional<Y16 && RESID_LN, X4<T>, f32x4>::type RV;      // (Y16: see gemm_nt_ln_kernel)
  const float* rsrc = RESID_LN ? reinterpret_cast<const float*>(g.ry) : g.resid;
  float rm[MI], rr[MI];
  long mrow[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    mrow[i] = min((long)m0 + i * 32 + li, (long)g.M - 1);
    rm[i] = 0.f; rr[i] = 1.f;
    if constexpr (RESID_LN) { rm[i] = g.rmean[mrow[i]]; rr[i] = g.rrstd[mrow[i]]; }
  }
  RV rv[2][8];
  auto fetch_unit = [&](int u, RV (&dst)[8]) {              // unit u = (row block u / JH, column blocks 2 (u % JH) and + 1)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const long off = mrow[u / JH] * BN + wave * WN + ((u % JH) * 2 + jj) * 32 + 8 * gq + 4 * h;
        if constexpr (Y16 && RESID_LN) dst[jj * 4 + gq] = *reinterpret_cast<const X4<T>*>(reinterpret_cast<const T*>(g.ry) + off);
        else dst[jj * 4 + gq] = *reinterpret_cast<const f32x4*>(rsrc + off);
      }
  };
  fetch_unit(0, rv[0]);
  __syncthreads();                                          // cvec visible
  float mean[MI], rstd[MI];
  const float invN = 1.f / (float)BN;
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = u / JH, jh = u % JH;
    if (u + 1 < NU) fetch_unit(u + 1, rv[(u + 1) & 1]);
    if (jh == 0) s = 0.f;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int j = jh * 2 + jj;
        const int n = wave * WN + j * 32 + 8 * gq + 4 * h;
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (float)rv[u & 1][jj * 4 + gq][e];
        if constexpr (RESID_LN) {
          const f32x4 ga = *reinterpret_cast<const f32x4*>(cvec + n), be = *reinterpret_cast<const f32x4*>(cvec + BN + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] = (r[e] - rm[i]) * rr[i] * ga[e] + be[e];
        }
        const f32x4 bi = *reinterpret_cast<const f32x4*>(cvec + 2 * BN + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[i][j][4 * gq + e] + bi[e] + r[e];
          acc[i][j][4 * gq + e] = v;
          s += v;
        }
      }
    if (jh == JH - 1) {
      s += __shfl_xor(s, 32, 64);
      if (h == 0) red[(i * 32 + li) * NWN + wave] = s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWN; ++w) t += red[(i * 32 + li) * NWN + w];
    mean[i] = t * invN;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = acc[i][j][r] - mean[i]; sq += d * d; }
    sq += __shfl_xor(sq, 32, 64);
    if (h == 0) red[(i * 32 + li) * NWN + wave] = sq;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWN; ++w) t += red[(i * 32 + li) * NWN + w];
    rstd[i] = rsqrtf(t * invN + g.eps);
  }
  // Outputs leave through wave-private LDS strips (one unit at a time: f32 rows of 256 B + operand-precision rows of
  // 128 B, padded by 16 B) so that every global store instruction writes whole lines -- 4 rows x 256 B of y, 8 rows x 128 B of
  // x_t -- instead of 32 bytes of 32 different rows (as in nt_big_epilogue).
  typedef __attribute__((address_space(3))) X4<T> lds_tx4;
  T* x_t = reinterpret_cast<T*>(g.x_t);
  LdsPtr sf = smem + LnEpiW<NWN, MI, NJ>::STRIPS + wave * LnEpiW<NWN, MI, NJ>::WAVE, st = sf + 32 * 272;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = u / JH, jh = u % JH;
    const long m_row = (long)m0 + i * 32 + li;
    const bool mvalid = m_row < g.M;
    const long m = mrow[i];
    if (jh == 0 && mvalid && wave == 0 && h == 0) { g.mean[m] = mean[i]; g.rstd[m] = rstd[i]; }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = jh * 2 + jj;
      const int nb = wave * WN + j * 32;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = nb + 8 * gq + 4 * h;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(cvec + 3 * BN + n), be = *reinterpret_cast<const f32x4*>(cvec + 4 * BN + n);
        f32x4 v, xo;
        X4<T> xt;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[i][j][4 * gq + e];
          xo[e] = (v[e] - mean[i]) * rstd[i] * ga[e] + be[e];
          xt[e] = (T)xo[e];
        }
        if constexpr (Y16) {
          X4<T> vt;
#pragma unroll
          for (int e = 0; e < 4; ++e) vt[e] = (T)v[e];
          *reinterpret_cast<lds_tx4*>(sf + li * 144 + (jj * 4 + gq) * 16 + h * 8) = vt;
        } else {
          lds_write16(sf + li * 272 + (jj * 8 + gq * 2 + h) * 16, __builtin_bit_cast(u32x4, v));
        }
        *reinterpret_cast<lds_tx4*>(st + li * 144 + (jj * 4 + gq) * 16 + h * 8) = xt;
        if (g.x_f32 && mvalid) *reinterpret_cast<f32x4*>(g.x_f32 + m * BN + n) = xo;   // (last layer only)
      }
    }
    __builtin_amdgcn_wave_barrier();
    const long mb = (long)m0 + i * 32;
    const int cb = wave * WN + jh * 64;                     // first column of the unit
    if constexpr (Y16) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rr_ = it * 8 + (lane >> 3), c = lane & 7;
        const u32x4 q = lds_read16(sf + rr_ * 144 + c * 16);
        if (mb + rr_ < g.M) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(g.y) + (mb + rr_) * BN + cb + c * 8) = q;
      }
    } else {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int rr_ = it * 4 + (lane >> 4), c = lane & 15;
        const u32x4 q = lds_read16(sf + rr_ * 272 + c * 16);
        if (mb + rr_ < g.M) *reinterpret_cast<u32x4*>(reinterpret_cast<float*>(g.y) + (mb + rr_) * BN + cb + c * 4) = q;
      }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rr_ = it * 8 + (lane >> 3), c = lane & 7;
      const u32x4 q = lds_read16(st + rr_ * 144 + c * 16);
      if (mb + rr_ < g.M) *reinterpret_cast<u32x4*>(x_t + (mb + rr_) * BN + cb + c * 8) = q;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// gemm_nt_lnbwd_kernel: a data-gradient GEMM whose output is the gradient w.r.t. a LayerNorm OUTPUT, with that LayerNorm's
// backward fused into the epilogue (GemmLNB in pfn_kernels.h):
//     v  = A . B^T + aux                                  (dx1 = dh . W1 + dy2   /   dx = dqkv . Win + dy1)
//     dx = rstd (gamma v - mean_n(gamma v) - xhat mean_n(gamma v xhat)),   xhat = (y - mean) rstd
//     dgamma += sum_rows v xhat,   dbeta += sum_rows v
// Same tile as gemm_nt_ln_kernel (128 rows x all N = 64 NWN columns, so a workgroup owns whole rows).  v never reaches HBM:
// the separate LayerNorm-backward kernel read it back (32.8 MB per launch at the north-star shape) together with y, and this
// GEMM wrote it.  xhat is needed twice (row sums, then dx); between the two it waits in LDS in operand precision (the stage
// buffers are dead), which only touches the xhat . mean_n(.) term of dx and the dgamma sum -- below the rounding of dx itself.
// The bias gradient of the Linear in front of the LayerNorm (column sums of dx) is left to the weight-gradient GEMM that
// consumes dx as its operand (gemm_tn_big_kernel's colsum).
// ---------------------------------------------------------------------------------------------
PFN_DEV float row16_sum(float v) {   // sum over the 16 lanes of a DPP row, left in every lane of the row
  auto sh = [](float x, auto ctrl) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true)); };
  v += sh(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v += sh(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v += sh(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  v += sh(v, std::integral_constant<int, 0x140>{});   // row_mirror
  return v;
}

template <int NWN, int BK> struct LnbCfg {
  static constexpr int BN = NWN * 64;
  static constexpr int STAGES = 2 * (128 + BN) * (BK * 2);
  static constexpr int RED = 128 * NWN * 4;              // one [128 rows][NWN] f32 array of row partial sums
  static constexpr int STRIP = 32 * 144;                 // a 32-row x 64-column strip in operand precision, rows padded by 16 B
  static constexpr int STASH = 2 * RED + BN * 4;         // offset of the per-wave xhat / dx strips
  static constexpr int EPI = STASH + NWN * 4 * STRIP;
  static constexpr int LDS = STAGES > EPI ? STAGES : EPI;
};

template <typename T, int NWN, int BK, bool Y16>
__global__ __launch_bounds__(NWN * 64) void gemm_nt_lnbwd_kernel(GemmLNB g) {
  operand_store_mode<T>();
  using C = LnbCfg<NWN, BK>;
  constexpr int BM = 128, RB = BK * 2, CPR = RB / 16, RPP = 1024 / RB, BN = NWN * 64;
  constexpr int PA = BM / RPP / NWN > 0 ? BM / RPP / NWN : 1, PB = BN / RPP / NWN;
  constexpr int TILE_A = BM * RB, STAGE = TILE_A + BN * RB;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr smem = lds_cast(smem_raw);
  const int m0 = blockIdx.x * BM;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int h = lane >> 5, li = lane & 31;

  const T* pa[PA];
  const T* pb[PB];
  constexpr int APIECES = BM / RPP;
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = (wave + NWN * i) * RPP + lane / CPR;
    pa[i] = reinterpret_cast<const T*>(g.A) + (long)min(m0 + row, g.M - 1) * g.lda + swz16<RB>(row, lane % CPR) * 8;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int row = (wave + NWN * i) * RPP + lane / CPR;
    pb[i] = reinterpret_cast<const T*>(g.B) + (long)row * g.ldb + swz16<RB>(row, lane % CPR) * 8;
  }
  auto stage = [&](int buf, int k0) {
    LdsPtr ta = smem + buf * STAGE + wave * 1024;
    LdsPtr tb = smem + buf * STAGE + TILE_A + wave * 1024;
#pragma unroll
    for (int i = 0; i < PA; ++i)
      if (wave + NWN * i < APIECES) __builtin_amdgcn_global_load_lds((gvoid_t*)(pa[i] + k0), (lvoid_t*)(ta + i * NWN * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < PB; ++i) __builtin_amdgcn_global_load_lds((gvoid_t*)(pb[i] + k0), (lvoid_t*)(tb + i * NWN * 1024), 16, 0, 0);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / BK;
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);
    const lds_char* ta = smem + cur * STAGE;
    const lds_char* tb = ta + TILE_A;
#pragma unroll
    for (int ks = 0; ks < BK; ks += 16) {
      Frag<T> fa[4], fb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = load_frag_row<T, RB>(tb, wave * 64 + j * 32 + li, ks);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = load_frag_row<T, RB>(ta, i * 32 + li, ks);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mma32(fb[j], fa[i], acc[i][j]);
    }
    __syncthreads();
  }

  // ---- epilogue ----
  typedef __attribute__((address_space(3))) X4<T> lds_tx4;
  float* red1 = reinterpret_cast<float*>(smem_raw);
  float* red2 = red1 + 128 * NWN;
  float* gam = red2 + 128 * NWN;
  LdsPtr stash = smem + C::STASH + wave * 4 * C::STRIP;
  const T* aux = reinterpret_cast<const T*>(g.aux);
  T* dx_t = reinterpret_cast<T*>(g.dx_t);
  gam[threadIdx.x] = g.gamma[threadIdx.x];                  // blockDim.x == BN
  int mrow[4];
  bool mvalid[4];
  float mu[4], rs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + i * 32 + li;
    mvalid[i] = m < g.M;
    mrow[i] = mvalid[i] ? m : g.M - 1;
    mu[i] = g.mean[mrow[i]]; rs[i] = g.rstd[mrow[i]];
  }
  // y rows and the residual-branch gradient: one half row block (a 32-column block of the lane's row) ahead of its use -- a whole
  // block ahead does not fit the registers.  v = product + that gradient replaces the accumulator; rows past M hold zeros.
  typedef typename std::conditional<Y16, X4<T>, f32x4>::type YV;      // (GemmLNB::y16: the LayerNorm's input rows as gemm_nt_ln_kernel<..., Y16> stored them)
  YV yv[2][4];
  X4<T> av[2][4];
  auto fetch_half = [&](int hb, YV (&dy)[4], X4<T> (&da)[4]) {              // hb = 2 i + j
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const long off = (long)mrow[hb >> 1] * BN + wave * 64 + (hb & 1) * 32 + 8 * gq + 4 * h;
      if constexpr (Y16) dy[gq] = *reinterpret_cast<const X4<T>*>(reinterpret_cast<const T*>(g.y) + off);
      else dy[gq] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.y) + off);
      da[gq] = *reinterpret_cast<const X4<T>*>(aux + off);
    }
  };
  fetch_half(0, yv[0], av[0]);
  __syncthreads();                                          // gam visible
  const float invN = 1.f / (float)BN;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int hb = 0; hb < 8; ++hb) {
    const int i = hb >> 1, j = hb & 1;
    asm volatile("" ::: "memory");                          // re-read gamma from LDS per block (kept in registers across blocks it costs 32 of them)
    if (hb + 1 < 8) fetch_half(hb + 1, yv[(hb + 1) & 1], av[(hb + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);                      // (the scheduler would hoist the loads of every later block up here: spills)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int n = wave * 64 + j * 32 + 8 * gq + 4 * h;
      const f32x4 ga = *reinterpret_cast<const f32x4*>(gam + n);
      X4<T> xh_t;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = mvalid[i] ? acc[i][j][4 * gq + e] + (float)av[hb & 1][gq][e] : 0.f;
        acc[i][j][4 * gq + e] = v;
        const float xh = ((float)yv[hb & 1][gq][e] - mu[i]) * rs[i];
        const float gv = ga[e] * v;
        s1 += gv;
        s2 += gv * xh;
        xh_t[e] = (T)xh;
      }
      *reinterpret_cast<lds_tx4*>(stash + i * C::STRIP + li * 144 + (j * 4 + gq) * 16 + h * 8) = xh_t;
    }
    if (j == 1) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (h == 0) { red1[(i * 32 + li) * NWN + wave] = s1; red2[(i * 32 + li) * NWN + wave] = s2; }
      s1 = 0.f; s2 = 0.f;
    }
  }
  __syncthreads();
  float m1[4], m2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < NWN; ++w) { t1 += red1[(i * 32 + li) * NWN + w]; t2 += red2[(i * 32 + li) * NWN + w]; }
    m1[i] = t1 * invN; m2[i] = t2 * invN;
  }
  // dbeta / dgamma: column sums over this tile's rows -- the four row blocks inside the lane, the 16 lanes of a DPP row, then
  // through LDS ([quantity][row half][column], the space of the row sums) so that the workgroup ends with ONE 64-lane atomic
  // instruction per 64 columns (issued by the lanes that hold the sums -- 4 per instruction -- the same atomics took 270 us).
  __syncthreads();                                          // every wave has read red1 / red2
  float* csum = red1;
  const int ccol = wave * 64 + 4 * h, chalf = (li >> 4) * BN;
  {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        f32x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * gq + e;
          t[e] = row16_sum(acc[0][j][r] + acc[1][j][r] + acc[2][j][r] + acc[3][j][r]);
        }
        if ((li & 15) == 0) *reinterpret_cast<f32x4*>(csum + chalf + ccol + j * 32 + 8 * gq) = t;
      }
  }
  float cg[2][16];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) cg[j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    LdsPtr strip = stash + i * C::STRIP;
    asm volatile("" ::: "memory");
    X4<T> xh_t[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) xh_t[c] = *reinterpret_cast<const lds_tx4*>(strip + li * 144 + c * 16 + h * 8);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = wave * 64 + j * 32 + 8 * gq + 4 * h;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gam + n);
        X4<T> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (float)xh_t[j * 4 + gq][e], v = acc[i][j][4 * gq + e];
          cg[j][4 * gq + e] += v * xh;
          o[e] = (T)(rs[i] * (ga[e] * v - m1[i] - xh * m2[i]));
        }
        *reinterpret_cast<lds_tx4*>(strip + li * 144 + (j * 4 + gq) * 16 + h * 8) = o;   // (the lane's own xhat slot)
      }
    __builtin_amdgcn_wave_barrier();
    const long mb = (long)m0 + i * 32;
#pragma unroll
    for (int it = 0; it < 4; ++it) {                        // whole 128-byte lines: 8 rows x 128 B per store instruction
      const int rr = it * 8 + (lane >> 3), c = lane & 7;
      const u32x4 q = lds_read16(strip + rr * 144 + c * 16);
      if (mb + rr < g.M) *reinterpret_cast<u32x4*>(dx_t + (mb + rr) * BN + wave * 64 + c * 8) = q;
    }
    // (else the dgamma additions are sunk to the end of the kernel and every block's 32 products stay in registers: spills)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(cg[j][r]));
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      f32x4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = row16_sum(cg[j][4 * gq + e]);
      if ((li & 15) == 0) *reinterpret_cast<f32x4*>(csum + 2 * BN + chalf + ccol + j * 32 + 8 * gq) = t;
    }
  __syncthreads();
  {
    const int n = threadIdx.x;                              // blockDim.x == BN
    const float osc = loss_scale_down(g.scale_amax);      // fp16 backward: the parameter gradients leave unscaled (pfn_device.h)
    unsafeAtomicAdd(g.dbeta + n, (csum[n] + csum[BN + n]) * osc);
    unsafeAtomicAdd(g.dgamma + n, (csum[2 * BN + n] + csum[3 * BN + n]) * osc);
  }
}

// gemm_nt_lnbwd_wide_kernel (round 3): the same kernel on 64-row x 1024-column tiles (emsize 1024, BASELINE configs[4]); see gemm_nt_ln_wide_kernel.
template <int NWN, int BK, int MI, int NJ> struct LnbCfgW {
  static constexpr int BM = MI * 32, BN = NWN * NJ * 32, NU = MI * (NJ / 2);
  static constexpr int STAGES = 2 * (BM + BN) * (BK * 2);
  static constexpr int RED = BM * NWN * 4;               // one [BM rows][NWN] f32 array of row partial sums
  static constexpr int STRIP = 32 * 144;                 // a 32-row x 64-column strip in operand precision, rows padded by 16 B
  static constexpr int CSUM = 2 * RED + BN * 4;          // [dbeta | dgamma][BN] column sums (the 128-row kernel re-uses the row-sum arrays: too small here)
  static constexpr int STASH = CSUM + 2 * BN * 4;        // offset of the per-wave xhat / dx strips
  static constexpr int EPI = STASH + NWN * NU * STRIP;
  static constexpr int LDS = STAGES > EPI ? STAGES : EPI;
};

template <typename T, int NWN, int BK, int MI, int NJ, bool Y16>
__global__ __launch_bounds__(NWN * 64, (NWN <= 4 ? 2 : 1)) void gemm_nt_lnbwd_wide_kernel(GemmLNB g) {
  operand_store_mode<T>();
  using C = LnbCfgW<NWN, BK, MI, NJ>;
  constexpr int BM = MI * 32, WN = NJ * 32, RB = BK * 2, CPR = RB / 16, RPP = 1024 / RB, BN = NWN * WN;
  constexpr int JH = NJ / 2, NU = MI * JH, NHB = MI * NJ;
  constexpr int PA = BM / RPP / NWN > 0 ? BM / RPP / NWN : 1, PB = BN / RPP / NWN;
  constexpr int TILE_A = BM * RB, STAGE = TILE_A + BN * RB;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr smem = lds_cast(smem_raw);
  const int m0 = blockIdx.x * BM;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int h = lane >> 5, li = lane & 31;

  const T* pa[PA];
  const T* pb[PB];
  constexpr int APIECES = BM / RPP;
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = (wave + NWN * i) * RPP + lane / CPR;
    pa[i] = reinterpret_cast<const T*>(g.A) + (long)min(m0 + row, g.M - 1) * g.lda + swz16<RB>(row, lane % CPR) * 8;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int row = (wave + NWN * i) * RPP + lane / CPR;
    pb[i] = reinterpret_cast<const T*>(g.B) + (long)row * g.ldb + swz16<RB>(row, lane % CPR) * 8;
  }
  auto stage = [&](int buf, int k0) {
    LdsPtr ta = smem + buf * STAGE + wave * 1024;
    LdsPtr tb = smem + buf * STAGE + TILE_A + wave * 1024;
#pragma unroll
    for (int i = 0; i < PA; ++i)
      if (wave + NWN * i < APIECES) __builtin_amdgcn_global_load_lds((gvoid_t*)(pa[i] + k0), (lvoid_t*)(ta + i * NWN * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < PB; ++i) __builtin_amdgcn_global_load_lds((gvoid_t*)(pb[i] + k0), (lvoid_t*)(tb + i * NWN * 1024), 16, 0, 0);
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / BK;
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);
    const lds_char* ta = smem + cur * STAGE;
    const lds_char* tb = ta + TILE_A;
#pragma unroll
    for (int ks = 0; ks < BK; ks += 16) {
      Frag<T> fa[MI], fb[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) fb[j] = load_frag_row<T, RB>(tb, wave * WN + j * 32 + li, ks);
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = load_frag_row<T, RB>(ta, i * 32 + li, ks);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mma32(fb[j], fa[i], acc[i][j]);
    }
    __syncthreads();
  }

  // ---- epilogue ----
  typedef __attribute__((address_space(3))) X4<T> lds_tx4;
  float* red1 = reinterpret_cast<float*>(smem_raw);
  float* red2 = red1 + BM * NWN;
  float* gam = red2 + BM * NWN;
  LdsPtr stash = smem + C::STASH + wave * NU * C::STRIP;
  const T* aux = reinterpret_cast<const T*>(g.aux);
  T* dx_t = reinterpret_cast<T*>(g.dx_t);
  for (int n = threadIdx.x; n < BN; n += NWN * 64) gam[n] = g.gamma[n];
  int mrow[MI];
  bool mvalid[MI];
  float mu[MI], rs[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + i * 32 + li;
    mvalid[i] = m < g.M;
    mrow[i] = mvalid[i] ? m : g.M - 1;
    mu[i] = g.mean[mrow[i]]; rs[i] = g.rstd[mrow[i]];
  }
  // y rows and the residual-branch gradient: one half row block (a 32-column block of the lane's row) ahead of its use -- a whole
  // block ahead does not fit the registers.  v = product + that gradient replaces the accumulator; rows past M hold zeros.
  typedef typename std::conditional<Y16, X4<T>, f32x4>::type YV;
  YV yv[2][4];
  X4<T> av[2][4];
  auto fetch_half = [&](int hb, YV (&dy)[4], X4<T> (&da)[4]) {              // hb = NJ i + j
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const long off = (long)mrow[hb / NJ] * BN + wave * WN + (hb % NJ) * 32 + 8 * gq + 4 * h;
      if constexpr (Y16) dy[gq] = *reinterpret_cast<const X4<T>*>(reinterpret_cast<const T*>(g.y) + off);
      else dy[gq] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.y) + off);
      da[gq] = *reinterpret_cast<const X4<T>*>(aux + off);
    }
  };
  fetch_half(0, yv[0], av[0]);
  __syncthreads();                                          // gam visible
  const float invN = 1.f / (float)BN;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int hb = 0; hb < NHB; ++hb) {
    const int i = hb / NJ, j = hb % NJ;
    asm volatile("" ::: "memory");                          // re-read gamma from LDS per block (kept in registers across blocks it costs 32 of them)
    if (hb + 1 < NHB) fetch_half(hb + 1, yv[(hb + 1) & 1], av[(hb + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);                      // (the scheduler would hoist the loads of every later block up here: spills)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int n = wave * WN + j * 32 + 8 * gq + 4 * h;
      const f32x4 ga = *reinterpret_cast<const f32x4*>(gam + n);
      X4<T> xh_t;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = mvalid[i] ? acc[i][j][4 * gq + e] + (float)av[hb & 1][gq][e] : 0.f;
        acc[i][j][4 * gq + e] = v;
        const float xh = ((float)yv[hb & 1][gq][e] - mu[i]) * rs[i];
        const float gv = ga[e] * v;
        s1 += gv;
        s2 += gv * xh;
        xh_t[e] = (T)xh;
      }
      *reinterpret_cast<lds_tx4*>(stash + (i * JH + j / 2) * C::STRIP + li * 144 + ((j & 1) * 4 + gq) * 16 + h * 8) = xh_t;
    }
    if (j == NJ - 1) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (h == 0) { red1[(i * 32 + li) * NWN + wave] = s1; red2[(i * 32 + li) * NWN + wave] = s2; }
      s1 = 0.f; s2 = 0.f;
    }
  }
  __syncthreads();
  float m1[MI], m2[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < NWN; ++w) { t1 += red1[(i * 32 + li) * NWN + w]; t2 += red2[(i * 32 + li) * NWN + w]; }
    m1[i] = t1 * invN; m2[i] = t2 * invN;
  }
  // dbeta / dgamma: column sums over this tile's rows -- the four row blocks inside the lane, the 16 lanes of a DPP row, then
  // through LDS ([quantity][row half][column], the space of the row sums) so that the workgroup ends with ONE 64-lane atomic
  // instruction per 64 columns (issued by the lanes that hold the sums -- 4 per instruction -- the same atomics took 270 us).
  __syncthreads();                                          // every wave has read red1 / red2
  float* csum = reinterpret_cast<float*>(smem_raw + C::CSUM);        // [dbeta | dgamma][BN]; the two 16-lane row halves are combined in the lanes
  const int ccol = wave * WN + 4 * h;
  {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        f32x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * gq + e;
          float cs = 0.f;
#pragma unroll
          for (int i = 0; i < MI; ++i) cs += acc[i][j][r];
          t[e] = row16_sum(cs);
          t[e] += __shfl_xor(t[e], 16, 64);
        }
        if (li == 0) *reinterpret_cast<f32x4*>(csum + ccol + j * 32 + 8 * gq) = t;
      }
  }
  float cg[NJ][16];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) cg[j][r] = 0.f;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = u / JH, jh = u % JH;
    LdsPtr strip = stash + u * C::STRIP;
    asm volatile("" ::: "memory");
    X4<T> xh_t[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) xh_t[c] = *reinterpret_cast<const lds_tx4*>(strip + li * 144 + c * 16 + h * 8);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int j = jh * 2 + jj;
        const int n = wave * WN + j * 32 + 8 * gq + 4 * h;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gam + n);
        X4<T> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (float)xh_t[jj * 4 + gq][e], v = acc[i][j][4 * gq + e];
          cg[j][4 * gq + e] += v * xh;
          o[e] = (T)(rs[i] * (ga[e] * v - m1[i] - xh * m2[i]));
        }
        *reinterpret_cast<lds_tx4*>(strip + li * 144 + (jj * 4 + gq) * 16 + h * 8) = o;   // (the lane's own xhat slot)
      }
    __builtin_amdgcn_wave_barrier();
    const long mb = (long)m0 + i * 32;
#pragma unroll
    for (int it = 0; it < 4; ++it) {                        // whole 128-byte lines: 8 rows x 128 B per store instruction
      const int rr = it * 8 + (lane >> 3), c = lane & 7;
      const u32x4 q = lds_read16(strip + rr * 144 + c * 16);
      if (mb + rr < g.M) *reinterpret_cast<u32x4*>(dx_t + (mb + rr) * BN + wave * WN + jh * 64 + c * 8) = q;
    }
    // (else the dgamma additions are sunk to the end of the kernel and every block's 32 products stay in registers: spills)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(cg[j][r]));
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      f32x4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) { t[e] = row16_sum(cg[j][4 * gq + e]); t[e] += __shfl_xor(t[e], 16, 64); }
      if (li == 0) *reinterpret_cast<f32x4*>(csum + BN + ccol + j * 32 + 8 * gq) = t;
    }
  __syncthreads();
  const float osc = loss_scale_down(g.scale_amax);
  for (int n = threadIdx.x; n < BN; n += NWN * 64) {
    unsafeAtomicAdd(g.dbeta + n, csum[n] * osc);
    unsafeAtomicAdd(g.dgamma + n, csum[BN + n] * osc);
  }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// 0: automatic, 1: only the generic 128x128 kernel, 2: the 256x256 LDS-DMA kernel whenever legal,
// 3: the 128x256 LDS-DMA kernel whenever legal (tests / profiling)
static int g_big_mode = 0;
void set_gemm_nt_big_mode(int mode) { g_big_mode = mode; }
// returns 0 (generic kernel), 1 (256x256) or 2 (128x256)
static int gemm_nt_pick(const GemmNT& g) {
  if (g_big_mode == 1 || !g.vec_ok || g.K % 64 || g.N % 4) return 0;
  if (g_big_mode == 2) return 1;
  if (g_big_mode == 3) return 2;
  const int ntail = g.N % BIG_BN;
  if (ntail && ntail < BIG_BN - BIG_BN / 8) return 0;   // no mostly-empty tile columns (1000 bars: the last of four is 232 wide)
  // measured at the north-star shape with the real epilogues (tools/bench_gemm_epi.py): the 256x256 tile wins
  // whenever it can occupy the chip (3.07 vs 3.53 ms per step); the 128x256 tile covers smaller token counts
  const long tn256 = (g.N + BIG_BN - 1) / BIG_BN;
  const long tiles256 = (long)((g.M + 255) / 256) * tn256, tiles128 = (long)((g.M + 127) / 128) * tn256;
  return tiles256 >= 192 ? 1 : (tiles128 >= 192 ? 2 : 0);
}

template <typename T, int FLAGS, int NWM, int BK> static void launch_big_t(const GemmNT& g, hipStream_t stream) {
  using C = BigCfg<NWM, BK>;
  static LdsAllowance allowance;
  constexpr size_t lds = 2 * C::STAGE + ((FLAGS & EPI_ROWSHIFT) ? 2 * C::BN * 4 : 0);      // (+ the shift rows of two datasets, gemm_nt_big_kernel)
  allowance.ensure(gemm_nt_big_kernel<T, FLAGS, NWM, BK>, lds);
  const int tiles = ((g.M + C::BM - 1) / C::BM) * ((g.N + C::BN - 1) / C::BN);
  hipLaunchKernelGGL((gemm_nt_big_kernel<T, FLAGS, NWM, BK>), dim3(tiles), dim3(C::NT), lds, stream, g);
}
// the epilogue flag combinations the encoder stack uses; anything else takes the generic kernel
template <typename T> static bool launch_big(const GemmNT& g, bool small_tile, hipStream_t stream) {
  switch (g.flags) {
#define PFN_BIG_CASE(F) case (F): if (small_tile) launch_big_t<T, (F), 1, 32>(g, stream); else launch_big_t<T, (F), 2, 64>(g, stream); return true;
    PFN_BIG_CASE(EPI_BIAS | EPI_OUT_T)                            // q/k/v projection
    PFN_BIG_CASE(EPI_BIAS | EPI_OUT_T | EPI_ROWSHIFT)             // ... with the keys centred per dataset (EPI_ROWSHIFT)
    PFN_BIG_CASE(EPI_BIAS | EPI_RESID | EPI_OUT_F32)              // out_proj, linear2 (+ residual)
    PFN_BIG_CASE(EPI_BIAS | EPI_GELU | EPI_OUT_T | EPI_OUT2_T)    // linear1 + GELU
    PFN_BIG_CASE(EPI_GELU_BWD | EPI_OUT_T)                        // d(hpre)
    PFN_BIG_CASE(EPI_RESID | EPI_OUT_F32)
    PFN_BIG_CASE(EPI_RESID_T | EPI_OUT_F32)                       // dx = dgrad + residual gradient (kept in operand precision)
    PFN_BIG_CASE(EPI_RESID_T | EPI_OUT_T)                         // ... between layers the sum stays in operand precision too
    PFN_BIG_CASE(EPI_BIAS | EPI_RESID_T | EPI_OUT_T)              // out_proj / linear2 + residual where the LayerNorm is its own kernel, fp16 sums (emsize 1024)
    PFN_BIG_CASE(EPI_OUT_T)                                       // d(ctx)
    PFN_BIG_CASE(EPI_OUT_T | EPI_ROWDOT)                          // d(ctx) + the attention backward's delta
    PFN_BIG_CASE(EPI_OUT_F32)
    PFN_BIG_CASE(EPI_BIAS | EPI_OUT_F32)
#undef PFN_BIG_CASE
    default: return false;
  }
}

static void nt_prepare(GemmNT& g, int precision);
bool gemm_nt_rowdot_fused(const GemmNT& g_in, int precision) {
  GemmNT g = g_in;
  if (!prec_is16(precision) || g.M <= 0 || g.flags != (EPI_OUT_T | EPI_ROWDOT) || g.N % 64 || g.rd_D % 64 || g.N % g.rd_D) return false;
  if ((g.lda * 2) % 16 || (g.ldb * 2) % 16 || !aligned16(g.A) || !aligned16(g.B)) return false;
  nt_prepare(g, precision);
  return gemm_nt_pick(g) == 1 && g.wide_t;      // the 256 x 256 tile only (the 4-wave 128 x 256 form has no registers left for the row sums: 56 bytes of scratch)
}
int launch_gemm_nt(const GemmNT& g_in, int precision, hipStream_t stream) {
  GemmNT g = g_in;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return PFN_OK;
  const size_t es = prec_esize(precision);
  if ((g.lda * es) % 16 || (g.ldb * es) % 16 || !aligned16(g.A) || !aligned16(g.B)) return PFN_ERR_ALIGNMENT;
  if ((g.flags & EPI_ROWSHIFT) && (!g.rowshift || g.rs_S <= 0 || g.rs_n0 % 64 || g.rs_n1 % 64 || g.rs_n0 < 0 || g.rs_n1 > g.N)) return PFN_ERR_ARGUMENT;
  nt_prepare(g, precision);
  if (prec_is16(precision)) {
    const int pick = gemm_nt_pick(g);
    if (pick && (precision == PFN_PREC_FP16 ? launch_big<f16>(g, pick == 2, stream) : launch_big<bf16>(g, pick == 2, stream)))
      return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
  }
  if (g.flags & EPI_ROWDOT) return PFN_ERR_UNSUPPORTED;      // (callers ask gemm_nt_rowdot_fused first)
  const int tiles = ((g.M + GEMM_BM - 1) / GEMM_BM) * ((g.N + GEMM_BN - 1) / GEMM_BN);
  PFN_DISPATCH_OP(precision, hipLaunchKernelGGL(gemm_nt_kernel<T>, dim3(tiles), dim3(256), 65536, stream, g));
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}
static void nt_prepare(GemmNT& g, int precision) {
  const size_t es = prec_esize(precision);
  // the epilogue moves 4 columns per lane when every stream it touches allows it
  bool vec = true;
  if ((g.flags & EPI_OUT_F32) && (g.ld_out_f32 % 4 || !aligned16(g.out_f32))) vec = false;
  if ((g.flags & EPI_OUT_T) && ((g.ld_out_t * es) % (4 * es) || !aligned16(g.out_t))) vec = false;
  if ((g.flags & EPI_OUT2_T) && ((g.ld_out2 * es) % (4 * es) || !aligned16(g.out2_t))) vec = false;
  if ((g.flags & EPI_RESID) && (g.ld_resid % 4 || !aligned16(g.resid))) vec = false;
  if ((g.flags & (EPI_GELU_BWD | EPI_RESID_T | EPI_ROWDOT)) && ((g.ld_aux * es) % (4 * es) || !aligned16(g.aux))) vec = false;
  if ((g.flags & EPI_BIAS) && !aligned16(g.bias)) vec = false;
  if ((g.flags & EPI_ROWSHIFT) && (!aligned16(g.rowshift) || g.rs_ld % 4 || g.rs_n0 % 4 || g.rs_n1 % 4)) vec = false;
  g.vec_ok = vec ? 1 : 0;
  g.wide_t = (!(g.flags & EPI_OUT_T) || g.ld_out_t % 8 == 0) && (!(g.flags & EPI_OUT2_T) || g.ld_out2 % 8 == 0) ? 1 : 0;   // 16-byte rows
}

bool gemm_ln_supported(const GemmLN& g) {
  return (g.N == 128 || g.N == 256 || g.N == 512 || g.N == 1024) && g.K % 32 == 0 && g.K >= 32 && (g.lda * 2) % 16 == 0 && (g.ldb * 2) % 16 == 0 &&
         aligned16(g.A) && aligned16(g.B) && aligned16(g.bias) && aligned16(g.gamma) && aligned16(g.beta) && aligned16(g.y) && aligned16(g.x_t) &&
         (g.resid ? aligned16(g.resid) : (aligned16(g.ry) && aligned16(g.rgamma) && aligned16(g.rbeta)));
}
template <typename T, bool Y, int NWN, bool RL, int BK> static void launch_gemm_ln_t(const GemmLN& g, hipStream_t stream) {
  const size_t lds = std::max<size_t>(2 * (128 + NWN * 64) * (BK * 2), LnEpi<NWN>::BYTES);
  static LdsAllowance allowance;
  allowance.ensure(gemm_nt_ln_kernel<T, NWN, RL, BK, Y>, lds);
  hipLaunchKernelGGL((gemm_nt_ln_kernel<T, NWN, RL, BK, Y>), dim3((g.M + 127) / 128), dim3(NWN * 64), lds, stream, g);
}
// PFN_TUNE_GEMM_LN_ROWS (test / profiling knob): 1 = the LayerNorm-fused GEMMs at N = 512 run on 64-ROW tiles (4 waves x (2 x 4) blocks, 32-deep stages:
// 72 KiB of LDS, 256 registers per wave) so that TWO workgroups share a CU and one's HBM-bound epilogue can run under the other's MFMA loop; 0 = the
// 128-row tiles that fill the CU's LDS alone
static int g_ln_rows64 = 0;
void set_gemm_ln_rows64(int on) { g_ln_rows64 = on; }
template <typename T, bool Y, int NWN, bool RL, int BK, int MI, int NJ> static void launch_gemm_ln_wide_t(const GemmLN& g, hipStream_t stream) {
  constexpr int BM = MI * 32;
  const size_t lds = std::max<size_t>(2 * (BM + NWN * NJ * 32) * (BK * 2), LnEpiW<NWN, MI, NJ>::BYTES);
  static LdsAllowance allowance;
  allowance.ensure(gemm_nt_ln_wide_kernel<T, NWN, RL, BK, MI, NJ, Y>, lds);
  hipLaunchKernelGGL((gemm_nt_ln_wide_kernel<T, NWN, RL, BK, MI, NJ, Y>), dim3((g.M + BM - 1) / BM), dim3(NWN * 64), lds, stream, g);
}
template <typename T, bool Y> static int launch_gemm_ln_op(const GemmLN& g, hipStream_t stream) {
  const bool rl = g.resid == nullptr;
  // 64-deep stages (two of them fill the CU's 160 KiB of LDS at N = 512) whenever K allows, else 32-deep
#define PFN_LN_CASE(NWN) \
  if (g.K % 64 == 0) { if (rl) launch_gemm_ln_t<T, Y, NWN, true, 64>(g, stream); else launch_gemm_ln_t<T, Y, NWN, false, 64>(g, stream); } \
  else { if (rl) launch_gemm_ln_t<T, Y, NWN, true, 32>(g, stream); else launch_gemm_ln_t<T, Y, NWN, false, 32>(g, stream); }
  if (g.N == 512 && g_ln_rows64) {
    if (rl) launch_gemm_ln_wide_t<T, Y, 4, true, 32, 2, 4>(g, stream); else launch_gemm_ln_wide_t<T, Y, 4, false, 32, 2, 4>(g, stream);
    return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
  }
  switch (g.N / 64) {
    case 2: PFN_LN_CASE(2) break;
    case 4: PFN_LN_CASE(4) break;
    case 8: PFN_LN_CASE(8) break;
    default:   // N = 1024: 64 rows x (8 waves x 128 columns), 32-deep stages (two of them are 136 KiB)
      if (rl) launch_gemm_ln_wide_t<T, Y, 8, true, 32, 2, 4>(g, stream); else launch_gemm_ln_wide_t<T, Y, 8, false, 32, 2, 4>(g, stream);
      break;
  }
#undef PFN_LN_CASE
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}
int launch_gemm_ln(const GemmLN& g, int precision, hipStream_t stream) {
  if (g.M <= 0) return PFN_OK;
  if (!prec_is16(precision) || !gemm_ln_supported(g)) return PFN_ERR_UNSUPPORTED;
  if (g.y16) return precision == PFN_PREC_FP16 ? launch_gemm_ln_op<f16, true>(g, stream) : PFN_ERR_UNSUPPORTED;      // (bf16 keeps its f32 sums: 8 bits on the residual path cost parity)
  return precision == PFN_PREC_FP16 ? launch_gemm_ln_op<f16, false>(g, stream) : launch_gemm_ln_op<bf16, false>(g, stream);
}

bool gemm_lnbwd_supported(const GemmLNB& g) {
  return (g.N == 128 || g.N == 256 || g.N == 512 || g.N == 1024) && g.K % 32 == 0 && g.K >= 32 && (g.lda * 2) % 16 == 0 && (g.ldb * 2) % 16 == 0 &&
         aligned16(g.A) && aligned16(g.B) && aligned16(g.aux) && aligned16(g.y) && aligned16(g.gamma) && aligned16(g.dx_t) && g.dgamma && g.dbeta;
}
template <typename T, bool Y, int NWN, int BK> static void launch_gemm_lnbwd_t(const GemmLNB& g, hipStream_t stream) {
  const size_t lds = LnbCfg<NWN, BK>::LDS;
  static LdsAllowance allowance;
  allowance.ensure(gemm_nt_lnbwd_kernel<T, NWN, BK, Y>, lds);
  hipLaunchKernelGGL((gemm_nt_lnbwd_kernel<T, NWN, BK, Y>), dim3((g.M + 127) / 128), dim3(NWN * 64), lds, stream, g);
}
template <typename T, bool Y, int NWN, int BK, int MI, int NJ> static void launch_gemm_lnbwd_wide_t(const GemmLNB& g, hipStream_t stream) {
  const size_t lds = LnbCfgW<NWN, BK, MI, NJ>::LDS;
  static LdsAllowance allowance;
  allowance.ensure(gemm_nt_lnbwd_wide_kernel<T, NWN, BK, MI, NJ, Y>, lds);
  hipLaunchKernelGGL((gemm_nt_lnbwd_wide_kernel<T, NWN, BK, MI, NJ, Y>), dim3((g.M + MI * 32 - 1) / (MI * 32)), dim3(NWN * 64), lds, stream, g);
}
template <typename T, bool Y> static int launch_gemm_lnbwd_op(const GemmLNB& g, hipStream_t stream) {
#define PFN_LNB_CASE(NWN) if (g.K % 64 == 0) launch_gemm_lnbwd_t<T, Y, NWN, 64>(g, stream); else launch_gemm_lnbwd_t<T, Y, NWN, 32>(g, stream);
  if (g.N == 512 && g_ln_rows64) {
    launch_gemm_lnbwd_wide_t<T, Y, 4, 32, 2, 4>(g, stream);
    return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
  }
  switch (g.N / 64) {
    case 2: PFN_LNB_CASE(2) break;
    case 4: PFN_LNB_CASE(4) break;
    case 8: PFN_LNB_CASE(8) break;
    default: launch_gemm_lnbwd_wide_t<T, Y, 8, 32, 2, 4>(g, stream); break;     // N = 1024: 64-row tiles, 32-deep stages
  }
#undef PFN_LNB_CASE
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}
int launch_gemm_lnbwd(const GemmLNB& g, int precision, hipStream_t stream) {
  if (g.M <= 0) return PFN_OK;
  if (!prec_is16(precision) || !gemm_lnbwd_supported(g)) return PFN_ERR_UNSUPPORTED;
  if (g.y16) return precision == PFN_PREC_FP16 ? launch_gemm_lnbwd_op<f16, true>(g, stream) : PFN_ERR_UNSUPPORTED;
  return precision == PFN_PREC_FP16 ? launch_gemm_lnbwd_op<f16, false>(g, stream) : launch_gemm_lnbwd_op<bf16, false>(g, stream);
}

}  // namespace pfn
