// Weight-gradient GEMMs of the PFN encoder stack (gfx950): C[P,Q] (+)= A[M,P]^T . B[M,Q], contraction over the (long) token axis.
//   gemm_tn_kernel      : one product, 128 x 128 tiles, token splits with f32 atomics (templated over every operand type; the decoder / embedding shapes, f32 mode)
//   gemm_tn_big_kernel  : the GROUPED launch of all 4 L weight (and fused bias) gradients on 256 x 256 tiles (16-bit operands)
// Replaces autograd's dW = dY^T X of every nn.Linear of the stack (reference train.py:93; torch nn/modules/transformer.py:952-982).
#include <type_traits>
#include "pfn_device.h"
#include "pfn_kernels.h"

namespace pfn {

// (as in gemm.hip) XCD-aware, bijective remap of the linear workgroup id
PFN_DEV int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

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
        const X8<T> v = __builtin_bit_cast(X8<T>, sa.regs[i]);
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

  const float osc = loss_scale_down(g.scale_amax);      // fp16 backward: the gradient leaves unscaled (pfn_device.h)
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
        if (col < g.P) unsafeAtomicAdd(g.colsum + col, t * loss_scale_down(g.scale_amax));
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
          if (g.atomic) unsafeAtomicAdd(c, acc[i][j][r] * osc);
          else *c = acc[i][j][r] * osc;
        }
      }
}

// ---------------------------------------------------------------------------------------------
// gemm_tn_big_kernel: grouped weight-gradient GEMMs  C_g[P,Q] (+)= A_g[M,P]^T . B_g[M,Q]  (bf16).
//   One launch covers every weight gradient of the encoder stack (pfn_stack_backward keeps the
//   per-layer output-gradient operands resident and defers the weight gradients to one grouped
//   launch), so there are enough 256 x 256 output tiles to fill the chip WITHOUT splitting the
//   long token axis: no atomics, a deterministic summation order, plain read-modify-write of C.
//   (An optional split over tokens with f32 atomics stays available for small groups.)
//   * 256 x 256 tile, 64-token stages, 8 waves as 2 (P) x 4 (Q), each 128 x 64;
//   * both operands are token-major, so their LDS images are [token][256 columns] (512-byte rows)
//     filled by LDS-DMA and read through ds_read_b64_tr_b16; the 64-byte-unit XOR swizzle of
//     load_frag_tr is applied to the per-lane DMA source address;
//   * bias gradients (column sums of A) ride along as one extra MFMA against a constant ones
//     fragment in the workgroups of the first Q tile, split over the four Q waves.
// Requirements: P % 256 == 0, Q % 256 == 0, 16-byte aligned rows.
// ---------------------------------------------------------------------------------------------
// Stage depth / ring length (experiment builds: tools/build_variants.sh with -DPFN_TNB_KT= -DPFN_TNB_NS=).  Measured at the north
// star (tools/bench_wgrad.py): 64 x 2 950 us, 32 x 3 952, 32 x 4 965, 32 x 5 965 -- up to four stages in flight instead of one change
// nothing, i.e. the operand stream is bound by its rate (6.9 TB/s of LDS-DMA traffic, 2.6 TB/s of it from HBM), not by latency.
#ifndef PFN_TNB_KT
#define PFN_TNB_KT 64
#define PFN_TNB_NS 2
#endif
constexpr int TNB_KT = PFN_TNB_KT;               // tokens per stage
constexpr int TNB_NS = PFN_TNB_NS;               // stages in the LDS ring: TNB_NS - 1 of them in flight under the one being multiplied
constexpr int TNB_TILE = TNB_KT * 512;           // bytes per operand per stage
constexpr int TNB_PW = TNB_KT / 16;              // 1-KiB DMA pieces per wave per operand per stage (a piece = 2 token rows x 512 B)
constexpr int TNB_LDS = TNB_NS * 2 * TNB_TILE;
static_assert(TNB_LDS <= 160 * 1024 && TNB_KT % 16 == 0 && TNB_NS >= 2 && (TNB_NS - 2) * 2 * TNB_PW < 64, "weight-gradient ring does not fit");

template <typename T>
__global__ __launch_bounds__(512, 1) void gemm_tn_big_kernel(GemmTNGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr smem = lds_cast(smem_raw);

  const int ntiles = g.tile_start[g.n];
  const int id = xcd_remap(blockIdx.x, ntiles * g.splits);
  const int split = id / ntiles, tile = id % ntiles;
  int pi = 0;
  while (tile >= g.tile_start[pi + 1]) ++pi;
  const TnProblem& pr = g.p[pi];
  const int tq = pr.Q / 256;
  const int tl = tile - g.tile_start[pi];
  const int p0 = (tl / tq) * 256, q0 = (tl % tq) * 256;
  const long mbeg = (long)split * g.m_chunk;
  const long mend = min((long)g.M, mbeg + g.m_chunk);
  const int rows_total = (int)(mend - mbeg);
  if (rows_total <= 0) return;

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wp = wave >> 2, wq = wave & 3;

  // DMA sources: a 1-KiB piece is 2 token rows x 512 B; wave w moves pieces w, w + 8, ... of each operand's stage
  const T* pa[TNB_PW];
  const T* pb[TNB_PW];
  int prow[TNB_PW];
#pragma unroll
  for (int i = 0; i < TNB_PW; ++i) {
    const int row = (wave + 8 * i) * 2 + (lane >> 5);
    const int unit = ((lane & 31) >> 2) ^ (row & 3);
    const int col = unit * 32 + (lane & 3) * 8;
    prow[i] = row;
    pa[i] = reinterpret_cast<const T*>(pr.A) + mbeg * pr.lda + p0 + col;
    pb[i] = reinterpret_cast<const T*>(pr.B) + mbeg * pr.ldb + q0 + col;
  }
  auto stage = [&](int slot, int r0) {
    LdsPtr ta = smem + slot * 2 * TNB_TILE + wave * 1024;
    LdsPtr tb = ta + TNB_TILE;
#pragma unroll
    for (int i = 0; i < TNB_PW; ++i) {
      // tail rows are re-zeroed in LDS below; debug_mask is all ones except when profiling with cache-resident operands
      const long r = min(r0 + prow[i], rows_total - 1) & g.debug_mask;
      // assembly form (pfn_device.h dma16): with the builtin hipcc waits vmcnt(0) in front of the first ds_read_b64_tr_b16 of
      // the stage being MULTIPLIED -- the intrinsic carries no address, so the DMA just issued for a LATER stage "may alias" --
      // and the copy never overlapped the MFMAs
      dma16_global(pa[i] + r * pr.lda, ta + i * 8192);
      dma16_global(pb[i] + r * pr.ldb, tb + i * 8192);
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x16 cs;
#pragma unroll
  for (int r = 0; r < 16; ++r) cs[r] = 0.f;
  const bool do_colsum = pr.colsum != nullptr && q0 == 0;
  Frag<T> ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones.v[e] = (T)1.0f;

  const int nt = (rows_total + TNB_KT - 1) / TNB_KT;
  // Ring of TNB_NS stages: stage t is multiplied while stages t+1 .. t+TNB_NS-2 are in flight and stage t+TNB_NS-1 is requested
  // into the slot stage t-1 has just released.  Loads retire in order, so "stage t has landed" is vmcnt(instructions of the
  // younger stages) -- explicit s_waitcnt + raw s_barrier (wait_vm_barrier): __syncthreads() carries vmcnt(0).
  // (Touching later stages' lines to pull them into L2 instead made the launch slower: 916 us without, 990 / 1030 / 1070 us
  // touching 2 / 3 / 5 stages ahead.)
#pragma unroll
  for (int st = 0; st < TNB_NS - 1; ++st)
    if (st < nt) stage(st, st * TNB_KT);
  auto wait_stage = [&](int younger) {     // `younger` stages were requested after the one wanted now (wave-uniform)
    switch (younger) {
      case 0: wait_vm_barrier<0>(); break;
      case 1: wait_vm_barrier<(TNB_NS > 2 ? 1 : 0) * 2 * TNB_PW>(); break;
      case 2: wait_vm_barrier<(TNB_NS > 3 ? 2 : 0) * 2 * TNB_PW>(); break;
      case 3: wait_vm_barrier<(TNB_NS > 4 ? 3 : 0) * 2 * TNB_PW>(); break;
      default: wait_vm_barrier<(TNB_NS - 2) * 2 * TNB_PW>(); break;
    }
  };
  static_assert(TNB_NS <= 6, "wait_stage cases");
  // the main loop exists twice (with / without the bias-gradient MFMA) so neither carries a branch
  auto main_loop = [&](auto with_colsum) {
    constexpr bool CS = decltype(with_colsum)::value;
    int slot = 0, slot_next = TNB_NS - 1;
    for (int t = 0; t < nt; ++t) {
      wait_stage(min(nt - t - 1, TNB_NS - 2));
      if (t + TNB_NS - 1 < nt) stage(slot_next, (t + TNB_NS - 1) * TNB_KT);
      LdsPtr ta = smem + slot * 2 * TNB_TILE;
      LdsPtr tb = ta + TNB_TILE;
      const int valid = rows_total - t * TNB_KT;
      if (valid < TNB_KT) {
        // ragged last stage: the DMA clamped its source rows; clear the rows past the end (A only:
        // a zero A row contributes nothing whatever B holds there, and B's clamped rows are finite data)
        for (int idx = threadIdx.x; idx < (TNB_KT - valid) * 32; idx += 512) {
          const u32x4 z = {0u, 0u, 0u, 0u};
          lds_write16(ta + (valid + idx / 32) * 512 + (idx % 32) * 16, z);
        }
        __syncthreads();
      }
#pragma unroll
      for (int ks = 0; ks < TNB_KT; ks += 16) {
        Frag<T> fa[4], fb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = load_frag_tr<T, 512, 1>(tb, ks, wq * 64 + j * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = load_frag_tr<T, 512, 1>(ta, ks, wp * 128 + i * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mma32(fa[i], fb[j], acc[i][j]);
        // bias gradient: Q wave wq sums the columns of P sub-tile wq (one more fragment read, one more MFMA)
        if constexpr (CS) cs = mma32(load_frag_tr<T, 512, 1>(ta, ks, wp * 128 + wq * 32), ones, cs);
      }
      slot_next = slot;
      slot = slot + 1 == TNB_NS ? 0 : slot + 1;
    }
  };
  if (do_colsum) main_loop(std::true_type{});
  else main_loop(std::false_type{});

  const int pv = pr.Pv > 0 ? pr.Pv : pr.P;    // rows of C that exist (A may end in zero-padding columns)
  const float osc = loss_scale_down(g.scale_amax);      // fp16 backward: the gradients leave unscaled (pfn_device.h)
  if (do_colsum && (lane & 31) == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pp = p0 + wp * 128 + wq * 32 + acc_row(r, lane);
      if (pp < pv) unsafeAtomicAdd(pr.colsum + pp, cs[r] * osc);
    }
  }
  // always atomic: token splits add into the same tile, and concurrent backward passes (micro-batches on several
  // HIP streams, streams.py) accumulate into the same gradient buffer
  const bool atomic = true;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pp = p0 + wp * 128 + i * 32 + acc_row(r, lane);
        const int qq = q0 + wq * 64 + j * 32 + (lane & 31);
        float* c = pr.C + (long)pp * pr.ldc + qq;
        if (pp >= pv) continue;
        if (atomic) unsafeAtomicAdd(c, acc[i][j][r] * osc);
        else *c += acc[i][j][r] * osc;
      }
}

// ---------------------------------------------------------------------------------------------
// gemm_tn_wide_kernel (round 6): the same grouped product on the same 256 x 256 tile with FOUR waves of 128 x 128 instead of eight of 128 x 64.
// Why: both operands are token-major, so every fragment is a transposing LDS read of 8 bytes per lane (ds_read_b64_tr_b16) -- 12 of them per 8 MFMAs in the 8-wave
// shape.  Counted per CU and 16-token step that is 96 LDS read instructions + the DMA's own 16 KiB of LDS writes against 512 cycles of MFMA per SIMD: the LDS port,
// not the matrix pipe, is what the 8-wave kernel runs at (838 TF/s = a third of the MFMA rate, whatever the depth of the operand ring: gemm_tn_big_kernel's header).
// A wave that owns 128 x 128 reads (128 + 128) columns for twice the MFMAs: 64 read instructions per CU and step, a third less LDS traffic per flop.
// Price: the 16 accumulator tiles are 256 registers, so a wave needs the whole register file (one wave per SIMD: no second wave to cover its LDS round trips) --
// the fragments of step s + 1 are therefore requested before the MFMAs of step s (two fragment sets, PFN_PIN_LDS_MFMA keeps the machine scheduler from sinking the
// reads back to their consumers), the first step of the next stage right behind the stage barrier.
// ---------------------------------------------------------------------------------------------
constexpr int TNW_NW = 4;                              // waves: 2 (P) x 2 (Q), each 128 x 128
constexpr int TNW_PW = TNB_KT / 2 / TNW_NW;            // 1-KiB DMA pieces per wave per operand per stage
static_assert(TNB_KT == 64 && TNB_NS == 2, "gemm_tn_wide_kernel is written for two 64-token stages");
template <typename T>
__global__ __launch_bounds__(TNW_NW * 64, 1) void gemm_tn_wide_kernel(GemmTNGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr smem = lds_cast(smem_raw);
  const int ntiles = g.tile_start[g.n];
  const int id = xcd_remap(blockIdx.x, ntiles * g.splits);
  const int split = id / ntiles, tile = id % ntiles;
  int pi = 0;
  while (tile >= g.tile_start[pi + 1]) ++pi;
  const TnProblem& pr = g.p[pi];
  const int tq = pr.Q / 256;
  const int tl = tile - g.tile_start[pi];
  const int p0 = (tl / tq) * 256, q0 = (tl % tq) * 256;
  const long mbeg = (long)split * g.m_chunk;
  const long mend = min((long)g.M, mbeg + g.m_chunk);
  const int rows_total = (int)(mend - mbeg);
  if (rows_total <= 0) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wp = wave >> 1, wq = wave & 1;

  // DMA sources: piece i of this wave = token rows (wave + 4 i) * 2 + (lane >> 5); the swizzled column depends on row & 3 only, i.e. not on i (8 i = 0 mod 4):
  // ONE pointer per operand, the row offset added per piece (sixteen 64-bit pointers would not fit beside 256 accumulator registers)
  const int prow0 = wave * 2 + (lane >> 5);
  const int pcol = ((((lane & 31) >> 2) ^ (prow0 & 3)) * 32) + (lane & 3) * 8;
  const T* pa = reinterpret_cast<const T*>(pr.A) + mbeg * pr.lda + p0 + pcol;
  const T* pb = reinterpret_cast<const T*>(pr.B) + mbeg * pr.ldb + q0 + pcol;
  const long lda = pr.lda, ldb = pr.ldb;
  auto stage = [&](int slot, int r0) {
    LdsPtr ta = smem + slot * 2 * TNB_TILE + wave * 1024;
    LdsPtr tb = ta + TNB_TILE;
#pragma unroll
    for (int i = 0; i < TNW_PW; ++i) {
      const long r = min(r0 + prow0 + 2 * TNW_NW * i, rows_total - 1) & g.debug_mask;
      dma16_global(pa + r * lda, ta + i * TNW_NW * 1024);
      dma16_global(pb + r * ldb, tb + i * TNW_NW * 1024);
    }
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // bias gradient (column sums of A over the tokens): on the vector ALU, straight from the A fragments the products read anyway -- lane l holds 8 tokens of column
  // l & 31 of each of its 4 sub-tiles; Q wave wq sums sub-tiles 2 wq and 2 wq + 1 (64 adds per step under 512 cycles of MFMA; two more accumulator tiles do not fit)
  float cs[2] = {0.f, 0.f};
  const bool do_colsum = pr.colsum != nullptr && q0 == 0;

  const int nt = (rows_total + TNB_KT - 1) / TNB_KT;
  struct Frs { Frag<T> a[4], b[4]; };
  auto main_loop = [&](auto with_colsum) {
    constexpr bool CS = decltype(with_colsum)::value;
    Frs f0, f1;
    auto ld = [&](Frs& f, const lds_char* ta, const lds_char* tb, int ks) {
#pragma unroll
      for (int j = 0; j < 4; ++j) f.b[j] = load_frag_tr<T, 512, 1>(tb, ks, wq * 128 + j * 32);
#pragma unroll
      for (int i = 0; i < 4; ++i) f.a[i] = load_frag_tr<T, 512, 1>(ta, ks, wp * 128 + i * 32);
    };
    auto mm = [&](const Frs& f) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mma32(f.a[i], f.b[j], acc[i][j]);
      // bias gradient: Q wave wq sums the columns of the wave's P sub-tiles 2 wq and 2 wq + 1 (their fragments are in registers already)
      if constexpr (CS) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int e = 0; e < 8; ++e) cs[c] += (float)(wq ? f.a[2 + c] : f.a[c]).v[e];
      }
    };
    // stage t has landed for every wave and every read of the slot it is about to overwrite has retired (wait_vm_barrier waits lgkmcnt(0) too);
    // request stage t + 1 into the other slot; clear the rows past the end of a ragged last stage
    auto enter_stage = [&](int t, int slot) {
      wait_vm_barrier<0>();
      if (t + 1 < nt) stage(slot ^ 1, (t + 1) * TNB_KT);
      const int valid = rows_total - t * TNB_KT;
      if (valid < TNB_KT) {
        LdsPtr ta = smem + slot * 2 * TNB_TILE;
        for (int idx = threadIdx.x; idx < (TNB_KT - valid) * 32; idx += TNW_NW * 64) {
          const u32x4 z = {0u, 0u, 0u, 0u};
          lds_write16(ta + (valid + idx / 32) * 512 + (idx % 32) * 16, z);
        }
        __syncthreads();
      }
    };
    int slot = 0;
    enter_stage(0, slot);
    ld(f0, smem, smem + TNB_TILE, 0);
    for (int t = 0; t < nt; ++t) {
      const lds_char* ta = smem + slot * 2 * TNB_TILE;
      const lds_char* tb = ta + TNB_TILE;
      ld(f1, ta, tb, 16);
      PFN_PIN_LDS_MFMA();
      mm(f0);
      PFN_PIN_LDS_MFMA();
      ld(f0, ta, tb, 32);
      PFN_PIN_LDS_MFMA();
      mm(f1);
      PFN_PIN_LDS_MFMA();
      ld(f1, ta, tb, 48);
      PFN_PIN_LDS_MFMA();
      mm(f0);
      PFN_PIN_LDS_MFMA();
      slot ^= 1;
      if (t + 1 < nt) {
        enter_stage(t + 1, slot);
        const lds_char* tan = smem + slot * 2 * TNB_TILE;
        ld(f0, tan, tan + TNB_TILE, 0);
      }
      PFN_PIN_LDS_MFMA();
      mm(f1);
      PFN_PIN_LDS_MFMA();
    }
  };
  stage(0, 0);
  if (do_colsum) main_loop(std::true_type{});
  else main_loop(std::false_type{});

  const int pv = pr.Pv > 0 ? pr.Pv : pr.P;
  const float osc = loss_scale_down(g.scale_amax);
  if (do_colsum) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float s = cs[c] + __shfl_xor(cs[c], 32, 64);      // the two half-waves hold the other 8 tokens of every step
      const int pp = p0 + wp * 128 + (2 * wq + c) * 32 + (lane & 31);
      if (lane < 32 && pp < pv) unsafeAtomicAdd(pr.colsum + pp, s * osc);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pp = p0 + wp * 128 + i * 32 + acc_row(r, lane);
        const int qq = q0 + wq * 128 + j * 32 + (lane & 31);
        if (pp < pv) unsafeAtomicAdd(pr.C + (long)pp * pr.ldc + qq, acc[i][j][r] * osc);
      }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int g_tn_debug_wrap = 0;
void set_gemm_tn_debug_wrap(int rows) { g_tn_debug_wrap = rows; }
static int g_tn_group_splits = 0;
void set_gemm_tn_group_splits(int splits) { g_tn_group_splits = splits; }
static bool g_tn_wide = false;
void set_gemm_tn_group_waves(int waves) { g_tn_wide = waves == 4; }

int launch_gemm_tn(GemmTN g, int precision, hipStream_t stream) {
  if (g.M <= 0 || g.P <= 0 || g.Q <= 0) return PFN_OK;
  const size_t es = prec_esize(precision);
  if ((g.lda * es) % 16 || (g.ldb * es) % 16 || !aligned16(g.A) || !aligned16(g.B)) return PFN_ERR_ALIGNMENT;
  const int tp = (g.P + 127) / 128, tq = (g.Q + 127) / 128;
  int splits = (1024 + tp * tq - 1) / (tp * tq);
  const int max_splits = (g.M + 4 * TN_BMK - 1) / (4 * TN_BMK);
  if (splits > max_splits) splits = max_splits;
  if (g.max_splits > 0 && splits > g.max_splits) splits = g.max_splits;
  if (splits < 1) splits = 1;
  if (!g.atomic) splits = 1;
  int chunk = (g.M + splits - 1) / splits;
  chunk = (chunk + TN_BMK - 1) / TN_BMK * TN_BMK;
  splits = (g.M + chunk - 1) / chunk;
  g.m_chunk = chunk;
  const size_t lds = 4 * TN_BMK * 128 * es;
  PFN_DISPATCH_OP(precision, hipLaunchKernelGGL(gemm_tn_kernel<T>, dim3(tq, tp, splits), dim3(256), lds, stream, g));
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}

bool gemm_tn_group_supported(const TnProblem& p) {
  return p.P % 256 == 0 && p.Q % 256 == 0 && (p.lda * 2) % 16 == 0 && (p.ldb * 2) % 16 == 0 && aligned16(p.A) && aligned16(p.B);
}

int launch_gemm_tn_group(GemmTNGroup g, int precision, hipStream_t stream) {
  if (g.n <= 0 || g.M <= 0) return PFN_OK;
  if (g.n > TN_GROUP_MAX) return PFN_ERR_ARGUMENT;
  if (!prec_is16(precision)) return PFN_ERR_UNSUPPORTED;
  int tiles = 0;
  for (int i = 0; i < g.n; ++i) {
    if (!gemm_tn_group_supported(g.p[i])) return PFN_ERR_UNSUPPORTED;
    g.tile_start[i] = tiles;
    tiles += (g.p[i].P / 256) * (g.p[i].Q / 256);
  }
  g.tile_start[g.n] = tiles;
  // split the token axis only when the group cannot occupy the chip by itself
  // automatic: the smallest split count (<= 8) whose workgroup count fills whole rounds of the 256 CUs to >= 90 %
  int splits = g.splits;
  if (splits <= 0 && g_tn_group_splits > 0) splits = g_tn_group_splits;      // (test / profiling knob)
  if (splits <= 0) {
    splits = 8;
    for (int sp = 1; sp <= 8; ++sp) {   // (a 256 x 256 partial tile is 65536 atomics: more than 8 splits cost more than they fill)
      const int blocks = tiles * sp, rounds = (blocks + 255) / 256;
      if (blocks >= 0.9 * rounds * 256) { splits = sp; break; }
    }
  }
  const int max_splits = (g.M + 4 * TNB_KT - 1) / (4 * TNB_KT);
  if (splits > max_splits) splits = max_splits;
  int chunk = (g.M + splits - 1) / splits;
  chunk = (chunk + TNB_KT - 1) / TNB_KT * TNB_KT;
  splits = (g.M + chunk - 1) / chunk;
  g.splits = splits;
  g.m_chunk = chunk;
  g.debug_mask = g_tn_debug_wrap > 0 ? g_tn_debug_wrap - 1 : 0x7fffffff;
  if (g_tn_wide) {      // PFN_TUNE_WGRAD_WAVES = 4: four waves of 128 x 128 (gemm_tn_wide_kernel)
    static LdsAllowance allow_w[2];
    if (precision == PFN_PREC_FP16) {
      allow_w[1].ensure(gemm_tn_wide_kernel<f16>, TNB_LDS);
      hipLaunchKernelGGL(gemm_tn_wide_kernel<f16>, dim3(tiles * splits), dim3(TNW_NW * 64), TNB_LDS, stream, g);
    } else {
      allow_w[0].ensure(gemm_tn_wide_kernel<bf16>, TNB_LDS);
      hipLaunchKernelGGL(gemm_tn_wide_kernel<bf16>, dim3(tiles * splits), dim3(TNW_NW * 64), TNB_LDS, stream, g);
    }
    return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
  }
  static LdsAllowance allowance[2];
  if (precision == PFN_PREC_FP16) {
    allowance[1].ensure(gemm_tn_big_kernel<f16>, TNB_LDS);
    hipLaunchKernelGGL(gemm_tn_big_kernel<f16>, dim3(tiles * splits), dim3(512), TNB_LDS, stream, g);
  } else {
    allowance[0].ensure(gemm_tn_big_kernel<bf16>, TNB_LDS);
    hipLaunchKernelGGL(gemm_tn_big_kernel<bf16>, dim3(tiles * splits), dim3(512), TNB_LDS, stream, g);
  }
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}

}  // namespace pfn
