// EXPERIMENTS, NOT PART OF libpfn_hip.so (hygiene): two main-loop structures of the 256 x 256 NT GEMM that round 5 built, tested and measured
// SLOWER than the shipped kernel (profiles/r05_gemm_experiments.txt) -- kept here, out of the product binary, as the record of what was tried:
//   gemm_nt_ring_kernel    : the operand stream as a ring of four 32-deep stages, three in flight              (-4 ... 9 %)
//   gemm_nt_persist_kernel : one workgroup per CU walking tiles, the next tile's first stage requested early    (within noise, more code)
// The third one, software-pipelined fragment reads (-DPFN_GEMM_FRAG_PIPE=1 bodies inside the big / LayerNorm-fused / grouped-TN kernels), lived in
// gemm.hip up to commit 9097de4 (round 5's last) and is in the git history only.
// This translation unit INCLUDES the product's gemm.hip (tile configuration, epilogue, launch helpers) and adds the two kernels plus one C entry point; it builds
// into its own library and never into the product:
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -shared -o libpfn_gemm_experiments.so experiments/gemm_nt_variants.hip
// bf16 operands only (the experiments predate the fp16 instantiations).
#include <cstring>
#include "../gemm.hip"

namespace pfn {

// ---------------------------------------------------------------------------------------------
// gemm_nt_ring_kernel (round 5 experiment, PFN_TUNE_GEMM_NT_KERNEL = 4): the 256 x 256 tile with its operand stream as a RING of four 32-deep stages
// (4 x 32 KiB = the same 128 KiB as two 64-deep stages), three of them in flight under the one being multiplied.  The big kernel has ONE stage in flight and ends
// every stage with vmcnt(0) + barrier: a stage can then never take less than the round trip of its own 64 KiB (issue -> last piece landed), whatever the matrix
// pipe could do with it.  Here stage t + 3 is requested when stage t is entered and the wait for stage t is vmcnt(pieces of the two younger stages): the stream
// is bound by its RATE, not by a round trip per stage.  All operand traffic is LDS-DMA from assembly, every barrier a raw s_barrier behind its own s_waitcnt
// (wait_vm_barrier), as in gemm_tn_big_kernel.  Price: a barrier and a fragment restart every 32 of K instead of every 64.
// ---------------------------------------------------------------------------------------------
constexpr int RING_NS = 4;
template <int FLAGS>
__global__ __launch_bounds__(512, 1) void gemm_nt_ring_kernel(GemmNT g) {
  using C = BigCfg<2, 32>;
  static_assert(RING_NS * C::STAGE >= BigEpi<2>::BYTES, "epilogue staging must fit the (dead) ring");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr smem = lds_cast(smem_raw);
  const int tiles_n = (g.N + C::BN - 1) / C::BN;
  const int tiles_m = (g.M + C::BM - 1) / C::BM;
  const int tid_lin = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = tid_lin / tiles_n, tn = tid_lin % tiles_n;
  const int m0 = tm * C::BM, n0 = tn * C::BN;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 31;

  const bf16* pa[C::PA];
  const bf16* pb[C::PB];
#pragma unroll
  for (int i = 0; i < C::PA; ++i) {
    const int row = (wave + C::NW * i) * C::RPP + lane / C::CPR;
    pa[i] = reinterpret_cast<const bf16*>(g.A) + (long)min(m0 + row, g.M - 1) * g.lda + swz16<C::RB>(row, lane % C::CPR) * 8;
  }
#pragma unroll
  for (int i = 0; i < C::PB; ++i) {
    const int row = (wave + C::NW * i) * C::RPP + lane / C::CPR;
    pb[i] = reinterpret_cast<const bf16*>(g.B) + (long)min(n0 + row, g.N - 1) * g.ldb + swz16<C::RB>(row, lane % C::CPR) * 8;
  }
  auto stage = [&](int slot, int k0) {
    LdsPtr ta = smem + slot * C::STAGE + wave * 1024;
    LdsPtr tb = smem + slot * C::STAGE + C::TILE_A + wave * 1024;
#pragma unroll
    for (int i = 0; i < C::PA; ++i) dma16_global(pa[i] + k0, ta + i * C::NW * 1024);
#pragma unroll
    for (int i = 0; i < C::PB; ++i) dma16_global(pb[i] + k0, tb + i * C::NW * 1024);
  };
  constexpr int PIECES = C::PA + C::PB;      // DMA instructions of a wave per stage

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / 32;
#pragma unroll
  for (int st = 0; st < RING_NS - 1; ++st)
    if (st < nk) stage(st, st * 32);
  int slot = 0, slot_next = RING_NS - 1;
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt has landed for every wave (loads retire in order: everything but the pieces of the younger stages), and every wave's reads of the slot that is
    // about to be overwritten -- stage kt - 1's -- have retired (lgkmcnt(0) in front of the barrier)
    switch (min(nk - kt - 1, RING_NS - 2)) {
      case 0: wait_vm_barrier<0>(); break;
      case 1: wait_vm_barrier<PIECES>(); break;
      default: wait_vm_barrier<2 * PIECES>(); break;
    }
    if (kt + RING_NS - 1 < nk) stage(slot_next, (kt + RING_NS - 1) * 32);
    const lds_char* ta = smem + slot * C::STAGE;
    const lds_char* tb = ta + C::TILE_A;
#pragma unroll
    for (int ks = 0; ks < 32; ks += 16) {
      Frag<bf16> fa[4], fb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = load_frag_row<bf16, C::RB>(tb, wn * 64 + j * 32 + li, ks);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = load_frag_row<bf16, C::RB>(ta, wm * 128 + i * 32 + li, ks);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mma32(fb[j], fa[i], acc[i][j]);
    }
    slot_next = slot;
    slot = slot + 1 == RING_NS ? 0 : slot + 1;
  }
  wait_vm_barrier<0>();      // every wave is done reading the ring before the epilogue's strips go there
  nt_big_epilogue<bf16, FLAGS, 2>(g, acc, m0, n0, wave, lane, smem);
}

// ---------------------------------------------------------------------------------------------
// gemm_nt_persist_kernel: the 256 x 256 kernel as ONE workgroup per CU walking tiles (t = blockIdx.x, + gridDim.x, ...).
// A tile of a K = 512 GEMM spends 7-8 of its ~20 us in latencies that sit in series: the first stage's DMA, the epilogue's loads, the
// stores' acknowledgements before the workgroup may retire, the next workgroup's launch (tools/bench_gemm_epi.py --k 64).  Here
//   * the first stage of the NEXT tile is requested during the last contraction stage of this one (into the buffer that stage
//     does not read: the stage count is even);
//   * the epilogue's stores are fire-and-forget: the next tile's first barrier waits with vmcnt(stores of this wave) -- loads and
//     stores retire in order, so everything older than the stores, i.e. that first stage, has landed while the stores drain under
//     the next tile's MFMAs.
// All operand traffic is LDS-DMA issued from assembly (pfn_device.h dma16_global) and every barrier is a raw s_barrier with its own
// s_waitcnt: the compiler's __syncthreads() carries vmcnt(0), which would wait for the stores.  The epilogue stages through the
// second stage buffer (+ 8 KiB past it), the first one is receiving the next tile.
// Requirements on top of the big kernel's: K / 64 even, every tile full in N.
// ---------------------------------------------------------------------------------------------
template <int FLAGS> struct PersistCfg {
  using C = BigCfg<2, 64>;
  static constexpr int EP_OFF = C::STAGE;                                  // epilogue strips start at the second stage buffer
  static constexpr int LDS = EP_OFF + BigEpi<2>::BYTES > 2 * C::STAGE ? EP_OFF + BigEpi<2>::BYTES : 2 * C::STAGE;
  // global store instructions of one wave's epilogue on the whole-line path (full tile)
  static constexpr int NST = ((FLAGS & EPI_OUT_T) ? 16 : 0) + ((FLAGS & EPI_OUT2_T) ? 16 : 0) + ((FLAGS & EPI_OUT_F32) ? 32 : 0);
  static_assert(NST > 0 && NST < 64, "vmcnt is a 6-bit counter");
};

template <int FLAGS>
__global__ __launch_bounds__(512, 1) void gemm_nt_persist_kernel(GemmNT g) {
  using C = BigCfg<2, 64>;
  using P = PersistCfg<FLAGS>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr smem = lds_cast(smem_raw);
  const int tiles_n = g.N / C::BN;
  const int tiles_m = (g.M + C::BM - 1) / C::BM;
  const int ntiles = tiles_m * tiles_n;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 31;
  const int nk = g.K / 64;

  // DMA sources of a tile: wave w moves the 1-KiB pieces w, w + 8, ... of each operand tile (swizzle on the source chunk, as in
  // the big kernel); per lane: row inside the tile and swizzled chunk are tile-independent
  int prow[C::PA], pchunk[C::PA];
#pragma unroll
  for (int i = 0; i < C::PA; ++i) {
    prow[i] = (wave + C::NW * i) * C::RPP + lane / C::CPR;
    pchunk[i] = swz16<C::RB>(prow[i], lane % C::CPR) * 8;
  }
  auto stage = [&](int buf, int m0, int n0, int k0) {
    LdsPtr ta = smem + buf * C::STAGE + wave * 1024;
    LdsPtr tb = ta + C::TILE_A;
#pragma unroll
    for (int i = 0; i < C::PA; ++i)
      dma16_global(reinterpret_cast<const bf16*>(g.A) + (long)min(m0 + prow[i], g.M - 1) * g.lda + pchunk[i] + k0, ta + i * C::NW * 1024);
#pragma unroll
    for (int i = 0; i < C::PB; ++i)
      dma16_global(reinterpret_cast<const bf16*>(g.B) + (long)(n0 + prow[i]) * g.ldb + pchunk[i] + k0, tb + i * C::NW * 1024);
  };
  auto coords = [&](int t, int& m0, int& n0) {
    const int lin = xcd_remap(t, ntiles);
    m0 = (lin / tiles_n) * C::BM; n0 = (lin % tiles_n) * C::BN;
  };

  int t = blockIdx.x;
  if (t >= ntiles) return;
  int m0, n0;
  coords(t, m0, n0);
  stage(0, m0, n0, 0);
  bool stores_pending = false;        // the previous tile's epilogue went through the whole-line path with every row valid
  for (; t < ntiles; t += gridDim.x) {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int tn = t + gridDim.x;
    int m0n = 0, n0n = 0;
    if (tn < ntiles) coords(tn, m0n, n0n);
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      // stage kt has landed (it is older than anything issued since), every wave is done with the other buffer
      if (kt == 0 && stores_pending) wait_vm_barrier<P::NST>();
      else wait_vm_barrier<0>();
      if (kt + 1 < nk) stage(cur ^ 1, m0, n0, (kt + 1) * 64);
      else if (tn < ntiles) stage(cur ^ 1, m0n, n0n, 0);
      const lds_char* ta = smem + cur * C::STAGE;
      const lds_char* tb = ta + C::TILE_A;
#pragma unroll
      for (int ks = 0; ks < 64; ks += 16) {
        Frag<bf16> fa[4], fb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = load_frag_row<bf16, C::RB>(tb, wn * 64 + j * 32 + li, ks);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = load_frag_row<bf16, C::RB>(ta, wm * 128 + i * 32 + li, ks);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mma32(fb[j], fa[i], acc[i][j]);
      }
    }
    // every wave is done reading the last stage (buffer 1) before the epilogue's strips go there; the next tile's first stage
    // (buffer 0) stays in flight
    asm volatile("s_barrier" ::: "memory");
    nt_big_epilogue<bf16, FLAGS, 2>(g, acc, m0, n0, wave, lane, smem + P::EP_OFF);
    stores_pending = m0 + C::BM <= g.M && g.wide_t;
    m0 = m0n; n0 = n0n;
  }
}


static int g_nt_persist = 0;   // workgroups of the persistent kernel
static int g_exp_variant = 0;
template <int FLAGS> static bool launch_persist_t(const GemmNT& g, hipStream_t stream) {
  using C = BigCfg<2, 64>;
  using P = PersistCfg<FLAGS>;
  if (g_nt_persist <= 0 || (g.K / 64) % 2 || g.N % C::BN || !g.wide_t) return false;
  if ((FLAGS & EPI_OUT_F32) && (g.ld_out_f32 % 4)) return false;
  static LdsAllowance allowance;
  allowance.ensure(gemm_nt_persist_kernel<FLAGS>, P::LDS);
  const int tiles = ((g.M + C::BM - 1) / C::BM) * (g.N / C::BN);
  hipLaunchKernelGGL((gemm_nt_persist_kernel<FLAGS>), dim3(std::min(tiles, g_nt_persist)), dim3(512), P::LDS, stream, g);
  return true;
}
template <int FLAGS> static bool launch_ring_t(const GemmNT& g, hipStream_t stream) {
  using C = BigCfg<2, 32>;
  if (g_exp_variant != 1 || g.K % 32) return false;
  static LdsAllowance allowance;
  allowance.ensure(gemm_nt_ring_kernel<FLAGS>, RING_NS * C::STAGE);
  const int tiles = ((g.M + C::BM - 1) / C::BM) * ((g.N + C::BN - 1) / C::BN);
  hipLaunchKernelGGL((gemm_nt_ring_kernel<FLAGS>), dim3(tiles), dim3(512), RING_NS * C::STAGE, stream, g);
  return true;
}

}  // namespace pfn

// variant 1: ring, variant 2: persistent with `wgs` workgroups.  Flags: the epilogue combinations of the product's launch_big; returns 0 / -1 (unsupported shape or flags).
extern "C" int pfn_exp_gemm_nt(int variant, int wgs, const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, int flags, const float* bias,
                               const void* aux, int64_t ld_aux, const float* resid, int64_t ld_resid, float* out_f32, int64_t ld_out_f32, void* out_t, int64_t ld_out_t,
                               void* out2_t, int64_t ld_out2, void* stream) {
  using namespace pfn;
  GemmNT g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K; g.flags = flags; g.bias = bias; g.aux = aux; g.ld_aux = ld_aux; g.resid = resid; g.ld_resid = ld_resid;
  g.out_f32 = out_f32; g.ld_out_f32 = ld_out_f32; g.out_t = out_t; g.ld_out_t = ld_out_t; g.out2_t = out2_t; g.ld_out2 = ld_out2;
  nt_prepare(g, PFN_PREC_BF16);
  if (!g.vec_ok || g.K % 64 || g.N % 4) return -1;
  g_exp_variant = variant; g_nt_persist = wgs;
  hipStream_t s = (hipStream_t)stream;
  switch (flags) {
#define PFN_EXP_CASE(F) case (F): return (variant == 1 ? launch_ring_t<(F)>(g, s) : launch_persist_t<(F)>(g, s)) ? 0 : -1;
    PFN_EXP_CASE(EPI_BIAS | EPI_OUT_T)
    PFN_EXP_CASE(EPI_BIAS | EPI_GELU | EPI_OUT_T | EPI_OUT2_T)
    PFN_EXP_CASE(EPI_GELU_BWD | EPI_OUT_T)
    PFN_EXP_CASE(EPI_RESID_T | EPI_OUT_T)
    PFN_EXP_CASE(EPI_RESID_T | EPI_OUT_F32)
    PFN_EXP_CASE(EPI_OUT_T)
    PFN_EXP_CASE(EPI_OUT_F32)
#undef PFN_EXP_CASE
    default: return -1;
  }
}
