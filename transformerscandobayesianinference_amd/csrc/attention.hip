// Fused QK^T -> masked online softmax -> PV ("flash") attention for the PFN eval-position mask,
// forward and backward, on gfx950 MFMA.
//
// Replaces the scores/softmax/PV part of torch multi_head_attention_forward as called from
// nn.TransformerEncoderLayer inside TransformerModel.forward (reference transformer.py:84) with the
// additive mask of TransformerModel.generate_D_q_matrix (transformer.py:34-41):
//     query i may attend key j   iff   j < sep   or   j == i
// The [S,S] mask is never built: the kernels stream key tiles over [0, sep) and treat the
// "self" key of a test row (i >= sep) as the initial state of the online softmax (forward) or
// as a row-wise correction term (backward).
//
// Formulation (all MFMAs are 32x32, "swapped" so softmax statistics are lane-local):
//   fwd / dQ kernels : lane <-> query.   S^T = K.Q^T, O^T = V^T.P^T, dP^T = V.dO^T, dQ^T = K^T.dS^T
//   dK/dV kernel     : lane <-> key.     S = Q.K^T, dP = dO.V^T, dV^T = dO^T.P, dK^T = Q^T.dS
// Operands that are k-major in memory (V, K^T, Q^T, dO^T) come from LDS through
// ds_read_b64_tr_b16 (bf16) in the accumulator-order slot mapping M2, so P / dS never move
// between lanes.
#include <algorithm>
#include <type_traits>
#include "pfn_device.h"
#include "pfn_kernels.h"

namespace pfn {

template <typename T, int D> struct AttnCfg {
  static constexpr int RB = D * (int)sizeof(T);              // bytes per K/V/Q row of one head
  static constexpr int KVB = (sizeof(T) == 2 && D <= 128) ? 64 : 32;  // keys per LDS tile
  static constexpr int NKB = KVB / 32;
  static constexpr int NKK = D / 16;
  static constexpr int NDB = D / 32;
  static constexpr int TILE = KVB * RB;
  static constexpr int RS = PadStride<RB>::ROW, CS = PadStride<RB>::COL;   // padded LDS row strides (pfn_device.h)
  static constexpr int RIMG = KVB * RS, CIMG = KVB * CS;                  // bytes of one row / col image of a tile
  // bf16 up to head dim 128 (product path): 8 waves share every K/V (or Q/dO) tile and the kernel is held to 256
  // registers, so two waves are resident per SIMD and one wave's MFMAs cover the other's softmax / LDS waits.
  // Head dim 256 and the exact-f32 parity mode have twice the accumulator / fragment registers and keep 4 waves
  // with the whole register file (at 256 registers they spill into scratch inside the tile loop).
  static constexpr int NW = (sizeof(T) == 2 && D <= 128) ? 8 : 4;
  static constexpr int NT = NW * 64;        // threads per workgroup
  static constexpr int QBLK = NW * 32;      // query (or key) rows per workgroup
};

template <typename T> PFN_DEV Frag<T> load_frag_global(const T* p) {
  Frag<T> f;
  if constexpr (sizeof(T) == 2) {
    f.v = *reinterpret_cast<const bf16x8*>(p);
  } else {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { f.v[e] = a[e]; f.v[4 + e] = b[e]; }
  }
  return f;
}
template <typename T> PFN_DEV void store_frag_global(T* p, const float (&x)[8]) {
  if constexpr (sizeof(T) == 2) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)x[e];
    *reinterpret_cast<bf16x8*>(p) = v;
  } else {
    f32x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = x[e]; b[e] = x[4 + e]; }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
  }
}
template <typename T> PFN_DEV float frag_get(const Frag<T>& f, int e) { return (float)f.v[e]; }
template <typename T> PFN_DEV float dot8(const Frag<T>& a, const Frag<T>& b) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += frag_get(a, e) * frag_get(b, e);
  return s;
}
template <typename T> PFN_DEV f32x4 load4(const T* p) {
  f32x4 r;
  if constexpr (sizeof(T) == 2) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (float)v[e];
  } else r = *reinterpret_cast<const f32x4*>(p);
  return r;
}
template <typename T> PFN_DEV void store4(T* p, f32x4 x) {
  if constexpr (sizeof(T) == 2) {
    bf16x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (bf16)x[e];
    *reinterpret_cast<bf16x4*>(p) = v;
  } else *reinterpret_cast<f32x4*>(p) = x;
}
PFN_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Workgroup -> (row block, head, dataset) for a 1-D launch of nblk * H * B workgroups.  Hardware deals
// consecutive workgroup ids round-robin over the 8 XCDs (each with a private L2); every row block of one
// (dataset, head) streams the SAME K/V (or Q/dO) rows, so the ids are remapped (bijectively, as in gemm.hip) to
// give each XCD whole (dataset, head) groups: the operand rows are then fetched into one L2 once instead of
// into all eight.
struct AttnBlock { int blk, hd, b; };
PFN_DEV AttnBlock attn_block(int nblk, int H) {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  AttnBlock o;
  o.blk = id % nblk;
  const int bh = id / nblk;
  o.hd = bh % H;
  o.b = bh / H;
  return o;
}

// Profiling builds only (tools/exp_attn_variants.py): -DPFN_ATTN_ABLATE=<bits> drops, in every main loop, the
// global tile requests (1), the LDS tile writes (2), the barrier (4).  Results are then garbage; the timing
// differences price the three parts (MI355X, north-star shape: requests 9-14 %, writes 6-7 %, barrier 4 %).
#ifndef PFN_ATTN_ABLATE
#define PFN_ATTN_ABLATE 0
#endif
constexpr int ABL = PFN_ATTN_ABLATE;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float RESCALE_THR = 6.f;  // log2 units: lazily raised running max of the online softmax (see attn_fwd_kernel)

// =============================================================================================
// forward
// =============================================================================================
template <typename T, int D>
__global__ __launch_bounds__((AttnCfg<T, D>::NT)) void attn_fwd_kernel(AttnArgs a) {
  using C = AttnCfg<T, D>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr smem = lds_cast(smem_raw);
  // two K buffers (row images) followed by three V buffers (col images): P.V runs one tile behind Q.K (below)
  auto Kt = [&](int buf) { return smem + buf * C::RIMG; };
  auto Vt = [&](int buf) { return smem + 2 * C::RIMG + buf * C::CIMG; };

  const AttnBlock wg = attn_block((a.S + C::QBLK - 1) / C::QBLK, a.H);
  const int b = wg.b, hd = wg.hd;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, li = lane & 31;
  const long rs = 3L * a.E;
  const T* base = reinterpret_cast<const T*>(a.qkv) + (long)b * a.S * rs;
  const T* Qp = base + hd * D;
  const T* Kp = base + a.E + hd * D;
  const T* Vp = base + 2 * a.E + hd * D;
  const int qi = wg.blk * C::QBLK + wave * 32 + li;
  const bool qvalid = qi < a.S;
  const int qc = min(qi, a.S - 1);
  const int sep = a.sep;
  const float scale_log2 = rsqrtf((float)D) * LOG2E;

  Frag<T> qf[C::NKK];
#pragma unroll
  for (int kk = 0; kk < C::NKK; ++kk) qf[kk] = load_frag_global<T>(Qp + (long)qc * rs + kk * 16 + 8 * h);

  // initial state: the self key for test rows, empty for train rows.  The self K / V rows are only fetched by waves
  // that hold a test row at all (wave-uniform branch): 7 of 8 waves at the north star skip 24 loads per lane.
  const bool is_test = qc >= sep;
  const bool wave_has_test = wg.blk * C::QBLK + wave * 32 + 31 >= sep;
  float m = -1e30f, lsum = 0.f;
  f32x16 o[C::NDB];
#pragma unroll
  for (int db = 0; db < C::NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  if (wave_has_test) {
    float part = 0.f;
#pragma unroll
    for (int kk = 0; kk < C::NKK; ++kk) part += dot8(qf[kk], load_frag_global<T>(Kp + (long)qc * rs + kk * 16 + 8 * h));
    part += __shfl_xor(part, 32, 64);
    m = is_test ? part * scale_log2 : -1e30f;
    lsum = (is_test && h == 0) ? 1.f : 0.f;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        f32x4 v = load4<T>(Vp + (long)qc * rs + db * 32 + 8 * rg + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[db][4 * rg + e] = is_test ? v[e] : 0.f;
      }
  }

  const int nfull = sep / C::KVB;
  const int ntiles = (sep + C::KVB - 1) / C::KVB;
  TileStage<T, C::KVB, C::RB, C::NT> sk, sv;
  if (ntiles > 0) {
    sk.issue(Kp, rs, sep, D);
    sv.issue(Vp, rs, sep, D);
    sk.template commit_p<C::RS>(Kt(0));
    sv.template commit_p<C::CS>(Vt(0));
    if (ntiles > 1) {   // tile 1 stays in flight across the barrier (see the loop)
      sk.issue(Kp + (long)C::KVB * rs, rs, sep - C::KVB, D);
      sv.issue(Vp + (long)C::KVB * rs, rs, sep - C::KVB, D);
    }
  }
  __syncthreads();

  // Software pipeline over key tiles, P.V ONE TILE BEHIND Q.K:
  //   iteration t :  S = K_t Q^T (MFMA)  |  running-max check (rare rescale)  |  { P_t = exp2(S - m)  +  O += V_{t-1} P_{t-1} }
  // The exponentials of tile t (the bulk of the vector work: fma, exp, row sums, bf16 packing) and the 16 P.V
  // MFMAs of tile t-1 are independent and sit in one basic block, so the matrix pipe runs under the vector ALU
  // instead of after it.  P_{t-1} waits in registers as packed operand fragments; V needs three LDS buffers
  // (tile t-1 is read while tile t+1 is written), K two.
  // The running max is only raised -- and O, the row sums and the pending P_{t-1} only rescaled -- when some row's
  // tile max exceeds it by more than RESCALE_THR (log2 units): P stays below 2^THR, which neither the bf16
  // operand rounding (relative) nor the f32 sums notice, and the O-wide rescale runs on the first tile and then
  // almost never.  The key mask only runs on the ragged last tile.
  constexpr int NPF = C::KVB / 16;
  Frag<T> pf[NPF];                       // P_{t-1}, in the slot order the P.V MFMA consumes
  auto pv_prev = [&](int vb) __attribute__((always_inline)) {
    const lds_char* vt = Vt(vb);
#pragma unroll
    for (int c = 0; c < NPF; ++c)
#pragma unroll
      for (int db = 0; db < C::NDB; ++db)
        o[db] = mma32(load_frag_tr_p<T, C::CS, 2>(vt, c * 16, db * 32), pf[c], o[db]);
  };
  int vb_prev = 0, vb_cur = 0;           // V buffer of tile t-1 / tile t
  for (int t = 0; t < ntiles; ++t) {
    const int k0 = t * C::KVB;
    // all K fragments of the tile are requested before anything else: the LDS round trip (>100 cycles under load)
    // then overlaps the global-load issue below and the first MFMAs instead of stalling each one
    const lds_char* kt = Kt(t & 1);
    Frag<T> kfr[C::NKK][C::NKB];
#pragma unroll
    for (int kk = 0; kk < C::NKK; ++kk)
#pragma unroll
      for (int kb = 0; kb < C::NKB; ++kb) kfr[kk][kb] = load_frag_row_p<T, C::RS>(kt, kb * 32 + li, kk * 16);
    PFN_PIN_LDS_MFMA();
    f32x16 st[C::NKB];
#pragma unroll
    for (int kb = 0; kb < C::NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < C::NKK; ++kk)
#pragma unroll
      for (int kb = 0; kb < C::NKB; ++kb) st[kb] = mma32(kfr[kk][kb], qf[kk], st[kb]);
    PFN_PIN_LDS_MFMA();
    // Global -> LDS staging, one tile ahead with the loads in flight across the barrier: the staged registers hold
    // tile t+1 (requested one iteration ago); they are written to the buffers tile t-1 has released -- the wait for
    // them sits behind the queued Q.K MFMAs -- and at once re-used for the request of tile t+2.  On the last tile the
    // registers are stale and go to buffers nobody reads again (a branch here would split the block and strand the
    // exponentials behind the P.V MFMAs).
    const int vb_next = vb_cur == 2 ? 0 : vb_cur + 1;   // the slot tile t-2 has released
    if (!(ABL & 2)) {
      sk.template commit_p<C::RS>(Kt((t + 1) & 1));
      sv.template commit_p<C::CS>(Vt(vb_next));
    }
    if (!(ABL & 1) && t + 2 < ntiles) {
      const long k2 = k0 + 2 * C::KVB;
      sk.issue(Kp + k2 * rs, rs, sep - (int)k2, D);
      sv.issue(Vp + k2 * rs, rs, sep - (int)k2, D);
    }
    if (t == nfull) {   // ragged last tile: keys >= sep do not exist
#pragma unroll
      for (int kb = 0; kb < C::NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + kb * 32 + acc_row(r, lane) >= sep) st[kb][r] = -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < C::NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * scale_log2;
    if (__builtin_amdgcn_ballot_w64(mc > m + RESCALE_THR) != 0) {
      const float m_new = fmaxf(m, mc);
      const float alpha = fast_exp2(m - m_new);
      m = m_new;
      lsum *= alpha;
#pragma unroll
      for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
      if (t > 0) {     // P_{t-1} is still at the old maximum
#pragma unroll
        for (int c = 0; c < NPF; ++c)
#pragma unroll
          for (int e = 0; e < 8; ++e) pf[c].set(e, frag_get(pf[c], e) * alpha);
      }
    }
    if (t > 0) {
      // one basic block: exponentials of tile t + P.V MFMAs of tile t-1
      float rsum = 0.f;
#pragma unroll
      for (int kb = 0; kb < C::NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = fast_exp2(__builtin_fmaf(st[kb][r], scale_log2, -m));
          st[kb][r] = p;
          rsum += p;
        }
      lsum += rsum;
      pv_prev(vb_prev);
    } else {
      float rsum = 0.f;
#pragma unroll
      for (int kb = 0; kb < C::NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = fast_exp2(__builtin_fmaf(st[kb][r], scale_log2, -m));
          st[kb][r] = p;
          rsum += p;
        }
      lsum += rsum;
    }
#pragma unroll
    for (int c = 0; c < NPF; ++c) pf[c] = acc_to_frag<T>(st[c >> 1], c & 1);
    vb_prev = vb_cur;
    vb_cur = vb_next;
    if (!(ABL & 4)) __syncthreads();
  }
  if (ntiles > 0) pv_prev(vb_prev);

  lsum += __shfl_xor(lsum, 32, 64);
  const float inv = 1.f / lsum;
  {
    T* out = reinterpret_cast<T*>(a.ctx) + ((long)b * a.S + qc) * a.E + hd * D;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = o[db][r] * inv;
      store_row_block<T>(out + db * 32, v, h, qvalid);
    }
    if (qvalid && h == 0) a.lse[((long)b * a.H + hd) * a.S + qi] = (m + __builtin_amdgcn_logf(lsum)) * LN2;
  }
}

// =============================================================================================
// backward, step 0: delta[b,h,i] = sum_d dO[i,d] * O[i,d]
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnArgs a, int D) {
  // one 16-lane group per (token, head) pair; grid-stride over B*S*H pairs
  const long pairs = (long)a.B * a.S * a.H;
  const int sub = threadIdx.x & 15;
  for (long pr = (long)blockIdx.x * 16 + (threadIdx.x >> 4); pr < pairs; pr += (long)gridDim.x * 16) {
    const long tok = pr / a.H;
    const int hd = (int)(pr % a.H);
    const T* o = reinterpret_cast<const T*>(a.ctx) + tok * a.E + hd * D;
    const T* d = reinterpret_cast<const T*>(a.dctx) + tok * a.E + hd * D;
    float s = 0.f;
    for (int c = sub * 4; c < D; c += 64) {
      f32x4 x = load4<T>(o + c), y = load4<T>(d + c);
      s += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (sub == 0) {
      const long bb = tok / a.S, i = tok % a.S;
      a.delta[(bb * a.H + hd) * a.S + i] = s;
    }
  }
}

// =============================================================================================
// backward, dQ (+ the self-key terms of test rows: dQ_i, dK_i, dV_i for i >= sep)
// =============================================================================================
template <typename T, int D>
__global__ __launch_bounds__((AttnCfg<T, D>::NT)) void attn_bwd_dq_kernel(AttnArgs a) {
  using C = AttnCfg<T, D>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr smem = lds_cast(smem_raw);
  // per buffer: K (row image), K (col image, transposed frags), V (row image)
  constexpr int DQ_BUF = 2 * C::RIMG + C::CIMG;
  auto Kr = [&](int buf) { return smem + buf * DQ_BUF; };
  auto Kc = [&](int buf) { return smem + buf * DQ_BUF + C::RIMG; };
  auto Vr = [&](int buf) { return smem + buf * DQ_BUF + C::RIMG + C::CIMG; };

  const AttnBlock wg = attn_block((a.S + C::QBLK - 1) / C::QBLK, a.H);
  const int b = wg.b, hd = wg.hd;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, li = lane & 31;
  const long rs = 3L * a.E;
  const T* base = reinterpret_cast<const T*>(a.qkv) + (long)b * a.S * rs;
  const T* Qp = base + hd * D;
  const T* Kp = base + a.E + hd * D;
  const T* Vp = base + 2 * a.E + hd * D;
  T* dbase = reinterpret_cast<T*>(a.dqkv) + (long)b * a.S * rs;
  const int qi = wg.blk * C::QBLK + wave * 32 + li;
  const bool qvalid = qi < a.S;
  const int qc = min(qi, a.S - 1);
  const int sep = a.sep;
  const float scale = rsqrtf((float)D);
  const float scale_log2 = scale * LOG2E;
  const T* dOp = reinterpret_cast<const T*>(a.dctx) + ((long)b * a.S + qc) * a.E + hd * D;

  Frag<T> qf[C::NKK], dof[C::NKK];
#pragma unroll
  for (int kk = 0; kk < C::NKK; ++kk) {
    qf[kk] = load_frag_global<T>(Qp + (long)qc * rs + kk * 16 + 8 * h);
    dof[kk] = load_frag_global<T>(dOp + kk * 16 + 8 * h);
  }
  const long stat = ((long)b * a.H + hd) * a.S + qc;
  const float lse2 = a.lse[stat] * LOG2E;
  const float delta = a.delta[stat];

  f32x16 dq[C::NDB];
#pragma unroll
  for (int db = 0; db < C::NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;

  const int nfull = sep / C::KVB;
  const bool has_edge = (sep % C::KVB) != 0;
  const int ntiles = nfull + (has_edge ? 1 : 0);
  TileStage<T, C::KVB, C::RB, C::NT> sk, sv;
  if (ntiles > 0) {
    sk.issue(Kp, rs, sep, D);
    sv.issue(Vp, rs, sep, D);
    sk.template commit_p<C::RS>(Kr(0));
    sk.template commit_p<C::CS>(Kc(0));
    sv.template commit_p<C::RS>(Vr(0));
    if (ntiles > 1) {   // tile 1 stays in flight across the barrier (staging as in attn_fwd_kernel)
      sk.issue(Kp + (long)C::KVB * rs, rs, sep - C::KVB, D);
      sv.issue(Vp + (long)C::KVB * rs, rs, sep - C::KVB, D);
    }
  }
  __syncthreads();
  // one key tile; BUF / EDGE compile-time as in the forward kernel
  auto tile = [&](auto buf_c, auto edge_c, int t) {
    constexpr int BUF = decltype(buf_c)::value;
    constexpr bool EDGE = decltype(edge_c)::value;
    const int k0 = t * C::KVB;
    const lds_char* kr = Kr(BUF);
    const lds_char* kc = Kc(BUF);
    const lds_char* vr = Vr(BUF);
#pragma unroll
    for (int kb = 0; kb < C::NKB; ++kb) {   // 32 keys at a time: only one S / dP pair is live
      // operand fragments of the S / dP products run PD k-steps ahead of the MFMAs that consume them
      constexpr int PD = C::NKK < 4 ? C::NKK : 4;
      Frag<T> kfr[C::NKK], vfr[C::NKK];
#pragma unroll
      for (int kk = 0; kk < PD; ++kk) {
        kfr[kk] = load_frag_row_p<T, C::RS>(kr, kb * 32 + li, kk * 16);
        vfr[kk] = load_frag_row_p<T, C::RS>(vr, kb * 32 + li, kk * 16);
      }
      PFN_PIN_LDS_MFMA();
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < C::NKK; ++kk) {
        if (kk + PD < C::NKK) {
          kfr[kk + PD] = load_frag_row_p<T, C::RS>(kr, kb * 32 + li, (kk + PD) * 16);
          vfr[kk + PD] = load_frag_row_p<T, C::RS>(vr, kb * 32 + li, (kk + PD) * 16);
        }
        st = mma32(kfr[kk], qf[kk], st);
        dp = mma32(vfr[kk], dof[kk], dp);
        PFN_PIN_LDS_MFMA();
      }
      if (kb == 0) {   // staged tile t+1 -> the other buffer (stale and unread after the last tile), then request t+2
        if (!(ABL & 2)) {
          sk.template commit_p<C::RS>(Kr(BUF ^ 1));
          sk.template commit_p<C::CS>(Kc(BUF ^ 1));
          sv.template commit_p<C::RS>(Vr(BUF ^ 1));
        }
        if (!(ABL & 1) && t + 2 < ntiles) {
          const long k2 = k0 + 2 * C::KVB;
          sk.issue(Kp + k2 * rs, rs, sep - (int)k2, D);
          sv.issue(Vp + k2 * rs, rs, sep - (int)k2, D);
        }
      }
      // transposed K fragments of the dQ products: the first half is requested before the softmax arithmetic, the
      // second before the first half's MFMAs
      Frag<T> cf[2][C::NDB];
#pragma unroll
      for (int db = 0; db < C::NDB; ++db) cf[0][db] = load_frag_tr_p<T, C::CS, 2>(kc, kb * 32, db * 32);
      PFN_PIN_LDS_MFMA();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = fast_exp2(__builtin_fmaf(st[r], scale_log2, -lse2));
        if (EDGE && (k0 + kb * 32 + acc_row(r, lane) >= sep)) p = 0.f;
        st[r] = p * (dp[r] - delta);  // dS (unscaled)
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const Frag<T> dsf = acc_to_frag<T>(st, c);
        if (c == 0) {
#pragma unroll
          for (int db = 0; db < C::NDB; ++db) cf[1][db] = load_frag_tr_p<T, C::CS, 2>(kc, kb * 32 + 16, db * 32);
        }
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) dq[db] = mma32(cf[c][db], dsf, dq[db]);
        PFN_PIN_LDS_MFMA();
      }
    }
    if (!(ABL & 4)) __syncthreads();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  {
    int t = 0;
    for (; t + 2 <= nfull; t += 2) {
      tile(I0{}, std::false_type{}, t);
      tile(I1{}, std::false_type{}, t + 1);
    }
    if (t < nfull) {
      tile(I0{}, std::false_type{}, t);
      if (has_edge) tile(I1{}, std::true_type{}, t + 1);
    } else if (has_edge) {
      tile(I0{}, std::true_type{}, t);
    }
  }

  // self key of test rows
  const bool is_test = qc >= sep;
  float ds_self = 0.f, p_self = 0.f;
  const bool wave_has_test = wg.blk * C::QBLK + wave * 32 + 31 >= sep;
  if (wave_has_test) {   // only waves that hold a test row fetch the self K / V rows
    float tq = 0.f, dpv = 0.f;
#pragma unroll
    for (int kk = 0; kk < C::NKK; ++kk) {
      tq += dot8(qf[kk], load_frag_global<T>(Kp + (long)qc * rs + kk * 16 + 8 * h));
      dpv += dot8(dof[kk], load_frag_global<T>(Vp + (long)qc * rs + kk * 16 + 8 * h));
    }
    tq += __shfl_xor(tq, 32, 64);
    dpv += __shfl_xor(dpv, 32, 64);
    if (is_test) {
      p_self = fast_exp2(tq * scale_log2 - lse2);
      ds_self = p_self * (dpv - delta);
    }
  }
  {
    T* dQo = dbase + (long)qc * rs + hd * D;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db) {
      float v[16];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d0 = db * 32 + 8 * rg + 4 * h;
        f32x4 kv = {0.f, 0.f, 0.f, 0.f};
        if (wave_has_test) kv = load4<T>(Kp + (long)qc * rs + d0);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * rg + e] = (dq[db][4 * rg + e] + ds_self * kv[e]) * scale;
      }
      store_row_block<T>(dQo + db * 32, v, h, qvalid);
    }
  }
  if (qvalid) {
    if (is_test) {
      T* dKo = dbase + (long)qi * rs + a.E + hd * D;
      T* dVo = dbase + (long)qi * rs + 2 * a.E + hd * D;
#pragma unroll
      for (int kk = 0; kk < C::NKK; ++kk) {
        float xk[8], xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xk[e] = ds_self * scale * frag_get(qf[kk], e);
          xv[e] = p_self * frag_get(dof[kk], e);
        }
        store_frag_global<T>(dKo + kk * 16 + 8 * h, xk);
        store_frag_global<T>(dVo + kk * 16 + 8 * h, xv);
      }
    }
  }
}

// =============================================================================================
// backward, dK and dV for the train keys [0, sep): a workgroup owns QBLK keys (one per lane, 32 per
// wave) and streams all S queries.  Two passes (MODE 0: dV = P^T dO, MODE 1: dK = dS^T Q) instead
// of one fused kernel: a fused pass needs dK and dV accumulators plus the K and V fragments of
// the wave's keys -- 240 registers before any temporary -- and would run one wave per SIMD with
// nothing to cover its LDS waits and softmax; each split pass fits two (dK) or three (dV) waves per
// SIMD at the price of recomputing S once.
// Keys >= sep in the last key block run on clamped (finite or not: never stored, and a lane's key
// column never mixes with another lane's) data instead of being masked.
// =============================================================================================
template <typename T, int D, int MODE>
__global__ __launch_bounds__((AttnCfg<T, D>::NT)) void attn_bwd_dkv_kernel(AttnArgs a) {
  using C = AttnCfg<T, D>;
  constexpr int QB = C::KVB;                 // queries per tile (64 for bf16 up to D = 128, else 32)
  constexpr int NQB = QB / 32;
  // images per buffer -- dV: Q rows, dO cols;  dK: Q rows, Q cols, dO rows;  then lse2[QB], delta[QB]
  constexpr int IMG_BYTES = C::RIMG + C::CIMG + (MODE == 1 ? C::RIMG : 0);
  constexpr int BUF_BYTES = IMG_BYTES + 2 * QB * 4;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  LdsPtr smem = lds_cast(smem_raw);
  auto Img = [&](int buf, int i) { return smem + buf * BUF_BYTES + (i == 0 ? 0 : i == 1 ? C::RIMG : C::RIMG + C::CIMG); };
  auto St = [&](int buf) { return smem + buf * BUF_BYTES + IMG_BYTES; };  // [lse2 x QB][delta x QB]

  const AttnBlock wg = attn_block((a.sep + C::QBLK - 1) / C::QBLK, a.H);
  const int b = wg.b, hd = wg.hd;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, li = lane & 31;
  const long rs = 3L * a.E;
  const T* base = reinterpret_cast<const T*>(a.qkv) + (long)b * a.S * rs;
  const T* Qp = base + hd * D;
  const T* Kp = base + a.E + hd * D;
  const T* Vp = base + 2 * a.E + hd * D;
  const T* dOp = reinterpret_cast<const T*>(a.dctx) + (long)b * a.S * a.E + hd * D;
  T* dbase = reinterpret_cast<T*>(a.dqkv) + (long)b * a.S * rs;
  const int sep = a.sep;
  const int key = wg.blk * C::QBLK + wave * 32 + li;
  const bool kvalid = key < sep;
  const int kc = min(key, a.S - 1);
  const float scale = rsqrtf((float)D);
  const float scale_log2 = scale * LOG2E;
  const float* lse_g = a.lse + ((long)b * a.H + hd) * a.S;
  const float* delta_g = a.delta + ((long)b * a.H + hd) * a.S;

  Frag<T> kf[C::NKK], vf[MODE == 1 ? C::NKK : 1];
#pragma unroll
  for (int kk = 0; kk < C::NKK; ++kk) {
    kf[kk] = load_frag_global<T>(Kp + (long)kc * rs + kk * 16 + 8 * h);
    if constexpr (MODE == 1) vf[kk] = load_frag_global<T>(Vp + (long)kc * rs + kk * 16 + 8 * h);
  }
  f32x16 acc[C::NDB];
#pragma unroll
  for (int db = 0; db < C::NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;

  const int ntiles = (a.S + QB - 1) / QB;
  TileStage<T, QB, C::RB, C::NT> sq, so;
  float st_reg = 0.f;  // threads 0..2*QB-1 stage lse2 / delta
  auto stage_stats = [&](int q0) {
    if (threadIdx.x < 2 * QB) {
      const int q = q0 + (threadIdx.x & (QB - 1));
      if (threadIdx.x < QB) st_reg = (q < a.S) ? lse_g[q] * LOG2E : 1e30f;
      else st_reg = (MODE == 1 && q < a.S) ? delta_g[q] : 0.f;
    }
  };
  auto commit_all = [&](int buf) {
    sq.template commit_p<C::RS>(Img(buf, 0));
    if constexpr (MODE == 0) {
      so.template commit_p<C::CS>(Img(buf, 1));
    } else {
      sq.template commit_p<C::CS>(Img(buf, 1));
      so.template commit_p<C::RS>(Img(buf, 2));
    }
    if (threadIdx.x < 2 * QB) lds_write_f32(St(buf) + threadIdx.x * 4, st_reg);
  };
  sq.issue(Qp, rs, a.S, D);
  so.issue(dOp, a.E, a.S, D);
  stage_stats(0);
  commit_all(0);
  auto request = [&](int t) {   // tile t -> staging registers
    const long q1 = (long)t * QB;
    sq.issue(Qp + q1 * rs, rs, a.S - (int)q1, D);
    so.issue(dOp + q1 * a.E, a.E, a.S - (int)q1, D);
    stage_stats((int)q1);
  };
  if (ntiles > 1) request(1);   // stays in flight across the barrier (staging as in attn_fwd_kernel)
  __syncthreads();

  auto tile = [&](auto buf_c, int t) {
    constexpr int BUF = decltype(buf_c)::value;
    const lds_char* qr = Img(BUF, 0);
    const lds_char* stt = St(BUF);
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      // operand fragments run PD k-steps ahead of the MFMAs that consume them (see attn_bwd_dq_kernel)
      constexpr int PD = C::NKK < 4 ? C::NKK : 4;
      Frag<T> qfr[C::NKK], ofr[MODE == 1 ? C::NKK : 1];
#pragma unroll
      for (int kk = 0; kk < PD; ++kk) {
        qfr[kk] = load_frag_row_p<T, C::RS>(qr, qb * 32 + li, kk * 16);
        if constexpr (MODE == 1) ofr[kk] = load_frag_row_p<T, C::RS>(Img(BUF, 2), qb * 32 + li, kk * 16);
      }
      PFN_PIN_LDS_MFMA();
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < C::NKK; ++kk) {
        if (kk + PD < C::NKK) {
          qfr[kk + PD] = load_frag_row_p<T, C::RS>(qr, qb * 32 + li, (kk + PD) * 16);
          if constexpr (MODE == 1) ofr[kk + PD] = load_frag_row_p<T, C::RS>(Img(BUF, 2), qb * 32 + li, (kk + PD) * 16);
        }
        s = mma32(qfr[kk], kf[kk], s);
        if constexpr (MODE == 1) dp = mma32(ofr[kk], vf[kk], dp);
        PFN_PIN_LDS_MFMA();
      }
      if (qb == 0) {   // staged tile t+1 -> the other buffer (stale and unread after the last tile), then request t+2
        if (!(ABL & 2)) commit_all(BUF ^ 1);
        if (!(ABL & 1) && t + 2 < ntiles) request(t + 2);
      }
      // transposed dO / Q fragments of the second product: first half requested before the softmax arithmetic
      const lds_char* col = Img(BUF, 1);   // dO (dV pass) or Q (dK pass), read transposed
      Frag<T> cf[2][C::NDB];
#pragma unroll
      for (int db = 0; db < C::NDB; ++db) cf[0][db] = load_frag_tr_p<T, C::CS, 2>(col, qb * 32, db * 32);
      PFN_PIN_LDS_MFMA();
      // rows of s/dp are queries: acc_row(r) = 8*(r>>2) + 4h + (r&3)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const f32x4 l2 = __builtin_bit_cast(f32x4, lds_read16(stt + (qb * 32 + 8 * rg + 4 * h) * 4));
        f32x4 dl = {0.f, 0.f, 0.f, 0.f};
        if constexpr (MODE == 1) dl = __builtin_bit_cast(f32x4, lds_read16(stt + (QB + qb * 32 + 8 * rg + 4 * h) * 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * rg + e;
          const float p = fast_exp2(__builtin_fmaf(s[r], scale_log2, -l2[e]));
          if constexpr (MODE == 0) s[r] = p;
          else s[r] = p * (dp[r] - dl[e]);
        }
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const Frag<T> f = acc_to_frag<T>(s, c);
        if (c == 0) {
#pragma unroll
          for (int db = 0; db < C::NDB; ++db) cf[1][db] = load_frag_tr_p<T, C::CS, 2>(col, qb * 32 + 16, db * 32);
        }
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) acc[db] = mma32(cf[c][db], f, acc[db]);
        PFN_PIN_LDS_MFMA();
      }
    }
    if (!(ABL & 4)) __syncthreads();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  for (int t = 0; t < ntiles; t += 2) {
    tile(I0{}, t);
    if (t + 1 < ntiles) tile(I1{}, t + 1);
  }

  {
    T* out = dbase + (long)kc * rs + (MODE == 0 ? 2 * a.E : a.E) + hd * D;
    const float f = MODE == 0 ? 1.f : scale;
#pragma unroll
    for (int db = 0; db < C::NDB; ++db) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[db][r] * f;
      store_row_block<T>(out + db * 32, v, h, kvalid);
    }
  }
}

// =============================================================================================
// launchers
// =============================================================================================
template <typename T, int D> static int launch_fwd_t(const AttnArgs& a, hipStream_t s) {
  using C = AttnCfg<T, D>;
  const size_t lds = 2 * C::RIMG + 3 * C::CIMG;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<T, D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((attn_fwd_kernel<T, D>), dim3(((a.S + C::QBLK - 1) / C::QBLK) * a.H * a.B), dim3(C::NT), lds, s, a);
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}
template <typename T, int D, int MODE> static void launch_dkv_t(const AttnArgs& a, hipStream_t s) {
  using C = AttnCfg<T, D>;
  const size_t lds = 2 * (C::RIMG + C::CIMG + (MODE == 1 ? C::RIMG : 0) + 2 * C::KVB * 4);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<T, D, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, D, MODE>), dim3(((a.sep + C::QBLK - 1) / C::QBLK) * a.H * a.B), dim3(C::NT), lds, s, a);
}
template <typename T, int D> static int launch_bwd_t(const AttnArgs& a, hipStream_t s) {
  using C = AttnCfg<T, D>;
  {
    const long pairs = (long)a.B * a.S * a.H;
    int grid = (int)std::min<long>((pairs + 15) / 16, 4096);
    hipLaunchKernelGGL(attn_delta_kernel<T>, dim3(grid), dim3(256), 0, s, a, D);
  }
  {
    const size_t lds = 2 * (2 * C::RIMG + C::CIMG);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<T, D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, D>), dim3(((a.S + C::QBLK - 1) / C::QBLK) * a.H * a.B), dim3(C::NT), lds, s, a);
  }
  if (a.sep > 0) {
    launch_dkv_t<T, D, 0>(a, s);
    launch_dkv_t<T, D, 1>(a, s);
  }
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}

static int check_attn(const AttnArgs& a, int precision) {
  if (a.B <= 0 || a.S <= 0 || a.H <= 0 || a.E % a.H) return PFN_ERR_ARGUMENT;
  if (a.sep < 0 || a.sep > a.S) return PFN_ERR_ARGUMENT;
  const int es = precision == PFN_PREC_BF16 ? 2 : 4;
  if ((a.E * es) % 16) return PFN_ERR_ALIGNMENT;
  return PFN_OK;
}

#define PFN_ATTN_DISPATCH(FN)                                                             \
  const int D = a.E / a.H;                                                                 \
  if (precision == PFN_PREC_BF16) {                                                        \
    switch (D) {                                                                           \
      case 32: return FN<bf16, 32>(a, s);                                                  \
      case 64: return FN<bf16, 64>(a, s);                                                  \
      case 128: return FN<bf16, 128>(a, s);                                                \
      case 256: return FN<bf16, 256>(a, s);                                                \
      default: return PFN_ERR_UNSUPPORTED;                                                 \
    }                                                                                      \
  } else {                                                                                 \
    switch (D) {                                                                           \
      case 32: return FN<float, 32>(a, s);                                                 \
      case 64: return FN<float, 64>(a, s);                                                 \
      case 128: return FN<float, 128>(a, s);                                               \
      default: return PFN_ERR_UNSUPPORTED;                                                 \
    }                                                                                      \
  }

int launch_attn_fwd(const AttnArgs& a, int precision, hipStream_t s) {
  int rc = check_attn(a, precision);
  if (rc != PFN_OK) return rc;
  PFN_ATTN_DISPATCH(launch_fwd_t)
}
int launch_attn_bwd(const AttnArgs& a, int precision, hipStream_t s) {
  int rc = check_attn(a, precision);
  if (rc != PFN_OK) return rc;
  PFN_ATTN_DISPATCH(launch_bwd_t)
}

}  // namespace pfn
