// HBM-bound row-wise / element-wise kernels of the PFN stack (gfx950): parameter shadow casts,
// x/y embedding (reference transformer.py:66-74), LayerNorm forward/backward (the norm1/norm2 of
// nn.TransformerEncoderLayer, torch nn/modules/transformer.py:952-957), bias-gradient column
// sums, and the train/test row gather of `output[single_eval_pos:]` (transformer.py:91).
// All of them move 16 bytes per lane per access; none is worth an MFMA.
#include <algorithm>
#include "pfn_device.h"
#include "pfn_kernels.h"

namespace pfn {

template <typename T> PFN_DEV void st4(T* p, f32x4 x) {
  if constexpr (sizeof(T) == 2) {
    X4<T> v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (T)x[e];
    *reinterpret_cast<X4<T>*>(p) = v;
  } else *reinterpret_cast<f32x4*>(p) = x;
}
template <typename T> PFN_DEV f32x4 ld4(const T* p) {
  f32x4 r;
  if constexpr (sizeof(T) == 2) {
    X4<T> v = *reinterpret_cast<const X4<T>*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (float)v[e];
  } else r = *reinterpret_cast<const f32x4*>(p);
  return r;
}

static int grid_for(long work_items, int per_block, int cap = 4096) {
  long g = (work_items + per_block - 1) / per_block;
  return (int)std::max<long>(1, std::min<long>(g, cap));
}
#define PFN_LAUNCH_OK() (hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH)

// ---------------------------------------------------------------------------------------------
// f32 -> T casts
// ---------------------------------------------------------------------------------------------
template <typename T> __global__ __launch_bounds__(256) void cast_kernel(const float* src, T* dst, long n4) {
  operand_store_mode<T>();
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
    st4<T>(dst + 4 * i, *reinterpret_cast<const f32x4*>(src + 4 * i));
}
int launch_cast_params(const float* src, void* dst, long n, int precision, hipStream_t s) {
  if (n % 4) return PFN_ERR_ALIGNMENT;
  const long n4 = n / 4;
  if (n4 == 0) return PFN_OK;
  PFN_DISPATCH_OP(precision, hipLaunchKernelGGL(cast_kernel<T>, dim3(grid_for(n4, 256)), dim3(256), 0, s, src, (T*)dst, n4));
  return PFN_LAUNCH_OK();
}

template <typename T> __global__ __launch_bounds__(256) void cast_rows_kernel(const float* src, long ld_src, T* dst, long ld_dst, long R, int C, const float* scale_amax) {
  operand_store_mode<T>();
  const long total = R * ld_dst;
  const float sc = loss_scale_up(scale_amax);      // fp16 backward: the gradient enters the chain times 2^k (pfn_device.h)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / ld_dst; const int c = (int)(i % ld_dst);
    dst[i] = (T)(c < C ? src[r * ld_src + c] * sc : 0.f);
  }
}
int launch_cast_rows(const float* src, long ld_src, void* dst, long ld_dst, long R, int C, int precision, hipStream_t s, const float* scale_amax) {
  const long total = R * ld_dst;
  if (total == 0) return PFN_OK;
  PFN_DISPATCH_OP(precision, hipLaunchKernelGGL(cast_rows_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0, s, src, ld_src, (T*)dst, ld_dst, R, C, scale_amax));
  return PFN_LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------
// fp16 backward: amax = max |x| over the incoming gradient (non-negative floats order like their bit patterns: one integer atomic per workgroup), and the
// scaled f32 copy for the entry that has no cast in front of the chain (no decoder: the gradient of the test rows goes straight into the LayerNorm backward)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const float* x, long n, unsigned* amax_bits, float bias) {
  __shared__ float red[4];
  float m = 0.f;
  const long n4 = n / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));      // (fmaxf drops a NaN operand: a NaN gradient stays a NaN downstream, the scale stays sane)
  }
  if (blockIdx.x == 0) for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(x[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (m > 0.f) atomicMax(amax_bits, __builtin_bit_cast(unsigned, m * bias));      // bias = 2^(6 - target): see loss_scale_exp (pfn_device.h)
  }
}
static int g_loss_scale_target = 2;      // log2 of where max|dlogits| lands after scaling (PFN_TUNE_LOSS_SCALE_TARGET; pfn_device.h)
int set_loss_scale_target(int t) {
  if (t < -8 || t > 12) return PFN_ERR_ARGUMENT;
  g_loss_scale_target = t;
  return PFN_OK;
}
int launch_absmax(const float* x, long n, float* amax, hipStream_t s) {
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return PFN_ERR_ALIGNMENT;
  if (hipMemsetAsync(amax, 0, sizeof(float), s) != hipSuccess) return PFN_ERR_LAUNCH;
  if (n <= 0) return PFN_OK;
  hipLaunchKernelGGL(absmax_kernel, dim3(grid_for(n / 4 + 1, 256 * 8, 1024)), dim3(256), 0, s, x, n, reinterpret_cast<unsigned*>(amax),
                     ldexpf(1.f, LOSS_SCALE_TARGET_LOG2 - g_loss_scale_target));
  return PFN_LAUNCH_OK();
}
__global__ __launch_bounds__(256) void scale_copy_kernel(const float* src, float* dst, long n, const float* scale_amax) {
  const float sc = loss_scale_up(scale_amax);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i] * sc;
}
int launch_scale_copy(const float* src, float* dst, long n, const float* scale_amax, hipStream_t s) {
  if (n <= 0) return PFN_OK;
  hipLaunchKernelGGL(scale_copy_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, s, src, dst, n, scale_amax);
  return PFN_LAUNCH_OK();
}

// dst[c, r] = (T) src[r, c], dst row stride ld_dst >= rows (padding columns zeroed by the caller's
// memset-free contract: they are written here as zero)   (32x32 LDS tile transpose)
template <typename T> __global__ __launch_bounds__(256) void transpose_cast_kernel(const float* src, T* dst, int rows, int cols, long ld_dst) {
  operand_store_mode<T>();
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (c < cols && r < ld_dst) dst[(long)c * ld_dst + r] = (T)tile[tx][j];  // r >= rows writes the zero padding
  }
}
// every transposed operand copy of a model in ONE launch (pfn_prepare_params runs after each optimizer step: 26 launches of a few microseconds each before)
template <typename T> __global__ __launch_bounds__(256) void transpose_cast_group_kernel(const float* src, T* dst, TransposeGroup g) {
  operand_store_mode<T>();
  __shared__ float tile[32][33];
  int e = 0;
  while (e + 1 < g.n && (int)blockIdx.x >= g.first_block[e + 1]) ++e;
  const int rows = g.rows[e], cols = g.cols[e];
  const long ld_dst = g.ld_dst[e];
  const int bx = (cols + 31) / 32, blk = blockIdx.x - g.first_block[e];
  const int c0 = (blk % bx) * 32, r0 = (blk / bx) * 32;
  const float* sp = src + g.src_off[e];
  T* dp = dst + g.dst_off[e];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < rows && c < cols) ? sp[(long)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (c < cols && r < ld_dst) dp[(long)c * ld_dst + r] = (T)tile[tx][j];  // r >= rows writes the zero padding
  }
}
void transpose_group_add(TransposeGroup& g, long src_off, long dst_off, int rows, int cols, int ld_dst) {
  const int e = g.n++;
  g.src_off[e] = src_off; g.dst_off[e] = dst_off; g.rows[e] = rows; g.cols[e] = cols; g.ld_dst[e] = ld_dst;
  g.first_block[e] = g.blocks;
  g.blocks += ((cols + 31) / 32) * ((ld_dst + 31) / 32);
}
int launch_transpose_cast_group(const float* src, void* dst, const TransposeGroup& g, int precision, hipStream_t s) {
  if (g.n == 0) return PFN_OK;
  PFN_DISPATCH_OP(precision, hipLaunchKernelGGL(transpose_cast_group_kernel<T>, dim3(g.blocks), dim3(256), 0, s, src, (T*)dst, g));
  return PFN_LAUNCH_OK();
}
int launch_transpose_cast(const float* src, void* dst, int rows, int cols, long ld_dst, int precision, hipStream_t s) {
  dim3 grid((cols + 31) / 32, (unsigned)((ld_dst + 31) / 32));
  PFN_DISPATCH_OP(precision, hipLaunchKernelGGL(transpose_cast_kernel<T>, grid, dim3(256), 0, s, src, (T*)dst, rows, cols, ld_dst));
  return PFN_LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------
// Key centring (pfn_kernels.h launch_key_shift): kshift[b, n] = sum_e w_k[n, e] * xbar[b, e], xbar = the mean of KS_SAMPLES evenly spaced rows of dataset b's
// layer input (train rows only).  One workgroup per (dataset, 64 outputs): the sample mean (64 rows x E, 16-byte loads, f32 sums through LDS), then a wave per output row of
// w_k at a time (16 bytes per lane: a whole row of E <= 512 per instruction).  Reads 2 x 64 KB per workgroup at E = 512: a few microseconds per layer.
// ---------------------------------------------------------------------------------------------
constexpr int KS_SAMPLES = 64;
template <typename T> __global__ __launch_bounds__(256) void key_shift_kernel(const T* x, const T* wk, float* kshift, int S, int E, int sep_all, const int* sep_of) {
  __shared__ float part[2048];                       // [R row lanes][E] partial sums, then xbar in part[0 .. E)
  const int b = blockIdx.y, n0 = blockIdx.x * 64;
  // the sample is taken from the TRAIN rows [0, sep) only: they are the keys every query sees, and a test row must not reach any other row's output -- not even
  // through a rounding-level shift (the reference's mask, transformer.py:34-41; tests pin the other test rows bit-identical when one changes)
  const int sep = sep_of ? sep_of[b] : sep_all;
  if (sep <= 0) {      // no train row: the only key of a query is its own
    for (int o = threadIdx.x; o < 64 && n0 + o < E; o += 256) kshift[(long)b * E + n0 + o] = 0.f;
    return;
  }
  const int ns = sep < KS_SAMPLES ? sep : KS_SAMPLES, stride = sep / ns;
  const int G = E / 8, R = G >= 256 ? 1 : 256 / G;
  const T* xb = x + (long)b * S * E;
  // (every load of a batch is issued before the first is used: written as "load, add" per row the loop is a chain of memory round trips -- 17 us for this
  // kernel in its first form, gpurun_out/r06c5)
  for (int idx = threadIdx.x; idx < G * R; idx += 256) {
    const int cg = idx % G, rl = idx / G;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r0 = rl; r0 < ns; r0 += 8 * R) {
      X8<T> v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const X8<T>*>(xb + (long)min(r0 + u * R, ns - 1) * stride * E + cg * 8);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r0 + u * R < ns) {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += (float)v[u][e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[rl * E + cg * 8 + e] = acc[e];
  }
  __syncthreads();
  const float inv = 1.f / (float)ns;
  for (int e = threadIdx.x; e < E; e += 256) {
    float s = 0.f;
    for (int rl = 0; rl < R; ++rl) s += part[rl * E + e];
    part[e] = s * inv;                               // (row lane 0's slot: every thread reads its own column of the other lanes before writing)
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int pass = 0; pass * 512 < E; ++pass) {       // (one pass at E <= 512; every lane walks every pass: the wave-wide sum below wants all 64)
    const int e0 = pass * 512 + lane * 8;
    const bool live = e0 < E;
    const int ec = live ? e0 : 0;
    X8<T> w[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) w[o] = *reinterpret_cast<const X8<T>*>(wk + (long)min(n0 + wave * 16 + o, E - 1) * E + ec);
    float xv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) xv[e] = live ? part[ec + e] : 0.f;
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)w[o][e] * xv[e];
      s = wave_sum(s);
      const int n = n0 + wave * 16 + o;
      if (lane == 0 && n < E) {
        if (pass == 0) kshift[(long)b * E + n] = s;
        else kshift[(long)b * E + n] += s;           // (E > 512: the later passes add their part, same lane, program order)
      }
    }
  }
}
int launch_key_shift(const void* x_t, const void* w_k_t, float* kshift, int B, int S, int E, int sep, const int* sep_of, int precision, hipStream_t s) {
  if (!prec_is16(precision) || E % 8 || E > 2048 || B < 1 || S < 1) return PFN_ERR_UNSUPPORTED;
  const dim3 grid((E + 63) / 64, B);
  if (precision == PFN_PREC_FP16) hipLaunchKernelGGL(key_shift_kernel<f16>, grid, dim3(256), 0, s, (const f16*)x_t, (const f16*)w_k_t, kshift, S, E, sep, sep_of);
  else hipLaunchKernelGGL(key_shift_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)x_t, (const bf16*)w_k_t, kshift, S, E, sep, sep_of);
  return PFN_LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------
// embedding: src[b,s,:] = Wx x[s,b,:] + bx  (+ Wy y[s,b] + by  when s < sep)
// ---------------------------------------------------------------------------------------------
constexpr int EMB_TOK = 16;
template <typename T> __global__ __launch_bounds__(256) void embed_fwd_kernel(EmbedArgs a) {
  operand_store_mode<T>();
  extern __shared__ float xs[];  // [EMB_TOK][nf + 1]  (last column: y or 0, plus flag in sign-free form)
  const long ntok = (long)a.B * a.S;
  const long t0 = (long)blockIdx.x * EMB_TOK;
  const int nfp = a.nf + 2;
  for (int i = threadIdx.x; i < EMB_TOK * nfp; i += 256) {
    const int tk = i / nfp, f = i % nfp;
    const long tok = t0 + tk;
    float v = 0.f;
    if (tok < ntok) {
      const long b = tok / a.S, sidx = tok % a.S;
      const int sep = a.sep_of ? a.sep_of[b] : a.sep;
      if (f < a.nf) v = a.x[sidx * a.x_st + b * a.x_sb + f];
      else if (f == a.nf) v = (sidx < sep) ? a.y[sidx * a.y_st + b * a.y_sb] : 0.f;
      else v = (sidx < sep) ? 1.f : 0.f;
    }
    xs[i] = v;
  }
  __syncthreads();
  T* out_t = reinterpret_cast<T*>(a.out_t);
  if (a.xaug_t) {   // the same values in operand precision, xaug_ld columns per token (zero padded)
    T* xa = reinterpret_cast<T*>(a.xaug_t);
    const int aug = a.xaug_ld;
    for (int i = threadIdx.x; i < EMB_TOK * aug; i += 256) {
      const int tk = i / aug, f = i % aug;
      if (t0 + tk < ntok) xa[(t0 + tk) * aug + f] = (T)(f < nfp ? xs[tk * nfp + f] : 0.f);
    }
  }
  for (int e = threadIdx.x; e < a.E; e += 256) {
    const float be = a.bx[e], wye = a.wy[e], bye = a.by[e];
    // explicit fused multiply-adds: left to the compiler, the 16 unrolled tokens were contracted differently (packed multiply + add for some pairs, fma for
    // others), so a token's value depended on its position modulo 16 in the last bit -- and with it a dataset's logits on its index in the batch (found in
    // round 5 by the ragged-batch test: bit-equal to the separate forwards only when the row offset was a multiple of 16)
    float acc[EMB_TOK];
#pragma unroll
    for (int tk = 0; tk < EMB_TOK; ++tk) acc[tk] = __builtin_fmaf(xs[tk * nfp + a.nf + 1], bye, __builtin_fmaf(xs[tk * nfp + a.nf], wye, be));
    for (int f = 0; f < a.nf; ++f) {
      const float w = a.wx[(long)e * a.nf + f];
#pragma unroll
      for (int tk = 0; tk < EMB_TOK; ++tk) acc[tk] = __builtin_fmaf(w, xs[tk * nfp + f], acc[tk]);
    }
#pragma unroll
    for (int tk = 0; tk < EMB_TOK; ++tk) {
      const long tok = t0 + tk;
      if (tok < ntok) {
        a.out_f32[tok * a.E + e] = acc[tk];
        out_t[tok * a.E + e] = (T)acc[tk];
      }
    }
  }
}
int launch_embed_fwd(const EmbedArgs& a, int precision, hipStream_t s) {
  const long ntok = (long)a.B * a.S;
  const int grid = (int)((ntok + EMB_TOK - 1) / EMB_TOK);
  const size_t lds = EMB_TOK * (a.nf + 2) * sizeof(float);
  PFN_DISPATCH_OP(precision, hipLaunchKernelGGL(embed_fwd_kernel<T>, dim3(grid), dim3(256), lds, s, a));
  return PFN_LAUNCH_OK();
}

// Workgroup = 64 embedding columns x one slice of the tokens; thread = (column, one of 4 token lanes).  d(src) is
// read exactly once (a wave reads 256 contiguous bytes per token), the token's features come from LDS as broadcast
// reads, all nf + 2 weight-gradient columns of a thread stay in registers; the 4 token lanes are reduced through LDS
// and every gradient element receives one atomic per workgroup -- gridDim.x (<= 32) of them in total instead of
// one per 128-token chunk.
constexpr int EMBB_TOK = 128;   // tokens staged per pass
constexpr int EMBB_MAXF = 32;   // nf + 2 rounded up to 8 must fit (wider encoders take the chunked path below)
template <int NF8>
__global__ __launch_bounds__(256) void embed_bwd_kernel(EmbedBwdArgs a) {
  extern __shared__ float xs[];  // [EMBB_TOK][NF8] : x features, y (masked), train flag, zero pad to 8;  later [4][64][NF8 + 1] partials
  const long ntok = (long)a.B * a.S;
  const int col = threadIdx.x & 63, tl = threadIdx.x >> 6;
  const int e = blockIdx.y * 64 + col;
  const long per = (ntok + gridDim.x - 1) / gridDim.x;
  const long tbeg = (long)blockIdx.x * per, tend = std::min(ntok, tbeg + per);
  float acc[NF8];
#pragma unroll
  for (int j = 0; j < NF8; ++j) acc[j] = 0.f;
  float db = 0.f;
  for (long t0 = tbeg; t0 < tend; t0 += EMBB_TOK) {
    __syncthreads();
    for (int i = threadIdx.x; i < EMBB_TOK * NF8; i += 256) {
      const int tk = i / NF8, f = i % NF8;
      const long tok = t0 + tk;
      float v = 0.f;
      if (tok < tend) {
        const long b = tok / a.S, sidx = tok % a.S;
        const int sep = a.sep_of ? a.sep_of[b] : a.sep;
        if (f < a.nf) v = a.x[sidx * a.x_st + b * a.x_sb + f];
        else if (f == a.nf) v = (sidx < sep) ? a.y[sidx * a.y_st + b * a.y_sb] : 0.f;
        else if (f == a.nf + 1) v = (sidx < sep) ? 1.f : 0.f;
      }
      xs[i] = v;
    }
    __syncthreads();
    if (e >= a.E) continue;
    const int ntk = (int)std::min<long>(EMBB_TOK, tend - t0);
    const float* dcol = a.dsrc + t0 * a.E + e;
#pragma unroll 8
    for (int tk = tl; tk < ntk; tk += 4) {   // independent loads in flight per thread
      const float d = dcol[(long)tk * a.E];
      db += d;
#pragma unroll
      for (int j = 0; j < NF8; j += 4) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + tk * NF8 + j);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j + q] += d * xv[q];
      }
    }
  }
  __syncthreads();
  float* red = xs;   // [4 token lanes][64 columns][NF8 + 1]
#pragma unroll
  for (int j = 0; j < NF8; ++j) red[(tl * 64 + col) * (NF8 + 1) + j] = acc[j];
  red[(tl * 64 + col) * (NF8 + 1) + NF8] = db;
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * (NF8 + 1); i += 256) {
    const int c = i / (NF8 + 1), f = i % (NF8 + 1);
    const int ee = blockIdx.y * 64 + c;
    if (ee >= a.E) continue;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) t += red[(q * 64 + c) * (NF8 + 1) + f];
    t *= loss_scale_down(a.scale_amax);
    if (f < a.nf) unsafeAtomicAdd(a.dwx + (long)ee * a.nf + f, t);
    else if (f == a.nf) unsafeAtomicAdd(a.dwy + ee, t);
    else if (f == a.nf + 1) unsafeAtomicAdd(a.dby + ee, t);
    else if (f == NF8) unsafeAtomicAdd(a.dbx + ee, t);
  }
}
// wide encoders (nf + 2 > 32): features in chunks of 8, d(src) re-read per chunk
__global__ __launch_bounds__(256) void embed_bwd_wide_kernel(EmbedBwdArgs a) {
  extern __shared__ float xs[];
  const long ntok = (long)a.B * a.S;
  const long t0 = (long)blockIdx.x * 128;
  const int nf8 = (a.nf + 2 + 7) / 8 * 8;
  for (int i = threadIdx.x; i < 128 * nf8; i += 256) {
    const int tk = i / nf8, f = i % nf8;
    const long tok = t0 + tk;
    float v = 0.f;
    if (tok < ntok) {
      const long b = tok / a.S, sidx = tok % a.S;
      const int sep = a.sep_of ? a.sep_of[b] : a.sep;
      if (f < a.nf) v = a.x[sidx * a.x_st + b * a.x_sb + f];
      else if (f == a.nf) v = (sidx < sep) ? a.y[sidx * a.y_st + b * a.y_sb] : 0.f;
      else if (f == a.nf + 1) v = (sidx < sep) ? 1.f : 0.f;
    }
    xs[i] = v;
  }
  __syncthreads();
  const int ntk = (int)std::min<long>(128, ntok - t0);
  const float osc = loss_scale_down(a.scale_amax);
  for (int e = threadIdx.x; e < a.E; e += 256) {
    float db = 0.f;
    for (int f0 = 0; f0 < nf8; f0 += 8) {
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int tk = 0; tk < ntk; ++tk) {
        const float d = a.dsrc[(t0 + tk) * a.E + e];
        if (f0 == 0) db += d;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += d * xs[tk * nf8 + f0 + j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int f = f0 + j;
        if (f < a.nf) unsafeAtomicAdd(a.dwx + (long)e * a.nf + f, acc[j] * osc);
        else if (f == a.nf) unsafeAtomicAdd(a.dwy + e, acc[j] * osc);
        else if (f == a.nf + 1) unsafeAtomicAdd(a.dby + e, acc[j] * osc);
      }
    }
    unsafeAtomicAdd(a.dbx + e, db * osc);
  }
}
__global__ __launch_bounds__(256) void embed_grad_scatter_kernel(const float* acc, float* dwx, float* dwy, float* dby, int E, int nf) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= E * (nf + 2)) return;
  const int e = i / (nf + 2), f = i % (nf + 2);
  const float v = acc[e * emb_aug_width(nf) + f];      // (already unscaled: the GEMM that filled acc took the loss scale out)
  if (f < nf) unsafeAtomicAdd(dwx + (long)e * nf + f, v);
  else if (f == nf) unsafeAtomicAdd(dwy + e, v);
  else unsafeAtomicAdd(dby + e, v);
}
int launch_embed_grad_scatter(const float* acc, float* dwx, float* dwy, float* dby, int E, int nf, hipStream_t s) {
  hipLaunchKernelGGL(embed_grad_scatter_kernel, dim3((E * (nf + 2) + 255) / 256), dim3(256), 0, s, acc, dwx, dwy, dby, E, nf);
  return PFN_LAUNCH_OK();
}
int launch_embed_bwd(const EmbedBwdArgs& a, hipStream_t s) {
  const long ntok = (long)a.B * a.S;
  const int nf8 = (a.nf + 2 + 7) / 8 * 8;
  if (nf8 <= EMBB_MAXF) {
    const int ey = (a.E + 63) / 64;
#ifndef PFN_EMBB_WGS
#define PFN_EMBB_WGS 512
#endif
    // (single_block: one workgroup per column block walks every token -- each gradient element then has one writer: PFN_SCHED_DETERMINISTIC)
    const dim3 grid(a.single_block ? 1u : (unsigned)std::max<long>(1, std::min<long>((ntok + EMBB_TOK - 1) / EMBB_TOK, std::max(1, PFN_EMBB_WGS / ey))), ey);
    const size_t lds = std::max((size_t)EMBB_TOK * nf8, (size_t)4 * 64 * (nf8 + 1)) * sizeof(float);
    switch (nf8) {
      case 8: hipLaunchKernelGGL(embed_bwd_kernel<8>, grid, dim3(256), lds, s, a); break;
      case 16: hipLaunchKernelGGL(embed_bwd_kernel<16>, grid, dim3(256), lds, s, a); break;
      case 24: hipLaunchKernelGGL(embed_bwd_kernel<24>, grid, dim3(256), lds, s, a); break;
      default: hipLaunchKernelGGL(embed_bwd_kernel<32>, grid, dim3(256), lds, s, a); break;
    }
    return PFN_LAUNCH_OK();
  }
  if (a.single_block) return PFN_ERR_UNSUPPORTED;      // the wide-encoder kernel splits the tokens over workgroups (atomics): no deterministic form
  const int grid = (int)((ntok + 127) / 128);
  const size_t lds = 128 * (size_t)nf8 * sizeof(float);
  if (lds > 64 * 1024) return PFN_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(embed_bwd_wide_kernel, dim3(grid), dim3(256), lds, s, a);
  return PFN_LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------
// layout copies between the reference [S,B,E] and the internal [B,S,E]
// ---------------------------------------------------------------------------------------------
template <typename T> __global__ __launch_bounds__(256) void sbe_to_bse_kernel(const float* src, float* o32, T* ot, int S, int B, int E) {
  operand_store_mode<T>();
  const long n4 = (long)S * B * E / 4;
  const int e4 = E / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long tok = i / e4; const int c = (int)(i % e4) * 4;
    const long b = tok / S, sidx = tok % S;
    f32x4 v = *reinterpret_cast<const f32x4*>(src + (sidx * B + b) * E + c);
    *reinterpret_cast<f32x4*>(o32 + tok * E + c) = v;
    st4<T>(ot + tok * E + c, v);
  }
}
int launch_sbe_to_bse(const float* src, float* o32, void* ot, int S, int B, int E, int precision, hipStream_t s) {
  if (E % 4) return PFN_ERR_ALIGNMENT;
  const long n4 = (long)S * B * E / 4;
  PFN_DISPATCH_OP(precision, hipLaunchKernelGGL(sbe_to_bse_kernel<T>, dim3(grid_for(n4, 256)), dim3(256), 0, s, src, o32, (T*)ot, S, B, E));
  return PFN_LAUNCH_OK();
}
__global__ __launch_bounds__(256) void bse_to_sbe_kernel(const float* src, float* dst, int S, int B, int E, const float* scale_amax) {
  const long n4 = (long)S * B * E / 4;
  const int e4 = E / 4;
  const float osc = loss_scale_down(scale_amax);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long tok = i / e4; const int c = (int)(i % e4) * 4;
    const long b = tok / S, sidx = tok % S;
    *reinterpret_cast<f32x4*>(dst + (sidx * B + b) * E + c) = *reinterpret_cast<const f32x4*>(src + tok * E + c) * osc;
  }
}
int launch_bse_to_sbe(const float* src, float* dst, int S, int B, int E, hipStream_t s, const float* scale_amax) {
  if (E % 4) return PFN_ERR_ALIGNMENT;
  const long n4 = (long)S * B * E / 4;
  hipLaunchKernelGGL(bse_to_sbe_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, s, src, dst, S, B, E, scale_amax);
  return PFN_LAUNCH_OK();
}

// gather / scatter of the test rows (s >= sep) into the decoder's compact [(S-sep)*B, E] matrix
template <typename T> __global__ __launch_bounds__(256) void gather_test_kernel(const float* src, T* dst, int S, int B, int E, int sep) {
  operand_store_mode<T>();
  const int e4 = E / 4;
  const long n4 = (long)(S - sep) * B * e4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long row = i / e4; const int c = (int)(i % e4) * 4;
    const long sidx = sep + row / B, b = row % B;
    st4<T>(dst + row * E + c, *reinterpret_cast<const f32x4*>(src + (b * S + sidx) * E + c));
  }
}
// ---- element-wise dropout (dropout > 0 only) ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void dropout_add_kernel(float* y, const float* resid, long n4, int cols4, unsigned seed, unsigned thr, float scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const unsigned r = (unsigned)(i / cols4), c = (unsigned)(i % cols4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4);
    const f32x4 x = *reinterpret_cast<const f32x4*>(resid + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = x[e] + (dropout_keep(seed, r, c + e, thr) ? v[e] * scale : 0.f);
    *reinterpret_cast<f32x4*>(y + i * 4) = v;
  }
}
int launch_dropout_add(float* y, const float* resid, long rows, int cols, unsigned site_seed, float p, hipStream_t s) {
  if (cols % 4) return PFN_ERR_UNSUPPORTED;
  const long n4 = rows * cols / 4;
  if (n4 == 0) return PFN_OK;
  hipLaunchKernelGGL(dropout_add_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, s, y, resid, n4, cols / 4, site_seed, dropout_threshold(p), 1.f / (1.f - p));
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}
template <typename T> __global__ __launch_bounds__(256) void dropout_scale_kernel(const T* src, T* dst, const T* src2, T* dst2, long n4, int cols4, unsigned seed, unsigned thr, float scale) {
  operand_store_mode<T>();
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const unsigned r = (unsigned)(i / cols4), c = (unsigned)(i % cols4) * 4;
    f32x4 v = ld4<T>(src + i * 4), w = {0.f, 0.f, 0.f, 0.f};
    if (src2) w = ld4<T>(src2 + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float k = dropout_keep(seed, r, c + e, thr) ? scale : 0.f;
      v[e] *= k; w[e] *= k;
    }
    st4<T>(dst + i * 4, v);
    if (src2) st4<T>(dst2 + i * 4, w);
  }
}
int launch_dropout_scale(const void* src, void* dst, const void* src2, void* dst2, long rows, int cols, unsigned site_seed, float p, int precision, hipStream_t s) {
  if (cols % 4) return PFN_ERR_UNSUPPORTED;
  const long n4 = rows * cols / 4;
  if (n4 == 0) return PFN_OK;
  const unsigned thr = dropout_threshold(p);
  const float scale = 1.f / (1.f - p);
  PFN_DISPATCH_OP(precision, hipLaunchKernelGGL(dropout_scale_kernel<T>, dim3(grid_for(n4, 256)), dim3(256), 0, s, (const T*)src, (T*)dst, (const T*)src2, (T*)dst2, n4, cols / 4, site_seed, thr, scale));
  return hipGetLastError() == hipSuccess ? PFN_OK : PFN_ERR_LAUNCH;
}

int launch_gather_test_rows(const float* src, void* dst, int S, int B, int E, int sep, int precision, hipStream_t s) {
  if (E % 4) return PFN_ERR_ALIGNMENT;
  const long n4 = (long)(S - sep) * B * E / 4;
  if (n4 == 0) return PFN_OK;
  PFN_DISPATCH_OP(precision, hipLaunchKernelGGL(gather_test_kernel<T>, dim3(grid_for(n4, 256)), dim3(256), 0, s, src, (T*)dst, S, B, E, sep));
  return PFN_LAUNCH_OK();
}
template <typename T> __global__ __launch_bounds__(256) void scatter_test_kernel(const float* src, T* dst, int S, int B, int E, int sep) {
  operand_store_mode<T>();
  const int e4 = E / 4;
  const long n4 = (long)S * B * e4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long tok = i / e4; const int c = (int)(i % e4) * 4;
    const long b = tok / S, sidx = tok % S;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (sidx >= sep) v = *reinterpret_cast<const f32x4*>(src + ((sidx - sep) * B + b) * E + c);
    st4<T>(dst + tok * E + c, v);
  }
}
int launch_scatter_test_rows(const float* src, void* dst, int S, int B, int E, int sep, int precision, hipStream_t s) {
  if (E % 4) return PFN_ERR_ALIGNMENT;
  const long n4 = (long)S * B * E / 4;
  PFN_DISPATCH_OP(precision, hipLaunchKernelGGL(scatter_test_kernel<T>, dim3(grid_for(n4, 256)), dim3(256), 0, s, src, (T*)dst, S, B, E, sep));
  return PFN_LAUNCH_OK();
}

// ragged batch (round 5: the micro-batches of one optimizer step in ONE launch set, each dataset with its own eval position): the decoder's compact rows are
// dataset-major, row_off[b] + (s - sep_of[b]).  Both kernels walk the [B, S] token order.
template <typename T> __global__ __launch_bounds__(256) void gather_test_ragged_kernel(const float* src, T* dst, int S, int B, int E, const int* sep_of, const long* row_off) {
  operand_store_mode<T>();
  const int e4 = E / 4;
  const long n4 = (long)S * B * e4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long tok = i / e4; const int c = (int)(i % e4) * 4;
    const long b = tok / S, sidx = tok % S;
    const int sep = sep_of[b];
    if (sidx >= sep) st4<T>(dst + (row_off[b] + sidx - sep) * E + c, *reinterpret_cast<const f32x4*>(src + tok * E + c));
  }
}
template <typename T> __global__ __launch_bounds__(256) void scatter_test_ragged_kernel(const float* src, T* dst, int S, int B, int E, const int* sep_of, const long* row_off) {
  operand_store_mode<T>();
  const int e4 = E / 4;
  const long n4 = (long)S * B * e4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long tok = i / e4; const int c = (int)(i % e4) * 4;
    const long b = tok / S, sidx = tok % S;
    const int sep = sep_of[b];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (sidx >= sep) v = *reinterpret_cast<const f32x4*>(src + (row_off[b] + sidx - sep) * E + c);
    st4<T>(dst + tok * E + c, v);
  }
}
int launch_gather_test_rows_ragged(const float* src, void* dst, int S, int B, int E, const int* sep_of, const long* row_off, int precision, hipStream_t s) {
  if (E % 4) return PFN_ERR_ALIGNMENT;
  const long n4 = (long)S * B * E / 4;
  PFN_DISPATCH_OP(precision, hipLaunchKernelGGL(gather_test_ragged_kernel<T>, dim3(grid_for(n4, 256)), dim3(256), 0, s, src, (T*)dst, S, B, E, sep_of, row_off));
  return PFN_LAUNCH_OK();
}
int launch_scatter_test_rows_ragged(const float* src, void* dst, int S, int B, int E, const int* sep_of, const long* row_off, int precision, hipStream_t s) {
  if (E % 4) return PFN_ERR_ALIGNMENT;
  const long n4 = (long)S * B * E / 4;
  PFN_DISPATCH_OP(precision, hipLaunchKernelGGL(scatter_test_ragged_kernel<T>, dim3(grid_for(n4, 256)), dim3(256), 0, s, src, (T*)dst, S, B, E, sep_of, row_off));
  return PFN_LAUNCH_OK();
}

// The top encoder layer runs on the test rows only (pfn_api.hip): rows of any operand move between the [B, S] token order and the
// compact [S - sep, B] order of the decoder's rows as bytes (W = 16 or 4 bytes per unit; a row is row_units units).
template <typename U> __global__ __launch_bounds__(256) void gather_rows_kernel(const U* src, U* dst, int S, int B, int row_units, int sep) {
  const long n = (long)(S - sep) * B * row_units;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long row = i / row_units; const int c = (int)(i % row_units);
    const long t = row / B, b = row % B;
    dst[i] = src[(b * S + sep + t) * row_units + c];
  }
}
int launch_gather_rows(const void* src_bs, void* dst_tb, int S, int B, long row_bytes, int sep, hipStream_t s) {
  if (row_bytes % 4) return PFN_ERR_ALIGNMENT;
  if ((long)(S - sep) * B * row_bytes == 0) return PFN_OK;
  if (row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(src_bs) | reinterpret_cast<uintptr_t>(dst_tb)) % 16 == 0) {
    const int ru = (int)(row_bytes / 16);
    hipLaunchKernelGGL(gather_rows_kernel<u32x4>, dim3(grid_for((long)(S - sep) * B * ru, 256)), dim3(256), 0, s, (const u32x4*)src_bs, (u32x4*)dst_tb, S, B, ru, sep);
  } else {
    const int ru = (int)(row_bytes / 4);
    hipLaunchKernelGGL(gather_rows_kernel<unsigned>, dim3(grid_for((long)(S - sep) * B * ru, 256)), dim3(256), 0, s, (const unsigned*)src_bs, (unsigned*)dst_tb, S, B, ru, sep);
  }
  return PFN_LAUNCH_OK();
}
// dst[b, s, :] = src[(s - sep) * B + b, :] for s >= sep, zeros for zero_from <= s < sep; rows below zero_from are not touched
template <typename U> __global__ __launch_bounds__(256) void scatter_rows_kernel(const U* src, U* dst, int S, int B, int row_units, int sep, int zero_from) {
  const long per_b = (long)(S - zero_from) * row_units, n = per_b * B;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / per_b, r = i % per_b;
    const long sidx = zero_from + r / row_units; const int c = (int)(r % row_units);
    U v = {};
    if (sidx >= sep) v = src[((sidx - sep) * B + b) * row_units + c];
    dst[(b * S + sidx) * row_units + c] = v;
  }
}
int launch_scatter_rows(const void* src_tb, void* dst_bs, int S, int B, long row_bytes, int sep, int zero_from, hipStream_t s) {
  if (row_bytes % 4) return PFN_ERR_ALIGNMENT;
  if (zero_from < 0 || zero_from > sep) return PFN_ERR_ARGUMENT;
  if ((long)(S - zero_from) * B * row_bytes == 0) return PFN_OK;
  if (row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(src_tb) | reinterpret_cast<uintptr_t>(dst_bs)) % 16 == 0) {
    const int ru = (int)(row_bytes / 16);
    hipLaunchKernelGGL(scatter_rows_kernel<u32x4>, dim3(grid_for((long)(S - zero_from) * B * ru, 256)), dim3(256), 0, s, (const u32x4*)src_tb, (u32x4*)dst_bs, S, B, ru, sep, zero_from);
  } else {
    const int ru = (int)(row_bytes / 4);
    hipLaunchKernelGGL(scatter_rows_kernel<unsigned>, dim3(grid_for((long)(S - zero_from) * B * ru, 256)), dim3(256), 0, s, (const unsigned*)src_tb, (unsigned*)dst_bs, S, B, ru, sep, zero_from);
  }
  return PFN_LAUNCH_OK();
}
// ---- the same three row moves for a ragged batch (per-dataset eval positions, compact rows dataset-major: row_off[b] + s - sep_of[b]); all walk the [B, S] order ----
template <typename U> __global__ __launch_bounds__(256) void gather_rows_ragged_kernel(const U* src, U* dst, int S, int B, int row_units, const int* sep_of, const long* row_off) {
  const long n = (long)S * B * row_units;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long tok = i / row_units; const int c = (int)(i % row_units);
    const long b = tok / S, sidx = tok % S;
    const int sep = sep_of[b];
    if (sidx >= sep) dst[(row_off[b] + sidx - sep) * row_units + c] = src[i];
  }
}
template <typename U> __global__ __launch_bounds__(256) void scatter_rows_ragged_kernel(const U* src, U* dst, int S, int B, int row_units, const int* sep_of, const long* row_off, int zero_from_block) {
  const long n = (long)S * B * row_units;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long tok = i / row_units; const int c = (int)(i % row_units);
    const long b = tok / S, sidx = tok % S;
    const int sep = sep_of[b];
    if (sidx < (zero_from_block ? sep / 256 * 256 : 0)) continue;      // rows below the dataset's first live query block are not touched
    U v = {};
    if (sidx >= sep) v = src[(row_off[b] + sidx - sep) * row_units + c];
    dst[i] = v;
  }
}
__global__ __launch_bounds__(256) void zero_row_prefix_ragged_kernel(u32x4* base, int S, int B, const int* sep_of, int row_units, int width_units) {
  const long n = (long)S * B * width_units;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long tok = i / width_units; const int c = (int)(i % width_units);
    const long b = tok / S, sidx = tok % S;
    if (sidx < sep_of[b] / 256 * 256) base[tok * row_units + c] = u32x4{0u, 0u, 0u, 0u};
  }
}
int launch_gather_rows_ragged(const void* src_bs, void* dst, int S, int B, long row_bytes, const int* sep_of, const long* row_off, hipStream_t s) {
  if (row_bytes % 4) return PFN_ERR_ALIGNMENT;
  if (row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(src_bs) | reinterpret_cast<uintptr_t>(dst)) % 16 == 0) {
    const int ru = (int)(row_bytes / 16);
    hipLaunchKernelGGL(gather_rows_ragged_kernel<u32x4>, dim3(grid_for((long)S * B * ru, 256)), dim3(256), 0, s, (const u32x4*)src_bs, (u32x4*)dst, S, B, ru, sep_of, row_off);
  } else {
    const int ru = (int)(row_bytes / 4);
    hipLaunchKernelGGL(gather_rows_ragged_kernel<unsigned>, dim3(grid_for((long)S * B * ru, 256)), dim3(256), 0, s, (const unsigned*)src_bs, (unsigned*)dst, S, B, ru, sep_of, row_off);
  }
  return PFN_LAUNCH_OK();
}
int launch_scatter_rows_ragged(const void* src, void* dst_bs, int S, int B, long row_bytes, const int* sep_of, const long* row_off, int zero_from_block, hipStream_t s) {
  if (row_bytes % 4) return PFN_ERR_ALIGNMENT;
  if (row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst_bs)) % 16 == 0) {
    const int ru = (int)(row_bytes / 16);
    hipLaunchKernelGGL(scatter_rows_ragged_kernel<u32x4>, dim3(grid_for((long)S * B * ru, 256)), dim3(256), 0, s, (const u32x4*)src, (u32x4*)dst_bs, S, B, ru, sep_of, row_off, zero_from_block);
  } else {
    const int ru = (int)(row_bytes / 4);
    hipLaunchKernelGGL(scatter_rows_ragged_kernel<unsigned>, dim3(grid_for((long)S * B * ru, 256)), dim3(256), 0, s, (const unsigned*)src, (unsigned*)dst_bs, S, B, ru, sep_of, row_off, zero_from_block);
  }
  return PFN_LAUNCH_OK();
}
int launch_zero_row_prefix_ragged(void* base, int S, int B, const int* sep_of, long row_bytes, long width_bytes, hipStream_t s) {
  if (row_bytes % 16 || width_bytes % 16 || reinterpret_cast<uintptr_t>(base) % 16) return PFN_ERR_ALIGNMENT;
  hipLaunchKernelGGL(zero_row_prefix_ragged_kernel, dim3(grid_for((long)S * B * (width_bytes / 16), 256)), dim3(256), 0, s, (u32x4*)base, S, B, sep_of,
                     (int)(row_bytes / 16), (int)(width_bytes / 16));
  return PFN_LAUNCH_OK();
}
// base[b, s, 0 : width] = 0 for s < nrows (rows of row_bytes bytes; width_bytes, row_bytes multiples of 16)
__global__ __launch_bounds__(256) void zero_row_prefix_kernel(u32x4* base, int S, int B, int nrows, int row_units, int width_units) {
  const long per_b = (long)nrows * width_units, n = per_b * B;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / per_b, r = i % per_b;
    base[(b * S + r / width_units) * row_units + r % width_units] = u32x4{0u, 0u, 0u, 0u};
  }
}
int launch_zero_row_prefix(void* base, int S, int B, int nrows, long row_bytes, long width_bytes, hipStream_t s) {
  if (row_bytes % 16 || width_bytes % 16 || reinterpret_cast<uintptr_t>(base) % 16) return PFN_ERR_ALIGNMENT;
  if ((long)nrows * B * width_bytes == 0) return PFN_OK;
  hipLaunchKernelGGL(zero_row_prefix_kernel, dim3(grid_for((long)nrows * B * (width_bytes / 16), 256)), dim3(256), 0, s, (u32x4*)base, S, B, nrows,
                     (int)(row_bytes / 16), (int)(width_bytes / 16));
  return PFN_LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, the row lives in registers (NV float4 per lane, E <= 256*NV)
// ---------------------------------------------------------------------------------------------
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const void* x_any, const float* gamma, const float* beta, float* y32, T* yt,
                                                            float* mean_o, float* rstd_o, long rows, int E, float eps, int x_is_t) {
  operand_store_mode<T>();
  const float* x = reinterpret_cast<const float*>(x_any);      // the pre-LayerNorm sums: f32, or operand precision when x_is_t (fp16 models: the GEMM in front stored them so)
  const T* x_t = reinterpret_cast<const T*>(x_any);
  const int lane = threadIdx.x & 63;
  const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const float invE = 1.f / (float)E;
  for (long row = wave0; row < rows; row += (long)gridDim.x * 4) {
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (k * 64 + lane) * 4;
      v[k] = (c < E) ? (x_is_t ? ld4<T>(x_t + row * E + c) : *reinterpret_cast<const f32x4*>(x + row * E + c)) : f32x4{0.f, 0.f, 0.f, 0.f};
      s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
    }
    const float mu = wave_sum(s) * invE;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < E) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[k][e] - mu; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * invE + eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < E) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), bb = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[k][e] - mu) * rstd * g[e] + bb[e];
        if (y32) *reinterpret_cast<f32x4*>(y32 + row * E + c) = o;
        if (yt) st4<T>(yt + row * E + c, o);
      }
    }
    if (lane == 0) { mean_o[row] = mu; rstd_o[row] = rstd; }
  }
}
int launch_layernorm_fwd(const void* x, const float* gamma, const float* beta, float* y32, void* yt, float* mean, float* rstd,
                         long rows, int E, float eps, int precision, hipStream_t s, int x_is_t) {
  if (E % 4 || E > 2048 || (x_is_t && !prec_is16(precision))) return PFN_ERR_UNSUPPORTED;
  if (rows == 0) return PFN_OK;
  const int grid = grid_for(rows, 4, 8192);
#define LN_FWD(TT, NV) hipLaunchKernelGGL((layernorm_fwd_kernel<TT, NV>), dim3(grid), dim3(256), 0, s, x, gamma, beta, y32, (TT*)yt, mean, rstd, rows, E, eps, x_is_t)
#define LN_FWD_NV(TT) do { if (E <= 256) LN_FWD(TT, 1); else if (E <= 512) LN_FWD(TT, 2); else if (E <= 1024) LN_FWD(TT, 4); else LN_FWD(TT, 8); } while (0)
  PFN_DISPATCH_OP(precision, LN_FWD_NV(T));
  return PFN_LAUNCH_OK();
}

// 8 waves per workgroup, one wave per row, TWO rows in flight per wave (loads of both rows are
// issued before either is reduced), 2 workgroups per CU: enough bytes in flight to stream at HBM
// rate, and few enough workgroups that the column reductions (d gamma, d beta, optional bias
// gradient) end in ~0.8 M atomics instead of several million.
constexpr int LNB_WAVES = 8;
// EV contiguous elements per lane and 64 EV-element chunks per wave pass: EV = 8 makes every operand-precision access a
// 16-byte one (bf16 dY loads and dX stores of 8 bytes per lane stream at little more than half the rate).
template <typename T, int EV> PFN_DEV void ln_load(const T* p, float (&v)[EV]) {
  if constexpr (sizeof(T) == 2 && EV == 8) {
    const X8<T> t = *reinterpret_cast<const X8<T>*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
  } else {
#pragma unroll
    for (int j = 0; j < EV; j += 4) {
      const f32x4 t = ld4<T>(p + j);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[j + e] = t[e];
    }
  }
}
template <typename T, int EV> PFN_DEV void ln_store(T* p, const float (&v)[EV]) {
  if constexpr (sizeof(T) == 2 && EV == 8) {
    X8<T> t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (T)v[e];
    *reinterpret_cast<X8<T>*>(p) = t;
  } else {
#pragma unroll
    for (int j = 0; j < EV; j += 4) st4<T>(p + j, f32x4{v[j], v[j + 1], v[j + 2], v[j + 3]});
  }
}
template <typename T, int NV, int EV, bool DY_T>
__global__ __launch_bounds__(LNB_WAVES * 64) void layernorm_bwd_kernel(const void* dy_any, const void* x_any, const float* gamma, const float* mean, const float* rstd,
                                                                      float* dx32, T* dxt, float* dgamma, float* dbeta, float* dbias, long rows, int E, float* partials, const float* scale_amax, int x_is_t) {
  operand_store_mode<T>();
  extern __shared__ __attribute__((aligned(16))) float part[];  // [LNB_WAVES][E]
  const float osc = loss_scale_down(scale_amax);                // fp16 backward: the parameter gradients leave unscaled (pfn_device.h)
  const float* dy = reinterpret_cast<const float*>(dy_any);     // upstream gradient: f32, or operand precision when DY_T
  const T* dy_t = reinterpret_cast<const T*>(dy_any);
  const float* x = reinterpret_cast<const float*>(x_any);      // the LayerNorm's input rows: f32, or operand precision when x_is_t (GemmLN::y16 stored them)
  const T* x_t = reinterpret_cast<const T*>(x_any);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long wave0 = (long)blockIdx.x * LNB_WAVES + wave;
  const long wstride = (long)gridDim.x * LNB_WAVES;
  const float invE = 1.f / (float)E;
  float ag[NV][EV], ab[NV][EV], ax[NV][EV], gam[NV][EV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = (k * 64 + lane) * EV;
#pragma unroll
    for (int e = 0; e < EV; ++e) { ag[k][e] = 0.f; ab[k][e] = 0.f; ax[k][e] = 0.f; gam[k][e] = 0.f; }
    if (c < E) ln_load<float, EV>(gamma + c, gam[k]);
  }
  for (long row0 = wave0; row0 < rows; row0 += 2 * wstride) {
    const long rw[2] = {row0, row0 + wstride};
    float xv[2][NV][EV], dv[2][NV][EV];
    float mu[2], rs[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bool live = rw[u] < rows;
      const long r = live ? rw[u] : row0;
      mu[u] = mean[r]; rs[u] = rstd[r];
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int c = (k * 64 + lane) * EV;
        if (c < E) {
          if (x_is_t) ln_load<T, EV>(x_t + r * E + c, xv[u][k]);
          else ln_load<float, EV>(x + r * E + c, xv[u][k]);
          if constexpr (DY_T) ln_load<T, EV>(dy_t + r * E + c, dv[u][k]);
          else ln_load<float, EV>(dy + r * E + c, dv[u][k]);
        }
#pragma unroll
        for (int e = 0; e < EV; ++e) {
          if (c >= E) xv[u][k][e] = 0.f;
          if (c >= E || !live) dv[u][k][e] = 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (rw[u] >= rows) break;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < EV; ++e) {
          const float xh = (xv[u][k][e] - mu[u]) * rs[u];
          const float g = dv[u][k][e] * gam[k][e];
          s1 += g;
          s2 += g * xh;
          ag[k][e] += dv[u][k][e] * xh;
          ab[k][e] += dv[u][k][e];
          xv[u][k][e] = xh;
          dv[u][k][e] = g;
        }
      s1 = wave_sum(s1) * invE;
      s2 = wave_sum(s2) * invE;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int c = (k * 64 + lane) * EV;
        if (c < E) {
          float o[EV];
#pragma unroll
          for (int e = 0; e < EV; ++e) { o[e] = rs[u] * (dv[u][k][e] - s1 - xv[u][k][e] * s2); ax[k][e] += o[e]; }
          if (dx32) ln_store<float, EV>(dx32 + rw[u] * E + c, o);
          if (dxt) ln_store<T, EV>(dxt + rw[u] * E + c, o);
        }
      }
    }
  }
  // block reduction of the column partials (one quantity at a time through [LNB_WAVES][E] of LDS), then one
  // atomic per column per block
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    float* out = q == 0 ? dgamma : q == 1 ? dbeta : dbias;
    if (out == nullptr) continue;
    if (q > 0) __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = (k * 64 + lane) * EV;
      if (c < E) ln_store<float, EV>(part + wave * E + c, q == 0 ? ag[k] : q == 1 ? ab[k] : ax[k]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < E; c += LNB_WAVES * 64) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < LNB_WAVES; ++w) t += part[w * E + c];
      t *= osc;
      // deterministic schedule (PFN_SCHED_DETERMINISTIC): the block's partial goes to scratch, ln_partials_reduce_kernel adds the blocks in index order
      if (partials) partials[((long)blockIdx.x * 3 + q) * E + c] = t;
      else unsafeAtomicAdd(out + c, t);
    }
  }
}
// out_q[c] += sum over blocks (in index order) of partials[block][q][c]: one writer per element, a fixed summation order
__global__ __launch_bounds__(256) void ln_partials_reduce_kernel(const float* partials, int nblocks, int E, float* dgamma, float* dbeta, float* dbias) {
  const int c = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y;
  float* out = q == 0 ? dgamma : q == 1 ? dbeta : dbias;
  if (c >= E || out == nullptr) return;
  float t = 0.f;
  for (int b = 0; b < nblocks; ++b) t += partials[((long)b * 3 + q) * E + c];
  out[c] += t;
}
int launch_layernorm_bwd(const void* dy, int dy_is_t, const void* x, const float* gamma, const float* mean, const float* rstd, float* dx32, void* dxt,
                         float* dgamma, float* dbeta, float* dbias, long rows, int E, int precision, hipStream_t s, float* partials, const float* scale_amax, int x_is_t) {
  if (E % 4 || E > 2048 || (x_is_t && !prec_is16(precision))) return PFN_ERR_UNSUPPORTED;
  if (rows == 0) return PFN_OK;
  const int grid = grid_for(rows, LNB_WAVES * 8, LNB_MAX_BLOCKS);
  const size_t lds = LNB_WAVES * E * sizeof(float);
#define LN_BWD_K(TT, NV, EV, DT) do { \
    static LdsAllowance allowance; \
    allowance.ensure(layernorm_bwd_kernel<TT, NV, EV, DT>, lds); \
    hipLaunchKernelGGL((layernorm_bwd_kernel<TT, NV, EV, DT>), dim3(grid), dim3(LNB_WAVES * 64), lds, s, dy, x, gamma, mean, rstd, dx32, (TT*)dxt, dgamma, dbeta, dbias, rows, E, partials, scale_amax, x_is_t); } while (0)
#define LN_BWD(TT, NV, EV) do { if (dy_is_t) LN_BWD_K(TT, NV, EV, true); else LN_BWD_K(TT, NV, EV, false); } while (0)
  // rows of >= 512 elements in 8-element lane chunks (16-byte operand-precision accesses), narrower rows in 4-element ones
#define LN_BWD_NV(TT) do { \
    if (E % 8 == 0 && E >= 512) { if (E <= 512) LN_BWD(TT, 1, 8); else if (E <= 1024) LN_BWD(TT, 2, 8); else LN_BWD(TT, 4, 8); } \
    else if (E <= 256) LN_BWD(TT, 1, 4); else if (E <= 512) LN_BWD(TT, 2, 4); else if (E <= 1024) LN_BWD(TT, 4, 4); else LN_BWD(TT, 8, 4); } while (0)
  PFN_DISPATCH_OP(precision, LN_BWD_NV(T));
  if (partials) hipLaunchKernelGGL(ln_partials_reduce_kernel, dim3((E + 255) / 256, 3), dim3(256), 0, s, partials, grid, E, dgamma, dbeta, dbias);
  return PFN_LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------
// out[n] += sum_m a[m, n]     (bias gradients of linears whose dY is stored in T)
// ---------------------------------------------------------------------------------------------
constexpr int CS_ROWS = 256;
template <typename T> __global__ __launch_bounds__(256) void colsum_kernel(const T* a, long lda, long rows, int cols, float* out) {
  __shared__ float red[256 * 4];
  const int cg = (cols + 3) / 4;                 // column groups of 4
  const int CG = cg < 256 ? cg : 256;            // groups handled per block pass
  const int rsub = threadIdx.x / CG, nrs = 256 / CG;
  const int g = blockIdx.y * CG + threadIdx.x % CG;
  const long r0 = (long)blockIdx.x * CS_ROWS;
  const long r1 = r0 + CS_ROWS < rows ? r0 + CS_ROWS : rows;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (rsub < nrs && g < cg) {
    const int c = g * 4;
    for (long r = r0 + rsub; r < r1; r += nrs) {
      if (c + 3 < cols) acc += ld4<T>(a + r * lda + c);
      else for (int e = 0; e < 4 && c + e < cols; ++e) acc[e] += (float)a[r * lda + c + e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[threadIdx.x * 4 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < CG && g < cg) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < nrs; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] += red[(j * CG + threadIdx.x) * 4 + e];
    for (int e = 0; e < 4 && g * 4 + e < cols; ++e) unsafeAtomicAdd(out + g * 4 + e, t[e]);
  }
}
int launch_colsum(const void* a, long lda, long rows, int cols, float* out, int precision, hipStream_t s) {
  if (rows == 0 || cols == 0) return PFN_OK;
  const int es = prec_esize(precision);
  if ((lda * es) % 8) return PFN_ERR_ALIGNMENT;
  const int cg = (cols + 3) / 4;
  dim3 grid((unsigned)((rows + CS_ROWS - 1) / CS_ROWS), (cg + 255) / 256);
  PFN_DISPATCH_OP(precision, hipLaunchKernelGGL(colsum_kernel<T>, grid, dim3(256), 0, s, (const T*)a, lda, rows, cols, out));
  return PFN_LAUNCH_OK();
}

}  // namespace pfn
